"""ppt_amd: the MI355X-native hot path of auniquesun/PPT."""
import os

# The step runs on two HIP streams (point tower | prompt side).  The runtime deals streams round-robin onto 4 hardware
# queues by default; once RCCL has created its own streams, the text stream can land on the queue of the caller's
# stream, and the two then execute serially (measured: 4.8 -> 6.2 ms per C2 step after init_process_group alone).
# Eight queues keep them apart.  Must be set before the HIP runtime initialises, i.e. import ppt_amd (or bench.py)
# before the first torch.cuda call; an explicit GPU_MAX_HW_QUEUES in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
