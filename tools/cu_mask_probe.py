"""Dev tool: does reserving CUs for the prompt chain pay?  The caller's stream (point tower + head) is created with a CU mask
that leaves `reserve` CUs (taken evenly from the XCDs: mask bit i -> XCD i % 8) to the text stream alone.
    python tools/cu_mask_probe.py C2 <reserve> [tower_own_stream 0|1]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.train import Trainer

name = sys.argv[1]
reserve = int(sys.argv[2])
torch.cuda.set_device(0)
cfg = bench.CONFIGS[name]
graphs.shared_text_stream(priority=-1 if cfg["head_type"] == 0 else 0)
hip = ctypes.CDLL("libamdhip64.so")
ncu = torch.cuda.get_device_properties(0).multi_processor_count
main = torch.cuda.current_stream()
if reserve:
    bits = [1] * ncu
    layout = sys.argv[3] if len(sys.argv) > 3 else "block"
    for i in range(max(reserve, 0)):
        if layout == "block":                    # bit -> XCD bit // 32: take CU (31 - i // 8) of XCD i % 8
            bits[32 * (i % 8) + 31 - i // 8] = 0
        elif layout == "low":
            bits[32 * (i % 8) + i // 8] = 0
        else:
            bits[ncu - 1 - i] = 0
    words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), words)
    assert rc == 0, rc
    main = torch.cuda.ExternalStream(s.value)
model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
model.train()
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
label = torch.randint(0, len(model.prompt_learner.classnames), (B,), device="cuda")
torch.cuda.synchronize()


def timed(fn, n=40, warm=12):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.cuda.stream(main):
    full = min(timed(lambda: tr.step(pc, label)) for _ in range(3))
    tr.finish()
    with torch.no_grad():
        tower = timed(lambda: model.point_encoder(pc))
print(f"{name} reserve {reserve} of {ncu} CUs: full step {full:.3f} ms | point tower alone {tower:.3f} ms", flush=True)
