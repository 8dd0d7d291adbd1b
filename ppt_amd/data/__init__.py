"""Host-side data helpers on the path's kernels: the dataset's numpy farthest point sampling."""
from .dataset_3d import farthest_point_sample, pc_normalize  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
from .fps_service import start_fps_service, stop_fps_service  # noqa: F401
