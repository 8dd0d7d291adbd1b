"""data/dataset_3d.py:33-61 of the reference: `pc_normalize` and the dataset-side `farthest_point_sample(point, npoint)` -- a
Python loop of `npoint` numpy passes per cloud that the datasets run in `_get_item` whenever a stored cloud has more points than
requested (:295, :366, :583: 8192 -> 1024 is 1024 passes over 8192 points per sample).  Here the walk is ONE launch of the
path's FPS kernel (csrc/fps.hip, the same kernel the PointBERT tokenizer uses): numpy in, numpy out, same rows selected.

Bit-exactness: the reference computes `dist` in the array's dtype -- float32 at every call site (the stored clouds are
`.astype(np.float32)`, :579, :742) -- as (dx^2 + dy^2) + dz^2, keeps the running minimum with a strict `<` and takes the first
arg-max; the kernel does exactly that in fp32, so the selected indices are identical (tests/test_kernels_gpu.py against
oracle.dataset_farthest_point_sample, which tests/golden/make_golden.py pins to the reference function).  A float64 array would
be summed in float64 by the reference; that is not what the kernel computes, so it is rejected rather than approximated.
"""
import numpy as np


def pc_normalize(pc):
    """data/dataset_3d.py:33-38: centre on the centroid, scale to the unit sphere (host arithmetic, as the reference)."""
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    return pc / m


def farthest_point_sample(point, npoint, start=None, return_index=False):
    """point [N, D] float32 (xyz in the first three columns), npoint -> point[centroids] [npoint, D] (data/dataset_3d.py:40-61).
    start: the first index; None draws `np.random.randint(0, N)` exactly where the reference does (:51), so a seeded loader
    selects the same points.  Runs on the current HIP device; there is no CPU fallback.

    Where to call it: the reference runs this loop inside Dataset._get_item, i.e. in DataLoader WORKER processes
    (num_workers = args.workers, fork start method).  A forked worker cannot initialise the device ("Cannot re-initialize CUDA in
    forked subprocess") and a spawned one would build a GPU context per worker for one tiny launch per sample.  Round 5: with
    `ppt_amd.data.start_fps_service()` called once in the main process before the loader forks, a worker hands its cloud (and the
    start index it drew) to a main-process thread that runs the launch and returns the indices -- the dataset class stays unchanged
    (ppt_amd/data/fps_service.py).  Without the service this function raises inside a worker and says what to do instead.
    INTEGRATION.md, "dataset-side FPS"."""
    import torch
    from .. import ops
    from . import fps_service
    winfo = torch.utils.data.get_worker_info()
    if winfo is not None and fps_service.service() is None:
        raise RuntimeError("ppt_amd.data.farthest_point_sample was called inside a DataLoader worker process: it runs on the HIP device "
                           "and has no CPU path, and a forked worker cannot initialise the device.  Call ppt_amd.data.start_fps_service() "
                           "in the main process before the DataLoader is iterated (the workers then hand their clouds to that thread), or "
                           "sample in the main process -- a collate_fn, or batched on the device with ppt_amd.ops.fps(pc [B,N,3], npoint, "
                           "start [B]) -- or use num_workers=0 (INTEGRATION.md, dataset-side FPS).")
    point = np.asarray(point)
    if point.ndim != 2 or point.shape[1] < 3:
        raise ValueError(f"point must be [N, D >= 3], got {point.shape}")
    if point.dtype != np.float32:
        raise TypeError(f"farthest_point_sample: float32 clouds only (got {point.dtype}): the reference's call sites hold float32 "
                        "arrays and a float64 array would be walked in float64 arithmetic, which the fp32 kernel does not reproduce")
    N = point.shape[0]
    if start is None:
        start = np.random.randint(0, N)
    if winfo is not None:
        # a DataLoader worker: the launch runs in the main process (ppt_amd/data/fps_service.py), the draw above stays here
        idx = fps_service.service().request(winfo.id, point[:, :3], int(npoint), int(start))
        out = point[idx.astype(np.int32)]
        return (out, idx) if return_index else out
    if not torch.cuda.is_available():
        raise RuntimeError("ppt_amd.data.farthest_point_sample needs the HIP device (libppt_hip.so: ppt_fps_f32)")
    xyz = torch.from_numpy(np.ascontiguousarray(point[:, :3])).cuda().view(1, N, 3)
    st = torch.tensor([int(start)], dtype=torch.int64, device=xyz.device)
    idx, _ = ops.fps(xyz, int(npoint), st)
    idx = idx.view(-1).cpu().numpy()
    out = point[idx.astype(np.int32)]
    return (out, idx) if return_index else out
