"""Mirror of the hot-path symbols of models/pointbert/misc.py: fps, index_points,
farthest_point_sample (misc.py:12-69) on the HIP FPS kernel."""
import torch

from ... import ops


def farthest_point_sample(xyz, npoint, start_idx=None):
    """misc.py:44-69.  xyz [B,N,3] -> centroids [B,npoint] int64.  The reference draws the start
    index with torch.randint (misc.py:59); pass start_idx [B] to inject it (parity tests)."""
    B, N, _ = xyz.shape
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device)
    idx, _ = ops.fps(xyz.contiguous().float(), npoint, start_idx.to(xyz.device).contiguous())
    return idx


def index_points(points, idx):
    """misc.py:26-42.  points [B,N,C], idx [B,S] -> [B,S,C] (pure gather: torch indexing)."""
    B = points.shape[0]
    batch = torch.arange(B, dtype=torch.long, device=points.device).view(B, *([1] * (idx.dim() - 1))).expand_as(idx)
    return points[batch, idx, :]


def fps(data, number, start_idx=None):
    """misc.py:12-24.  data [B,N,3] -> sampled points [B,number,3] (index + gather fused in the kernel)."""
    B, N, _ = data.shape
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long, device=data.device)
    _, ctr = ops.fps(data.contiguous().float(), number, start_idx.to(data.device).contiguous())
    return ctr
