"""Mirror of the hot-path symbols of models/pointbert/dvae.py: knn_point, square_distance,
Group, Encoder (dvae.py:116-215).  The dVAE pre-training classes are out of scope."""
import torch
import torch.nn as nn

from ... import engine, ops
from . import misc


def square_distance(src, dst):
    """dvae.py:130-149: src [B,N,3], dst [B,M,3] -> [B,N,M], the expanded form -2 src.dst + |src|^2 + |dst|^2 with the
    rounding sequence of the reference's CPU path (ppt_square_distance_f32).  The product path never
    materialises this matrix (ppt_knn_group_f32 fuses it with the selection); this serves direct callers."""
    return ops.square_distance(src.contiguous().float(), dst.contiguous().float())


def knn_point(nsample, xyz, new_xyz):
    """dvae.py:116-127 -> group_idx [B,S,nsample] int64, sorted by (distance, index)."""
    idx, _ = ops.knn_group(xyz.contiguous().float(), new_xyz.contiguous().float(), nsample, want_nbhd=False)
    return idx


class Group(nn.Module):
    """dvae.py:152-181."""

    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group = num_group
        self.group_size = group_size

    def forward(self, xyz, start_idx=None):
        """xyz [B,N,3] -> (neighborhood [B,G,M,3] centre-subtracted, center [B,G,3])."""
        xyz = xyz.contiguous().float()
        center = misc.fps(xyz, self.num_group, start_idx)
        idx, neighborhood = ops.knn_group(xyz, center, self.group_size)
        assert idx.size(1) == self.num_group
        assert idx.size(2) == self.group_size
        return neighborhood, center


class Encoder(nn.Module):
    """dvae.py:184-215 mini-PointNet; parameters / buffers keep the reference names
    (first_conv.{0,1,3}, second_conv.{0,1,3})."""

    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, self.encoder_channel, 1))
        self.precision = torch.bfloat16
        self._wc = None

    def forward(self, point_groups):
        """point_groups [B,G,N,3] -> [B,G,C] (fp32 view of the kernel's output)."""
        if self._wc is None or self._wc.dtype != self.precision:
            self._wc = engine.WeightCache(self.precision)
        sd = self.state_dict(keep_vars=True)
        B, G = point_groups.shape[:2]
        with torch.no_grad():
            tok = engine.mini_pointnet(sd, "", self._wc, point_groups.contiguous().float(), self.training)
        return tok.float().view(B, G, self.encoder_channel)
