"""Mirror of the part-segmentation symbols of models/pointbert/pointnet2_utils.py: knn_point (20), square_distance
(51), index_points (75), farthest_point_sample (95), PointNetFeaturePropagation (297-368), DGCNN_Propagation
(371-467) -- identical names / state-dict keys.  Neighbour searches (3-NN, k=4 kNN, FPS) and every 1x1 convolution
run on the HIP kernels (ppt_knn_group_f32, ppt_fps_f32, ppt_gemm via ppt_amd.autograd.linear, which also
provides the weight gradients the decoder needs), and so do the 3-NN interpolation + concatenation (ppt_three_nn_interp_fwd),
BatchNorm1d + ReLU, GroupNorm + LeakyReLU + max, and every backward including the gathers' (ppt_scatter_rows_bwd): each
reference module is one autograd node of ppt_amd.autograd."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...autograd import (batch_norm_relu_rows, dgcnn_layer, feature_propagation, gather_add_rows, group_norm_lrelu_max,
                         linear)
from .dvae import knn_point, square_distance          # noqa: F401  (same semantics as pointnet2_utils.py:20-72)
from .misc import farthest_point_sample, index_points  # noqa: F401


def _gather_rows(points, idx):
    """points [B,S,C], idx [B,N,k] -> [B,N,k,C]."""
    B = points.shape[0]
    return points[torch.arange(B, device=points.device)[:, None, None], idx]


class PointNetFeaturePropagation(nn.Module):
    """pointnet2_utils.py:297-368.  forward() takes the reference's channel-first tensors; the model itself calls
    forward_rows() (points are rows everywhere in this build)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel
        self.precision = torch.bfloat16

    def forward(self, xyz1, xyz2, points1, points2):
        """Reference layout (pointnet2_utils.py:310-326): xyz1 [B,3,N] targets, xyz2 [B,3,S] sources, points1 [B,D1,N] |
        None, points2 [B,D2,S] -> [B,mlp[-1],N].  The compute runs on rows (forward_rows)."""
        p1 = points1.permute(0, 2, 1) if points1 is not None else None
        out = self.forward_rows(xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1), p1, points2.permute(0, 2, 1))
        return out.permute(0, 2, 1)

    def forward_rows(self, xyz1, xyz2, points1, points2):
        """Row layout: xyz1 [B,N,3], xyz2 [B,S,3], points1 [B,N,D1] | None, points2 [B,S,D2] -> [B,N,mlp[-1]]."""
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S > 1 and len(self.mlp_convs) == 2:
            # 3 nearest sources under (distance, index) == the reference's full sort + [:3]; expanded-form distances.
            # Interpolation, concatenation, both conv + BN + ReLU layers and their backward: one node (ppt_amd.autograd)
            with torch.no_grad():
                idx, _, d = ops.knn_group(xyz2.contiguous().float(), xyz1.contiguous().float(), 3, want_nbhd=False, want_dist=True)
            return feature_propagation(self, points1, points2, idx, d, self.precision)
        if S == 1:
            interpolated = points2.repeat(1, N, 1)
        else:
            idx, _, d = ops.knn_group(xyz2.contiguous(), xyz1.contiguous(), 3, want_nbhd=False, want_dist=True)
            recip = 1.0 / (d + 1e-8)
            weight = recip / recip.sum(dim=2, keepdim=True)
            interpolated = (_gather_rows(points2, idx) * weight.unsqueeze(-1)).sum(dim=2)
        x = interpolated if points1 is None else torch.cat([points1, interpolated], dim=-1)
        x = x.reshape(B * N, -1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = linear(x, conv.weight, conv.bias, self.precision)
            x = batch_norm_relu_rows(x, bn, self.training)        # statistics, ReLU and backward on the HIP kernels
        return x.view(B, N, -1)


class DGCNN_Propagation(nn.Module):
    """pointnet2_utils.py:371-467.  forward() takes the reference's channel-first tensors, forward_rows() the row layout."""

    def __init__(self, k=16):
        super().__init__()
        self.k = k
        self.layer1 = nn.Sequential(nn.Conv2d(768, 512, kernel_size=1, bias=False), nn.GroupNorm(4, 512),
                                    nn.LeakyReLU(negative_slope=0.2))
        self.layer2 = nn.Sequential(nn.Conv2d(1024, 384, kernel_size=1, bias=False), nn.GroupNorm(4, 384),
                                    nn.LeakyReLU(negative_slope=0.2))
        self.precision = torch.bfloat16

    def get_graph_feature(self, coor_q, x_q, coor_k, x_k):
        """-> [B,Nq,k,2C] = cat(x_k[nn] - x_q, x_q)."""
        with torch.no_grad():
            idx, _ = ops.knn_group(coor_k.contiguous(), coor_q.contiguous(), self.k, want_nbhd=False)
        assert idx.shape[2] == self.k
        nb = _gather_rows(x_k, idx)
        xq = x_q.unsqueeze(2).expand(-1, -1, self.k, -1)
        return torch.cat((nb - xq, xq), dim=-1)

    def _layer(self, seq, coor_q, x_q, coor_k, x_k):
        """conv(cat(x_k[nn] - x_q, x_q)) + GroupNorm + LeakyReLU + max over the k neighbours (:404-440).  The 1x1 conv is
        linear, W = [Wa | Wb]: Wa.(x_j - x_q) + Wb.x_q = Wa.x_j + (Wb - Wa).x_q -- so it runs once per SOURCE point and once
        per query point (k times fewer rows, half the channels) and ppt_gather_add forms the [B,Nq,k,Cout] rows; the
        gathered / concatenated graph feature is never built."""
        conv, gn = seq[0], seq[1]
        with torch.no_grad():
            idx, _ = ops.knn_group(coor_k.contiguous(), coor_q.contiguous(), self.k, want_nbhd=False)
        assert idx.shape[2] == self.k
        return dgcnn_layer(x_k, x_q, conv, gn, idx, 0.2, self.precision)       # [B,Nq,Cout]: one autograd node per layer

    def forward(self, coor, f, coor_q, f_q):
        """Reference layout (pointnet2_utils.py:433): coor [B,3,S], f [B,C,S], coor_q [B,3,Nq], f_q [B,C,Nq] -> [B,384,Nq]."""
        out = self.forward_rows(coor.permute(0, 2, 1), f.permute(0, 2, 1), coor_q.permute(0, 2, 1), f_q.permute(0, 2, 1))
        return out.permute(0, 2, 1)

    def forward_rows(self, coor, f, coor_q, f_q):
        """Row layout: coor [B,S,3], f [B,S,C], coor_q [B,Nq,3], f_q [B,Nq,C] -> [B,Nq,384]."""
        f_q = self._layer(self.layer1, coor_q, f_q, coor, f)
        f_q = self._layer(self.layer2, coor_q, f_q, coor_q, f_q)
        return f_q
