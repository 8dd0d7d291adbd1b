"""Mirror of the part-segmentation symbols of models/pointbert/pointnet2_utils.py: knn_point (20), square_distance
(51), index_points (75), farthest_point_sample (95), PointNetFeaturePropagation (297-368), DGCNN_Propagation
(371-467) -- identical names / state-dict keys.  Neighbour searches (3-NN, k=4 kNN, FPS) and every 1x1 convolution
run on the HIP kernels (ppt_knn_group_f32, ppt_fps_f32, ppt_gemm via ppt_amd.autograd.linear, which also
provides the weight gradients the decoder needs) and so does BatchNorm1d + ReLU with its backward
(ppt_amd.autograd.batch_norm_relu_rows); GroupNorm / LeakyReLU / the gathers of the DGCNN propagation and their backward
are still ATen ops in this round -- see DESIGN.md section 8."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...autograd import batch_norm_relu_rows, gather_add_rows, group_norm_lrelu_max, linear
from .dvae import knn_point, square_distance          # noqa: F401  (same semantics as pointnet2_utils.py:20-72)
from .misc import farthest_point_sample, index_points  # noqa: F401


def _gather_rows(points, idx):
    """points [B,S,C], idx [B,N,k] -> [B,N,k,C]."""
    B = points.shape[0]
    return points[torch.arange(B, device=points.device)[:, None, None], idx]


class PointNetFeaturePropagation(nn.Module):
    """pointnet2_utils.py:297-368.  Row layout: xyz1 [B,N,3] targets, xyz2 [B,S,3] sources, points1 [B,N,D1] | None,
    points2 [B,S,D2] -> [B,N,mlp[-1]]."""

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
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interpolated = points2.repeat(1, N, 1)
        else:
            # 3 nearest sources under (distance, index) == the reference's full sort + [:3]; expanded-form distances
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
    """pointnet2_utils.py:371-467.  coor [B,S,3], f [B,S,C], coor_q [B,Nq,3], f_q [B,Nq,C] -> [B,Nq,384]."""

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
        C = x_q.shape[-1]
        w = conv.weight.reshape(conv.weight.shape[0], -1)
        wa = w[:, :C].contiguous()
        wd = (w[:, C:] - w[:, :C]).contiguous()
        P = linear(x_k, wa, None, self.precision)                              # [B,S,Cout]
        Q = linear(x_q, wd, None, self.precision)                              # [B,Nq,Cout]
        y = gather_add_rows(P, Q, idx)                                         # [B,Nq,k,Cout]
        return group_norm_lrelu_max(y, gn, 0.2)                                # GroupNorm + LeakyReLU(0.2) + max over k: [B,Nq,Cout]

    def forward(self, coor, f, coor_q, f_q):
        f_q = self._layer(self.layer1, coor_q, f_q, coor, f)
        f_q = self._layer(self.layer2, coor_q, f_q, coor_q, f_q)
        return f_q
