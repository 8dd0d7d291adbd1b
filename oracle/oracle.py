"""CPU restatement (fp32, torch-CPU tensor math + the C index oracle) of PPT's point-cloud
encoder hot path.  PARITY PINNED by tests/golden/*.npz, which tests/golden/make_golden.py
produced by importing the upstream reference (/root/reference) in the build container.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this module, and only as the checker / reported CPU baseline.  Nothing under
ppt_amd/ imports it; the product path raises if the HIP library is missing.

Everything is a pure function over a flat {state-dict key: tensor} mapping using the reference's
key names.  Citations are relative to /root/reference/.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libppt_oracle.so")
_lib = None

ERF_GELU = "gelu"
QUICK_GELU = "quick_gelu"


# ------------------------------------------------------------------------------------------------
# C index oracle (oracle/ppt_oracle.c)
# ------------------------------------------------------------------------------------------------
_C_FLAGS = ["-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC"]


def build_c_oracle(force=False):
    """Compile oracle/ppt_oracle.c.  The .so is reused only when the sha256 of the SOURCE BYTES (and the flags) it was built
    from is the one on record next to it: the checker travels with the gpurun snapshot, where modification times mean
    nothing -- a fresh checkout with new mtimes must never pair a stale checker with a newer source (VERDICT r4 weak #13;
    the product build, ppt_amd/build.py, keys on content the same way)."""
    import hashlib
    src = os.path.join(_HERE, "ppt_oracle.c")
    stamp = _SO + ".sha256"
    with open(src, "rb") as fh:
        want = hashlib.sha256(" ".join(_C_FLAGS).encode() + b"\0" + fh.read()).hexdigest()
    have = None
    if os.path.exists(_SO) and os.path.exists(stamp):
        with open(stamp) as fh:
            have = fh.read().strip()
    if force or have != want:
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        tmp = _SO + f".{os.getpid()}.tmp"                    # (concurrent test processes: build aside, rename into place)
        subprocess.check_call(["gcc"] + _C_FLAGS + [src, "-o", tmp, "-lm"])
        os.replace(tmp, _SO)
        with open(stamp + f".{os.getpid()}.tmp", "w") as fh:
            fh.write(want + "\n")
        os.replace(stamp + f".{os.getpid()}.tmp", stamp)
    return _SO


def _c():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_c_oracle())
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fps(xyz, M, start):
    """misc.py:44-69 farthest_point_sample with the start index injected.
    xyz [B,N,3] f32, start [B] -> idx [B,M] int64 (numpy in, numpy out)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    start = np.ascontiguousarray(start, np.int64)
    B, N, _ = xyz.shape
    out = np.empty((B, M), np.int64)
    _c().oracle_fps_f32(_fp(xyz), B, N, M, _fp(start), _fp(out))
    return out


def dataset_farthest_point_sample(point, npoint, start):
    """data/dataset_3d.py:40-61 farthest_point_sample(point [N,D], npoint) -> point[centroids] [npoint,D], with the random
    start (np.random.randint, :51) injected.  Restated as the loop it is: float32 coordinates, `dist` in the array's dtype
    ((dx^2 + dy^2) + dz^2, numpy's 3-term sum), the running `distance` in float64 starting at 1e10, strict `<` update, first
    arg-max.  Returns (rows, indices)."""
    point = np.asarray(point)
    xyz = point[:, :3]
    N = xyz.shape[0]
    idx = np.zeros((npoint,), np.int64)
    distance = np.ones((N,)) * 1e10
    far = int(start)
    for i in range(npoint):
        idx[i] = far
        d = xyz - xyz[far]
        dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        mask = dist < distance
        distance[mask] = dist[mask]
        far = int(np.argmax(distance))
    return point[idx], idx


def square_distance(src, dst):
    """dvae.py:130-149.  src [B,S,3], dst [B,N,3] -> [B,S,N]."""
    src = np.ascontiguousarray(src, np.float32)
    dst = np.ascontiguousarray(dst, np.float32)
    B, S, _ = src.shape
    N = dst.shape[1]
    out = np.empty((B, S, N), np.float32)
    _c().oracle_square_distance_f32(_fp(src), _fp(dst), B, S, N, _fp(out))
    return out


def knn(xyz, query, k):
    """dvae.py:116-127 knn_point -> (idx [B,S,k] sorted by (d, index), kth [B,S,2])."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    B, N, _ = xyz.shape
    S = query.shape[1]
    idx = np.empty((B, S, k), np.int64)
    kth = np.empty((B, S, 2), np.float32)
    _c().oracle_knn_f32(_fp(xyz), _fp(query), B, N, S, k, _fp(idx), _fp(kth))
    return idx, kth


def group(xyz, center_idx, k):
    """dvae.py:159-181 Group.forward given the FPS indices.
    -> nbr_idx [B,G,k], neighborhood [B,G,k,3], center [B,G,3]."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    center_idx = np.ascontiguousarray(center_idx, np.int64)
    B, N, _ = xyz.shape
    G = center_idx.shape[1]
    nbr = np.empty((B, G, k), np.int64)
    nb = np.empty((B, G, k, 3), np.float32)
    ce = np.empty((B, G, 3), np.float32)
    _c().oracle_group_f32(_fp(xyz), _fp(center_idx), B, N, G, k, _fp(nbr), _fp(nb), _fp(ce))
    return nbr, nb, ce


def ball_query(xyz, query, radius, K):
    """pointnet2/pointnet2_utils.py:87-107 query_ball_point -> idx [B,S,K]."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    B, N, _ = xyz.shape
    S = query.shape[1]
    idx = np.empty((B, S, K), np.int64)
    _c().oracle_ball_query_f32(_fp(xyz), _fp(query), B, N, S, ctypes.c_double(radius), K, _fp(idx))
    return idx


def three_nn(xyz1, xyz2):
    """pointbert/pointnet2_utils.py:333-343 -> idx [B,N,3], weight [B,N,3]."""
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = np.empty((B, N, 3), np.int64)
    w = np.empty((B, N, 3), np.float32)
    _c().oracle_three_nn_f32(_fp(xyz1), _fp(xyz2), B, N, S, _fp(idx), _fp(w))
    return idx, w


# ------------------------------------------------------------------------------------------------
# elementary tensor math (written out; no nn.Module / functional wrappers)
# ------------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def quick_gelu(x):
    """ULIP_models.py:30-32."""
    return x * torch.sigmoid(1.702 * x)


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def batch_norm_rows(x, sd, prefix, train, eps=1e-5, momentum=0.1, new_stats=None):
    """nn.BatchNorm1d over a [rows, C] matrix (the Conv1d layout [BG,C,n] flattened to rows).
    train: biased batch variance normalises; running stats get the UNBIASED variance with
    momentum 0.1.  new_stats (dict) receives the updated running buffers."""
    w, b = sd[prefix + "weight"], sd[prefix + "bias"]
    if train:
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        if new_stats is not None:
            n = x.shape[0]
            new_stats[prefix + "running_mean"] = (1 - momentum) * sd[prefix + "running_mean"] + momentum * mean.detach()
            new_stats[prefix + "running_var"] = (1 - momentum) * sd[prefix + "running_var"] \
                + momentum * var.detach() * (n / (n - 1))
            new_stats[prefix + "num_batches_tracked"] = sd[prefix + "num_batches_tracked"] + 1
    else:
        mean, var = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    return (x - mean) / torch.sqrt(var + eps) * w + b


# ------------------------------------------------------------------------------------------------
# point branch
# ------------------------------------------------------------------------------------------------
def mini_pointnet(sd, neighborhood, train, prefix="point_encoder.encoder.", new_stats=None):
    """dvae.py:184-215 Encoder.forward.  neighborhood [B,G,n,3] -> [B,G,256].
    The k=1 Conv1d layers are per-point linear maps, so the [BG,C,n] tensors are handled as
    [BG*n, C] row matrices."""
    B, G, n, _ = neighborhood.shape
    x = neighborhood.reshape(B * G * n, 3)
    f = linear(x, sd[prefix + "first_conv.0.weight"][:, :, 0], sd[prefix + "first_conv.0.bias"])
    f = torch.relu(batch_norm_rows(f, sd, prefix + "first_conv.1.", train, new_stats=new_stats))
    f = linear(f, sd[prefix + "first_conv.3.weight"][:, :, 0], sd[prefix + "first_conv.3.bias"])   # [BGn,256]
    f = f.reshape(B * G, n, 256)
    g = f.max(dim=1, keepdim=True)[0]                                   # dvae.py:210
    f = torch.cat([g.expand(-1, n, -1), f], dim=2).reshape(B * G * n, 512)   # dvae.py:211 (global first)
    f = linear(f, sd[prefix + "second_conv.0.weight"][:, :, 0], sd[prefix + "second_conv.0.bias"])
    f = torch.relu(batch_norm_rows(f, sd, prefix + "second_conv.1.", train, new_stats=new_stats))
    f = linear(f, sd[prefix + "second_conv.3.weight"][:, :, 0], sd[prefix + "second_conv.3.bias"])
    return f.reshape(B * G, n, -1).max(dim=1)[0].reshape(B, G, -1)      # dvae.py:213-214


def vit_attention(x, w_qkv, w_proj, b_proj, heads):
    """point_encoder.py:33-58: no qkv bias, scale applied AFTER q@k^T."""
    B, T, C = x.shape
    hd = C // heads
    qkv = linear(x, w_qkv).reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    a = torch.softmax(a, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, T, C)
    return linear(o, w_proj, b_proj)


def vit_block(sd, p, x, heads, dp1=None, dp2=None):
    """point_encoder.py:61-79 Block.forward.  dp1/dp2: per-sample DropPath factors [B]
    (0 or 1/keep; timm 0.4.12 semantics) or None."""
    a = vit_attention(layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"]),
                      sd[p + "attn.qkv.weight"], sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], heads)
    if dp1 is not None:
        a = a * dp1.view(-1, 1, 1)
    x = x + a
    h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = gelu_erf(linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    h = linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if dp2 is not None:
        h = h * dp2.view(-1, 1, 1)
    return x + h


def point_tokens(sd, neighborhood, center, train, prefix="point_encoder.", new_stats=None):
    """point_encoder.py:238-247: tokens + positional embeddings [B,513,384] each."""
    B = neighborhood.shape[0]
    tok = mini_pointnet(sd, neighborhood, train, prefix + "encoder.", new_stats)
    tok = linear(tok, sd[prefix + "reduce_dim.weight"], sd[prefix + "reduce_dim.bias"])
    pos = gelu_erf(linear(center, sd[prefix + "pos_embed.0.weight"], sd[prefix + "pos_embed.0.bias"]))
    pos = linear(pos, sd[prefix + "pos_embed.2.weight"], sd[prefix + "pos_embed.2.bias"])
    x = torch.cat([sd[prefix + "cls_token"].expand(B, -1, -1), tok], dim=1)
    pos = torch.cat([sd[prefix + "cls_pos"].expand(B, -1, -1), pos], dim=1)
    return x, pos


def point_transformer(sd, pc, fps_start, train=False, dp_masks=None, prefix="point_encoder.",
                      num_group=512, group_size=32, heads=6, depth=12, new_stats=None, aux=None):
    """point_encoder.py:234-257 PointTransformer.forward -> [B,768].
    pc [B,N,3] tensor; fps_start [B] (injected RNG); dp_masks[l] = (dp1, dp2) or None."""
    pc_np = pc.detach().numpy()
    cidx = fps(pc_np, num_group, np.asarray(fps_start))
    nbr, nb, ce = group(pc_np, cidx, group_size)
    if aux is not None:
        aux.update(center_idx=cidx, nbr_idx=nbr, neighborhood=nb, center=ce)
    x, pos = point_tokens(sd, torch.from_numpy(nb), torch.from_numpy(ce), train, prefix, new_stats)
    for l in range(depth):                                  # point_encoder.py:102-103: block(x + pos)
        dp = dp_masks[l] if dp_masks is not None else (None, None)
        x = vit_block(sd, f"{prefix}blocks.blocks.{l}.", x + pos, heads, dp[0], dp[1])
    x = layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])
    return torch.cat([x[:, 0], x[:, 1:].max(1)[0]], dim=-1)  # point_encoder.py:251


# ------------------------------------------------------------------------------------------------
# part-segmentation encoder/decoder (models/pointbert/point_encoder.py:347-420, pointnet2_utils.py:297-467)
# ------------------------------------------------------------------------------------------------
def group_norm_rows(x, w, b, groups, eps=1e-5):
    """nn.GroupNorm(groups, C) on [B, C, *]: statistics per (sample, channel group) over the group's channels and
    all trailing positions."""
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1)
    mu = xg.mean(-1, keepdim=True)
    var = ((xg - mu) ** 2).mean(-1, keepdim=True)
    xn = ((xg - mu) / torch.sqrt(var + eps)).reshape(x.shape)
    shape = (1, C) + (1,) * (x.dim() - 2)
    return xn * w.view(shape) + b.view(shape)


def feature_propagation(sd, p, xyz1, xyz2, points1, points2, train, new_stats):
    """PointNetFeaturePropagation.forward (pointnet2_utils.py:310-368) in row layout.
    xyz1 [B,N,3] numpy (targets), xyz2 [B,S,3] numpy (sources), points1 [B,N,D1] tensor|None, points2 [B,S,D2] -> [B,N,C]."""
    B, N, _ = xyz1.shape
    idx, w = three_nn(xyz1, xyz2)
    gathered = points2[torch.arange(B)[:, None, None], torch.from_numpy(idx)]            # [B,N,3,D2]
    interp = (gathered * torch.from_numpy(w)[..., None]).sum(dim=2)
    x = interp if points1 is None else torch.cat([points1, interp], dim=-1)
    x = x.reshape(B * N, -1)
    for j in range(2):
        wgt = sd[f"{p}mlp_convs.{j}.weight"]
        x = linear(x, wgt.reshape(wgt.shape[0], -1), sd[f"{p}mlp_convs.{j}.bias"])
        x = torch.relu(batch_norm_rows(x, sd, f"{p}mlp_bns.{j}.", train, new_stats=new_stats))
    return x.reshape(B, N, -1)


def _graph_feature(coor_q, x_q, coor_k, x_k, k):
    """DGCNN_Propagation.get_graph_feature (pointnet2_utils.py:392-431): cat(x_k[nn] - x_q, x_q) -> [B,Nq,k,2C]."""
    B = x_q.shape[0]
    idx, _ = knn(coor_k, coor_q, k)                                                       # [B,Nq,k]
    nb = x_k[torch.arange(B)[:, None, None], torch.from_numpy(idx)]                       # [B,Nq,k,C]
    xq = x_q[:, :, None, :].expand(-1, -1, k, -1)
    return torch.cat([nb - xq, xq], dim=-1)


def dgcnn_propagation(sd, p, coor, f, coor_q, f_q, k=4):
    """DGCNN_Propagation.forward (pointnet2_utils.py:433-467).  coor [B,S,3] numpy, f [B,S,C], coor_q [B,Nq,3], f_q [B,Nq,C]."""
    for layer, (ck, fk) in (("layer1", (coor, f)), ("layer2", (coor_q, None))):
        fk = f_q if fk is None else fk
        g = _graph_feature(coor_q, f_q, ck, fk, k)                                         # [B,Nq,k,2C]
        w = sd[f"{p}{layer}.0.weight"]
        y = g @ w.reshape(w.shape[0], -1).t()                                              # [B,Nq,k,Cout]
        y = group_norm_rows(y.permute(0, 3, 1, 2), sd[f"{p}{layer}.1.weight"], sd[f"{p}{layer}.1.bias"], 4)
        y = torch.where(y >= 0, y, 0.2 * y)
        f_q = y.max(dim=-1)[0].permute(0, 2, 1)                                            # [B,Nq,Cout]
    return f_q


def partseg_features(sd, pc, onehot, fps_starts, train=False, dp_masks=None, drop_mask=None, prefix="point_encoder.",
                     new_stats=None):
    """PointTransformer_partseg.forward (point_encoder.py:347-420) -> [B,N,128].
    fps_starts = (tokenizer [B], level-1 [B], level-2 [B]); drop_mask [B,N,128] multiplicative Dropout(0.5) factors."""
    pc_np = pc.detach().numpy()
    B, N, _ = pc_np.shape
    cidx = fps(pc_np, 512, np.asarray(fps_starts[0]))
    _, nb, ce = group(pc_np, cidx, 32)
    x, pos = point_tokens(sd, torch.from_numpy(nb), torch.from_numpy(ce), train, prefix, new_stats)
    feats = []
    for l in range(12):
        dp = dp_masks[l] if dp_masks is not None else (None, None)
        x = vit_block(sd, f"{prefix}blocks.blocks.{l}.", x + pos, 6, dp[0], dp[1])
        if l in (3, 7, 11):
            feats.append(layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])[:, 1:])
    c1 = np.take_along_axis(pc_np, fps(pc_np, 512, np.asarray(fps_starts[1]))[:, :, None], axis=1)
    c2 = np.take_along_axis(pc_np, fps(pc_np, 256, np.asarray(fps_starts[2]))[:, :, None], axis=1)
    f0 = torch.cat([onehot[:, None, :].expand(-1, N, -1), pc], dim=-1)                     # [B,N,19]
    f2 = feature_propagation(sd, prefix + "propagation_2.", c2, ce, torch.from_numpy(c2), feats[1], train, new_stats)
    f1 = feature_propagation(sd, prefix + "propagation_1.", c1, ce, torch.from_numpy(c1), feats[0], train, new_stats)
    f2 = dgcnn_propagation(sd, prefix + "dgcnn_pro_2.", ce, feats[2], c2, f2)
    f1 = dgcnn_propagation(sd, prefix + "dgcnn_pro_1.", c2, f2, c1, f1)
    f0 = feature_propagation(sd, prefix + "propagation_0.", pc_np, c1, f0, f1, train, new_stats)
    w = sd[prefix + "conv1.weight"]
    y = linear(f0.reshape(B * N, -1), w.reshape(w.shape[0], -1), sd[prefix + "conv1.bias"])
    y = torch.relu(batch_norm_rows(y, sd, prefix + "bn1.", train, new_stats=new_stats)).reshape(B, N, -1)
    return y if drop_mask is None else y * drop_mask


def partseg_logits(sd, pc, onehot, fps_starts, embedding, name_lengths, eot_pos, position="middle", train=False,
                   dp_masks=None, drop_mask=None, new_stats=None):
    """ULIP_WITH_IMAGE.forward, task='partseg' (ULIP_models.py:250-283) -> per-point logits [B,N,C]."""
    feat = partseg_features(sd, pc, onehot, fps_starts, train, dp_masks, drop_mask, new_stats=new_stats)
    pc_embed = feat @ sd["pc_projection"]
    prompts = splice_prompts(embedding, sd["prompt_learner.learnable_tokens"], name_lengths, position)
    te = text_tower(sd, prompts, eot_pos)
    te = te / te.norm(dim=-1, keepdim=True)
    return sd["logit_scale"].exp() * pc_embed @ te.t()


# ------------------------------------------------------------------------------------------------
# PointNet2-MSG encoder (models/pointnet2/pointnet2.py:40-73, pointnet2_utils.py:161-266)
# ------------------------------------------------------------------------------------------------
PN2_MSG = dict(
    sa1=dict(npoint=512, radii=[0.1, 0.2, 0.4], nsample=[16, 32, 128]),
    sa2=dict(npoint=128, radii=[0.2, 0.4, 0.8], nsample=[32, 64, 128]))


def _conv_bn_relu_stack(sd, x, conv_fmt, bn_fmt, n_layers, train, new_stats):
    """x [rows, C]: 1x1 Conv2d == per-row linear; BatchNorm2d over (B, K, S) == over all rows."""
    for j in range(n_layers):
        w = sd[conv_fmt.format(j) + "weight"]
        x = linear(x, w.reshape(w.shape[0], -1), sd[conv_fmt.format(j) + "bias"])
        x = torch.relu(batch_norm_rows(x, sd, bn_fmt.format(j), train, new_stats=new_stats))
    return x


def _sa_msg(sd, p, cfg, xyz, feats, start, train, new_stats):
    """PointNetSetAbstractionMsg.forward (pointnet2_utils.py:228-266).  xyz [B,N,3] numpy, feats [B,N,D] tensor|None
    -> new_xyz [B,S,3] numpy, new_feats [B,S,sum C] tensor."""
    B, N, _ = xyz.shape
    S = cfg["npoint"]
    cidx = fps(xyz, S, start)
    new_xyz = np.take_along_axis(xyz, cidx[:, :, None], axis=1)
    outs = []
    for i, (r, K) in enumerate(zip(cfg["radii"], cfg["nsample"])):
        gidx = ball_query(xyz, new_xyz, r, K)                                    # [B,S,K]
        g = torch.from_numpy(np.take_along_axis(xyz[:, None], gidx[..., None], axis=2) if False else
                             xyz[np.arange(B)[:, None, None], gidx])             # [B,S,K,3]
        g = g - torch.from_numpy(new_xyz)[:, :, None, :]
        if feats is not None:
            gf = feats[torch.arange(B)[:, None, None], torch.from_numpy(gidx)]   # [B,S,K,D]
            g = torch.cat([gf, g], dim=-1)                                       # features first (:250)
        n_layers = sum(1 for k in sd if k.startswith(f"{p}conv_blocks.{i}.") and k.endswith("weight"))
        y = _conv_bn_relu_stack(sd, g.reshape(B * S * K, -1), p + f"conv_blocks.{i}." + "{}.", p + f"bn_blocks.{i}." + "{}.",
                                n_layers, train, new_stats)
        outs.append(y.reshape(B, S, K, -1).max(dim=2)[0])
    return new_xyz, torch.cat(outs, dim=-1)


def pointnet2_msg(sd, pc, fps_starts, train=False, drop_masks=None, prefix="point_encoder.", new_stats=None):
    """Pointnet2_Msg.forward (pointnet2.py:56-73) -> [B,256].  fps_starts = (start level 1 [B], start level 2 [B]);
    drop_masks = (mask1 [B,512], mask2 [B,256]) multiplicative Dropout factors (0 or 1/(1-p)) or None."""
    xyz = pc.detach().numpy()
    B = xyz.shape[0]
    l1_xyz, l1 = _sa_msg(sd, prefix + "sa1.", PN2_MSG["sa1"], xyz, None, np.asarray(fps_starts[0]), train, new_stats)
    l2_xyz, l2 = _sa_msg(sd, prefix + "sa2.", PN2_MSG["sa2"], l1_xyz, l1, np.asarray(fps_starts[1]), train, new_stats)
    g = torch.cat([torch.from_numpy(l2_xyz), l2], dim=-1)                         # group_all: xyz first (:152-157)
    y = _conv_bn_relu_stack(sd, g.reshape(B * 128, -1), prefix + "sa3.mlp_convs.{}.", prefix + "sa3.mlp_bns.{}.", 3, train,
                            new_stats)
    x = y.reshape(B, 128, -1).max(dim=1)[0]                                       # [B,1024]
    x = torch.relu(batch_norm_rows(linear(x, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"]), sd, prefix + "bn1.", train,
                                   new_stats=new_stats))
    if drop_masks is not None:
        x = x * drop_masks[0]
    x = torch.relu(batch_norm_rows(linear(x, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"]), sd, prefix + "bn2.", train,
                                   new_stats=new_stats))
    if drop_masks is not None:
        x = x * drop_masks[1]
    return x


PN2_SSG = dict(   # pointnet2.py:11-12
    sa1=dict(npoint=512, radius=0.2, nsample=32),
    sa2=dict(npoint=128, radius=0.4, nsample=64))


def _sa_ssg(sd, p, cfg, xyz, feats, start, train, new_stats):
    """PointNetSetAbstraction.forward with sample_and_group (pointnet2_utils.py:110-138,177-206): one radius, channels
    = [centred xyz | features] (xyz FIRST, :132 -- the MSG module puts the features first).  xyz [B,N,3] numpy,
    feats [B,N,D] tensor | None -> new_xyz [B,S,3] numpy, new_feats [B,S,C] tensor."""
    B, N, _ = xyz.shape
    S, K = cfg["npoint"], cfg["nsample"]
    cidx = fps(xyz, S, start)
    new_xyz = np.take_along_axis(xyz, cidx[:, :, None], axis=1)
    gidx = ball_query(xyz, new_xyz, cfg["radius"], K)                             # [B,S,K]
    g = torch.from_numpy(xyz[np.arange(B)[:, None, None], gidx]) - torch.from_numpy(new_xyz)[:, :, None, :]
    if feats is not None:
        g = torch.cat([g, feats[torch.arange(B)[:, None, None], torch.from_numpy(gidx)]], dim=-1)
    n_layers = sum(1 for k in sd if k.startswith(p + "mlp_convs.") and k.endswith("weight"))
    y = _conv_bn_relu_stack(sd, g.reshape(B * S * K, -1), p + "mlp_convs.{}.", p + "mlp_bns.{}.", n_layers, train, new_stats)
    return new_xyz, y.reshape(B, S, K, -1).max(dim=2)[0]


def pointnet2_ssg(sd, pc, fps_starts, train=False, drop_masks=None, prefix="point_encoder.", new_stats=None):
    """Pointnet2_Ssg.forward (pointnet2.py:22-38) -> [B,256]: two single-scale set abstractions, a group_all one
    (xyz first, :152-157), and the same FC head as the MSG encoder with Dropout(0.4) twice."""
    xyz = pc.detach().numpy()
    B = xyz.shape[0]
    l1_xyz, l1 = _sa_ssg(sd, prefix + "sa1.", PN2_SSG["sa1"], xyz, None, np.asarray(fps_starts[0]), train, new_stats)
    l2_xyz, l2 = _sa_ssg(sd, prefix + "sa2.", PN2_SSG["sa2"], l1_xyz, l1, np.asarray(fps_starts[1]), train, new_stats)
    g = torch.cat([torch.from_numpy(l2_xyz), l2], dim=-1)
    y = _conv_bn_relu_stack(sd, g.reshape(B * 128, -1), prefix + "sa3.mlp_convs.{}.", prefix + "sa3.mlp_bns.{}.", 3, train,
                            new_stats)
    x = y.reshape(B, 128, -1).max(dim=1)[0]                                       # [B,1024]
    x = torch.relu(batch_norm_rows(linear(x, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"]), sd, prefix + "bn1.", train,
                                   new_stats=new_stats))
    if drop_masks is not None:
        x = x * drop_masks[0]
    x = torch.relu(batch_norm_rows(linear(x, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"]), sd, prefix + "bn2.", train,
                                   new_stats=new_stats))
    if drop_masks is not None:
        x = x * drop_masks[1]
    return x


POINTMLP = dict(points=1024, k_neighbors=[24] * 4, reducers=[2] * 4, pre_blocks=[2] * 4, pos_blocks=[2] * 4)        # pointMLP.py:359-363
POINTMLP_ELITE = dict(points=1024, k_neighbors=[24] * 4, reducers=[2] * 4, pre_blocks=[1, 1, 2, 1], pos_blocks=[1, 1, 2, 1])  # :366-370


def _res_block(sd, p, x, train, new_stats):
    """ConvBNReLURes1D.forward (pointMLP.py:188-221, groups=1, bias=False): act(net2(net1(x)) + x) on rows [R, C]."""
    w1, w2 = sd[p + "net1.0.weight"], sd[p + "net2.0.weight"]
    h = torch.relu(batch_norm_rows(linear(x, w1.reshape(w1.shape[0], -1)), sd, p + "net1.1.", train, new_stats=new_stats))
    z = batch_norm_rows(linear(h, w2.reshape(w2.shape[0], -1)), sd, p + "net2.1.", train, new_stats=new_stats)
    return torch.relu(z + x)


def pointmlp(sd, pc, fps_starts, train=False, drop_masks=None, prefix="point_encoder.", new_stats=None, cfg=None):
    """Model.forward of pointMLP() / pointMLPElite() (pointMLP.py:320-334) -> [B,256]; widths come from the weights.  pc [B,N,3]; fps_starts = the four start indices
    furthest_point_sample draws (:77, one [B] vector per stage); drop_masks = (m1 [B,512], m2 [B,256]) multiplicative
    Dropout(0.5) factors of the classifier (:307-316) or None."""
    xyz = np.ascontiguousarray(pc.detach().numpy(), np.float32)
    B, N, _ = xyz.shape
    w = sd[prefix + "embedding.net.0.weight"]
    x = torch.relu(batch_norm_rows(linear(torch.from_numpy(xyz).reshape(B * N, 3), w.reshape(w.shape[0], -1)), sd,
                                   prefix + "embedding.net.1.", train, new_stats=new_stats)).reshape(B, N, -1)   # :324
    cfg = cfg or POINTMLP
    anchors = cfg["points"]
    bi = torch.arange(B)[:, None]
    for i in range(len(cfg["reducers"])):
        anchors //= cfg["reducers"][i]
        S, k = anchors, cfg["k_neighbors"][i]
        # LocalGrouper.forward (:152-181), normalize="anchor", use_xyz=False
        cidx = fps(xyz, S, np.asarray(fps_starts[i]))                                    # :157
        new_xyz = np.take_along_axis(xyz, cidx[:, :, None], axis=1)                      # :158
        anchor = x[bi, torch.from_numpy(cidx)]                                           # :159 [B,S,d]
        # knn_point (:109-121): topk of the matmul-form distances; the set of k neighbours is all that matters below
        d2 = torch.from_numpy(square_distance(new_xyz, xyz))
        nidx = torch.topk(d2, k, dim=-1, largest=False)[1]                               # [B,S,k]
        grouped = x[bi[:, :, None], nidx]                                                # :163 [B,S,k,d]
        diff = grouped - anchor[:, :, None, :]                                           # :170-173 mean = anchor
        std = torch.std(diff.reshape(B, -1), dim=-1, keepdim=True)[:, :, None, None]     # :174 one scalar per cloud
        g = diff / (std + 1e-5)                                                          # :175
        g = sd[f"{prefix}local_grouper_list.{i}.affine_alpha"] * g + sd[f"{prefix}local_grouper_list.{i}.affine_beta"]
        g = torch.cat([g, anchor[:, :, None, :].expand(-1, -1, k, -1)], dim=-1)          # :178 [B,S,k,2d]
        # PreExtraction.forward (:243-253): rows (b, s, k)
        pp = f"{prefix}pre_blocks_list.{i}."
        wt = sd[pp + "transfer.net.0.weight"]
        y = torch.relu(batch_norm_rows(linear(g.reshape(B * S * k, -1), wt.reshape(wt.shape[0], -1)), sd,
                                       pp + "transfer.net.1.", train, new_stats=new_stats))
        for j in range(cfg["pre_blocks"][i]):
            y = _res_block(sd, f"{pp}operation.{j}.", y, train, new_stats)
        y = y.reshape(B * S, k, -1).max(dim=1)[0]                                        # :251 adaptive_max_pool1d
        # PosExtraction.forward (:272-273): rows (b, s)
        for j in range(cfg["pos_blocks"][i]):
            y = _res_block(sd, f"{prefix}pos_blocks_list.{i}.operation.{j}.", y, train, new_stats)
        x, xyz = y.reshape(B, S, -1), np.ascontiguousarray(new_xyz)
    f = x.max(dim=1)[0]                                                                  # :332 [B,1024]
    c = prefix + "classifier."
    f = torch.relu(batch_norm_rows(linear(f, sd[c + "0.weight"], sd[c + "0.bias"]), sd, c + "1.", train, new_stats=new_stats))
    if drop_masks is not None:
        f = f * drop_masks[0]
    f = torch.relu(batch_norm_rows(linear(f, sd[c + "4.weight"], sd[c + "4.bias"]), sd, c + "5.", train, new_stats=new_stats))
    if drop_masks is not None:
        f = f * drop_masks[1]
    return f


# ------------------------------------------------------------------------------------------------
# text branch
# ------------------------------------------------------------------------------------------------
def splice_prompts(embedding, learnable_tokens, name_lengths, position="middle"):
    """ULIP_models.py:104-151 PromptLearner.forward.  embedding [C,77,512] frozen,
    learnable_tokens [n_ctx,512] -> prompts [C,77,512]."""
    C = embedding.shape[0]
    n_ctx = learnable_tokens.shape[0]
    prefix, suffix = embedding[:, :1], embedding[:, 1 + n_ctx:]
    if position == "end":
        return torch.cat([prefix, learnable_tokens.unsqueeze(0).expand(C, -1, -1), suffix], dim=1)
    rows = []
    half = n_ctx // 2
    for i in range(C):
        L = name_lengths[i]
        if position == "middle":
            parts = [prefix[i], learnable_tokens[:half], suffix[i, :L], learnable_tokens[half:], suffix[i, L:]]
        elif position == "front":
            parts = [prefix[i], suffix[i, :L], learnable_tokens, suffix[i, L:]]
        else:
            raise ValueError(position)
        rows.append(torch.cat(parts, dim=0))
    return torch.stack(rows, dim=0)


def clip_attention(x, w_in, b_in, w_out, b_out, heads):
    """nn.MultiheadAttention(512, 8) as used by ULIP_models.py:38,49-51 with the additive causal
    mask of :224-230.  x [C,L,D] (batch-first here; the reference permutes to LND, :214-216).
    q is scaled by hd^-0.5 BEFORE the product (torch MHA), unlike the ViT blocks."""
    C, L, D = x.shape
    hd = D // heads
    qkv = linear(x, w_in, b_in).reshape(C, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    a = q @ k.transpose(-2, -1)
    mask = torch.full((L, L), float("-inf")).triu(1)
    a = torch.softmax(a + mask, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(C, L, D)
    return linear(o, w_out, b_out)


def text_tower(sd, prompts, eot_pos, heads=8, layers=12):
    """ULIP_models.py:203-222 encode_text -> [C,512] (before L2 normalisation)."""
    x = prompts + sd["positional_embedding"].unsqueeze(0)
    for i in range(layers):                                     # ULIP_models.py:53-56
        p = f"transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        x = x + clip_attention(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                               sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads)
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        h = quick_gelu(linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
        x = x + linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    x = x[torch.arange(x.shape[0]), torch.as_tensor(eot_pos)]
    return x @ sd["text_projection"]


# ------------------------------------------------------------------------------------------------
# whole model / training step
# ------------------------------------------------------------------------------------------------
def ulip_logits(sd, pc, fps_start, embedding, name_lengths, eot_pos, position="middle",
                train=False, dp_masks=None, new_stats=None, aux=None):
    """ULIP_models.py:250-283 ULIP_WITH_IMAGE.forward (task='cls') -> logits [B,C]."""
    pc_feat = point_transformer(sd, pc, fps_start, train, dp_masks, new_stats=new_stats, aux=aux)
    pc_embed = pc_feat @ sd["pc_projection"]                    # :257 (not normalised)
    prompts = splice_prompts(embedding, sd["prompt_learner.learnable_tokens"], name_lengths, position)
    te = text_tower(sd, prompts, eot_pos)
    te = te / te.norm(dim=-1, keepdim=True)                     # :277
    if aux is not None:
        aux.update(pc_feat=pc_feat.detach(), pc_embed=pc_embed.detach(), text_embed=te.detach())
    return sd["logit_scale"].exp() * pc_embed @ te.t()          # :279-281


def cross_entropy_ls(logits, labels, smoothing):
    """nn.CrossEntropyLoss(label_smoothing=s), mean reduction (main_cls.py:52):
    (1-s)*nll + s*mean_c(-log p_c)."""
    logp = logits - torch.logsumexp(logits, dim=1, keepdim=True)
    nll = -logp[torch.arange(logits.shape[0]), labels]
    return ((1 - smoothing) * nll + smoothing * (-logp.mean(dim=1))).mean()


TRAINABLE_TIERS = {   # ULIP_models.py:461-470 (cumulative)
    1: ["norm2.weight", "norm2.bias", "mlp.fc2.weight", "mlp.fc2.bias"],
    2: ["norm1.weight", "norm1.bias", "mlp.fc1.weight", "mlp.fc1.bias"],
    3: ["attn.qkv.weight", "attn.proj.weight", "attn.proj.bias"],
}


def trainable_keys(head_type):
    keys = ["prompt_learner.learnable_tokens"]
    for t in (1, 2, 3):
        if head_type >= t:
            keys += ["point_encoder.blocks.blocks.11." + k for k in TRAINABLE_TIERS[t]]
    return keys


def adamw_update(p, g, state, lr, betas=(0.9, 0.98), eps=1e-8, wd=0.1):
    """torch.optim.AdamW single-tensor step (main_cls.py:58-60 hyper-parameters)."""
    state["step"] = state.get("step", 0) + 1
    m = state.setdefault("m", torch.zeros_like(p))
    v = state.setdefault("v", torch.zeros_like(p))
    t = state["step"]
    p = p * (1 - lr * wd)
    m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    denom = v.sqrt() / math.sqrt(1 - betas[1] ** t) + eps
    return p - (lr / (1 - betas[0] ** t)) * m / denom


def train_step(sd, pc, labels, fps_start, embedding, name_lengths, eot_pos, head_type=0,
               position="middle", smoothing=0.2, lr=3e-3, dp_masks=None, opt_state=None,
               train=True):
    """One iteration of main_cls.py:179-214 (zero_grad, forward, CE, backward, AdamW, clamp).
    Returns dict(logits, loss, grads{key}, new_params{key}, new_stats{key})."""
    sd = dict(sd)
    keys = trainable_keys(head_type)
    for k in keys:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    new_stats = {}
    logits = ulip_logits(sd, pc, fps_start, embedding, name_lengths, eot_pos, position,
                         train=train, dp_masks=dp_masks, new_stats=new_stats)
    loss = cross_entropy_ls(logits, labels, smoothing)
    grads = torch.autograd.grad(loss, [sd[k] for k in keys])
    opt_state = {} if opt_state is None else opt_state
    new_params = {}
    for k, g in zip(keys, grads):
        new_params[k] = adamw_update(sd[k].detach(), g, opt_state.setdefault(k, {}), lr)
    return dict(logits=logits.detach(), loss=loss.detach(), grads=dict(zip(keys, grads)),
                new_params=new_params, new_stats=new_stats, opt_state=opt_state)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0.0):
    """utils/utils.py:253-264: linear warm-up then half-cosine, one value per iteration."""
    warm = warmup_epochs * niter_per_ep
    head = np.linspace(start_warmup_value, base_value, warm) if warmup_epochs > 0 else np.array([])
    it = np.arange(epochs * niter_per_ep - warm)
    tail = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
    return np.concatenate((head, tail))
