"""Deterministic, name-keyed synthetic weights for the ULIP + PointBERT state dict.

No checkpoints exist in this environment (reference models/ULIP_models.py:474-485 loads
./data/pretrained_models/pointbert.pt and ./data/initialize_models/slip_base_100ep.pt), so both
the product and the oracle side of every test -- and bench.py -- draw the SAME values from this
generator: every tensor is seeded by crc32(key) and produced with numpy's PCG64, which is
identical on every machine.  Keys and shapes are the reference's.
"""
import zlib
from collections import OrderedDict

import numpy as np

TRANS_DIM, DEPTH, HEADS, ENC_DIM = 384, 12, 6, 256          # PointTransformer_8192point.yaml:15-25
TXT_WIDTH, TXT_LAYERS, TXT_HEADS, CTX_LEN, EMBED = 512, 12, 8, 77, 512   # ULIP_models.py:456-459
VOCAB = 49408


def pointbert_spec(prefix="point_encoder."):
    """(key, shape) list of PointTransformer.state_dict() (point_encoder.py:113-152)."""
    p, s = prefix, []
    s += [(p + "cls_token", (1, 1, TRANS_DIM)), (p + "cls_pos", (1, 1, TRANS_DIM))]
    e = p + "encoder."
    s += [(e + "first_conv.0.weight", (128, 3, 1)), (e + "first_conv.0.bias", (128,))]
    s += [(e + "first_conv.1." + k, (128,)) for k in ("weight", "bias", "running_mean", "running_var")]
    s += [(e + "first_conv.1.num_batches_tracked", ())]
    s += [(e + "first_conv.3.weight", (256, 128, 1)), (e + "first_conv.3.bias", (256,))]
    s += [(e + "second_conv.0.weight", (512, 512, 1)), (e + "second_conv.0.bias", (512,))]
    s += [(e + "second_conv.1." + k, (512,)) for k in ("weight", "bias", "running_mean", "running_var")]
    s += [(e + "second_conv.1.num_batches_tracked", ())]
    s += [(e + "second_conv.3.weight", (ENC_DIM, 512, 1)), (e + "second_conv.3.bias", (ENC_DIM,))]
    s += [(p + "reduce_dim.weight", (TRANS_DIM, ENC_DIM)), (p + "reduce_dim.bias", (TRANS_DIM,))]
    s += [(p + "pos_embed.0.weight", (128, 3)), (p + "pos_embed.0.bias", (128,)),
          (p + "pos_embed.2.weight", (TRANS_DIM, 128)), (p + "pos_embed.2.bias", (TRANS_DIM,))]
    for i in range(DEPTH):
        b = f"{p}blocks.blocks.{i}."
        s += [(b + "norm1.weight", (TRANS_DIM,)), (b + "norm1.bias", (TRANS_DIM,)),
              (b + "norm2.weight", (TRANS_DIM,)), (b + "norm2.bias", (TRANS_DIM,)),
              (b + "mlp.fc1.weight", (4 * TRANS_DIM, TRANS_DIM)), (b + "mlp.fc1.bias", (4 * TRANS_DIM,)),
              (b + "mlp.fc2.weight", (TRANS_DIM, 4 * TRANS_DIM)), (b + "mlp.fc2.bias", (TRANS_DIM,)),
              (b + "attn.qkv.weight", (3 * TRANS_DIM, TRANS_DIM)),
              (b + "attn.proj.weight", (TRANS_DIM, TRANS_DIM)), (b + "attn.proj.bias", (TRANS_DIM,))]
    s += [(p + "norm.weight", (TRANS_DIM,)), (p + "norm.bias", (TRANS_DIM,))]
    return s


def ulip_spec(pc_feat_dims=768, with_token_embedding=True):
    """(key, shape) list of ULIP_WITH_IMAGE.state_dict() minus the point encoder
    (ULIP_models.py:154-201)."""
    s = [("positional_embedding", (CTX_LEN, TXT_WIDTH)), ("text_projection", (TXT_WIDTH, EMBED)),
         ("pc_projection", (pc_feat_dims, EMBED)), ("logit_scale", ())]
    for i in range(TXT_LAYERS):
        b = f"transformer.resblocks.{i}."
        s += [(b + "attn.in_proj_weight", (3 * TXT_WIDTH, TXT_WIDTH)), (b + "attn.in_proj_bias", (3 * TXT_WIDTH,)),
              (b + "attn.out_proj.weight", (TXT_WIDTH, TXT_WIDTH)), (b + "attn.out_proj.bias", (TXT_WIDTH,)),
              (b + "ln_1.weight", (TXT_WIDTH,)), (b + "ln_1.bias", (TXT_WIDTH,)),
              (b + "mlp.c_fc.weight", (4 * TXT_WIDTH, TXT_WIDTH)), (b + "mlp.c_fc.bias", (4 * TXT_WIDTH,)),
              (b + "mlp.c_proj.weight", (TXT_WIDTH, 4 * TXT_WIDTH)), (b + "mlp.c_proj.bias", (TXT_WIDTH,)),
              (b + "ln_2.weight", (TXT_WIDTH,)), (b + "ln_2.bias", (TXT_WIDTH,))]
    if with_token_embedding:
        s += [("token_embedding.weight", (VOCAB, TXT_WIDTH))]
    s += [("ln_final.weight", (TXT_WIDTH,)), ("ln_final.bias", (TXT_WIDTH,))]
    s += [("prompt_learner.learnable_tokens", (32, TXT_WIDTH))]
    return s


def partseg_decoder_spec(prefix="point_encoder."):
    """(key, shape) list of the decoder that PointTransformer_partseg adds to the PointBERT backbone
    (point_encoder.py:299-310)."""
    p, s = prefix, []

    def bn(q, c):
        return [(q + k, (c,)) for k in ("weight", "bias", "running_mean", "running_var")] + [(q + "num_batches_tracked", ())]
    for name, cin in (("propagation_2", 387), ("propagation_1", 387), ("propagation_0", 403)):
        s += [(f"{p}{name}.mlp_convs.0.weight", (1536, cin, 1)), (f"{p}{name}.mlp_convs.0.bias", (1536,)),
              (f"{p}{name}.mlp_convs.1.weight", (384, 1536, 1)), (f"{p}{name}.mlp_convs.1.bias", (384,))]
        s += bn(f"{p}{name}.mlp_bns.0.", 1536) + bn(f"{p}{name}.mlp_bns.1.", 384)
    for name in ("dgcnn_pro_1", "dgcnn_pro_2"):
        s += [(f"{p}{name}.layer1.0.weight", (512, 768, 1, 1)), (f"{p}{name}.layer1.1.weight", (512,)), (f"{p}{name}.layer1.1.bias", (512,)),
              (f"{p}{name}.layer2.0.weight", (384, 1024, 1, 1)), (f"{p}{name}.layer2.1.weight", (384,)), (f"{p}{name}.layer2.1.bias", (384,))]
    s += [(p + "conv1.weight", (128, 384, 1)), (p + "conv1.bias", (128,))] + bn(p + "bn1.", 128)
    s += [(p + "conv2.weight", (40, 128, 1)), (p + "conv2.bias", (40,))]
    return s


def ulip_partseg_state_dict(seed=0, with_token_embedding=False, as_torch=True):
    return synth_state_dict(ulip_spec(128, with_token_embedding) + pointbert_spec() + partseg_decoder_spec(), seed, as_torch)


PN2_MSG = dict(   # models/pointnet2/pointnet2.py:44-46
    sa1=dict(npoint=512, radii=[0.1, 0.2, 0.4], nsample=[16, 32, 128], in_channel=0,
             mlps=[[32, 32, 64], [64, 64, 128], [64, 96, 128]]),
    sa2=dict(npoint=128, radii=[0.2, 0.4, 0.8], nsample=[32, 64, 128], in_channel=320,
             mlps=[[64, 64, 128], [128, 128, 256], [128, 128, 256]]),
    sa3=dict(in_channel=640 + 3, mlp=[256, 512, 1024]))


def pointnet2_msg_spec(prefix="point_encoder."):
    """(key, shape) list of Pointnet2_Msg.state_dict() (models/pointnet2/pointnet2.py:40-55)."""
    def bn(p, c):
        return [(p + k, (c,)) for k in ("weight", "bias", "running_mean", "running_var")] + [(p + "num_batches_tracked", ())]
    s = []
    for name in ("sa1", "sa2"):
        cfg = PN2_MSG[name]
        convs, bns = [], []
        for i, mlp in enumerate(cfg["mlps"]):
            last = cfg["in_channel"] + 3
            for j, out in enumerate(mlp):
                convs += [(f"{prefix}{name}.conv_blocks.{i}.{j}.weight", (out, last, 1, 1)),
                          (f"{prefix}{name}.conv_blocks.{i}.{j}.bias", (out,))]
                bns += bn(f"{prefix}{name}.bn_blocks.{i}.{j}.", out)
                last = out
        s += convs + bns
    last = PN2_MSG["sa3"]["in_channel"]
    convs, bns = [], []
    for j, out in enumerate(PN2_MSG["sa3"]["mlp"]):
        convs += [(f"{prefix}sa3.mlp_convs.{j}.weight", (out, last, 1, 1)), (f"{prefix}sa3.mlp_convs.{j}.bias", (out,))]
        bns += bn(f"{prefix}sa3.mlp_bns.{j}.", out)
        last = out
    s += convs + bns
    s += [(prefix + "fc1.weight", (512, 1024)), (prefix + "fc1.bias", (512,))] + bn(prefix + "bn1.", 512)
    s += [(prefix + "fc2.weight", (256, 512)), (prefix + "fc2.bias", (256,))] + bn(prefix + "bn2.", 256)
    return s


PN2_SSG = dict(   # models/pointnet2/pointnet2.py:11-13
    sa1=dict(in_channel=3, mlp=[64, 64, 128]), sa2=dict(in_channel=128 + 3, mlp=[128, 128, 256]),
    sa3=dict(in_channel=256 + 3, mlp=[256, 512, 1024]))


def pointnet2_ssg_spec(prefix="point_encoder."):
    """(key, shape) list of Pointnet2_Ssg.state_dict() (models/pointnet2/pointnet2.py:6-20)."""
    def bn(p, c):
        return [(p + k, (c,)) for k in ("weight", "bias", "running_mean", "running_var")] + [(p + "num_batches_tracked", ())]
    s = []
    for name in ("sa1", "sa2", "sa3"):
        last = PN2_SSG[name]["in_channel"]
        convs, bns = [], []
        for j, out in enumerate(PN2_SSG[name]["mlp"]):
            convs += [(f"{prefix}{name}.mlp_convs.{j}.weight", (out, last, 1, 1)), (f"{prefix}{name}.mlp_convs.{j}.bias", (out,))]
            bns += bn(f"{prefix}{name}.mlp_bns.{j}.", out)
            last = out
        s += convs + bns
    s += [(prefix + "fc1.weight", (512, 1024)), (prefix + "fc1.bias", (512,))] + bn(prefix + "bn1.", 512)
    s += [(prefix + "fc2.weight", (256, 512)), (prefix + "fc2.bias", (256,))] + bn(prefix + "bn2.", 256)
    return s


POINTMLP = dict(   # models/pointmlp/pointMLP.py:359-363 pointMLP(): the encoder ULIP_PN_MLP builds (ULIP_models.py:399-400)
    points=1024, embed_dim=64, k_neighbors=24, stages=4, pre_blocks=2, pos_blocks=2)


def pointmlp_spec(prefix="point_encoder."):
    """(key, shape) list of pointMLP().state_dict() (models/pointmlp/pointMLP.py:261-318; bias=False convs,
    normalize="anchor", use_xyz=False)."""
    def bn(p, c):
        return [(p + k, (c,)) for k in ("weight", "bias", "running_mean", "running_var")] + [(p + "num_batches_tracked", ())]

    def res(p, c):
        return [(p + "net1.0.weight", (c, c, 1))] + bn(p + "net1.1.", c) + [(p + "net2.0.weight", (c, c, 1))] + bn(p + "net2.1.", c)
    s = [(prefix + "embedding.net.0.weight", (POINTMLP["embed_dim"], 3, 1))] + bn(prefix + "embedding.net.1.", POINTMLP["embed_dim"])
    groupers, pres, poss = [], [], []
    d = POINTMLP["embed_dim"]
    for i in range(POINTMLP["stages"]):
        out = 2 * d
        groupers += [(f"{prefix}local_grouper_list.{i}.affine_alpha", (1, 1, 1, d)), (f"{prefix}local_grouper_list.{i}.affine_beta", (1, 1, 1, d))]
        pres += [(f"{prefix}pre_blocks_list.{i}.transfer.net.0.weight", (out, 2 * d, 1))] + bn(f"{prefix}pre_blocks_list.{i}.transfer.net.1.", out)
        for j in range(POINTMLP["pre_blocks"]):
            pres += res(f"{prefix}pre_blocks_list.{i}.operation.{j}.", out)
        for j in range(POINTMLP["pos_blocks"]):
            poss += res(f"{prefix}pos_blocks_list.{i}.operation.{j}.", out)
        d = out
    s += groupers + pres + poss
    s += [(prefix + "classifier.0.weight", (512, d)), (prefix + "classifier.0.bias", (512,))] + bn(prefix + "classifier.1.", 512)
    s += [(prefix + "classifier.4.weight", (256, 512)), (prefix + "classifier.4.bias", (256,))] + bn(prefix + "classifier.5.", 256)
    return s


def ulip_pn_mlp_state_dict(seed=0, with_token_embedding=False, as_torch=True):
    return synth_state_dict(ulip_spec(256, with_token_embedding) + pointmlp_spec(), seed, as_torch)


def ulip_pn2_ssg_state_dict(seed=0, with_token_embedding=False, as_torch=True):
    return synth_state_dict(ulip_spec(256, with_token_embedding) + pointnet2_ssg_spec(), seed, as_torch)


def ulip_pn2_msg_state_dict(seed=0, with_token_embedding=False, as_torch=True):
    return synth_state_dict(ulip_spec(256, with_token_embedding) + pointnet2_msg_spec(), seed, as_torch)


def _rng(key, seed):
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def synth_tensor(key, shape, seed=0):
    """One synthetic tensor (numpy).  Scales are chosen so activations stay O(1) through the depth
    of both towers (fan-in scaled weights), norm scales hover around 1 and every bias / running
    statistic is non-trivial, so that a kernel that drops a bias, a gamma or an eps is caught."""
    r = _rng(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros((), np.int64)
    if key == "logit_scale":
        return np.asarray(np.log(1 / 0.07), np.float32)
    if leaf == "running_mean":
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return (1.0 + 0.5 * r.random(shape)).astype(np.float32)
    if leaf == "affine_alpha":                 # pointMLP.py:149 (ones at init): a per-channel scale around 1
        return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "affine_beta":
        return (0.02 * r.standard_normal(shape)).astype(np.float32)
    is_norm = any(t in key for t in (".net.1.", ".net1.1.", ".net2.1.", "classifier.1.", "classifier.5.")) \
        or any(t in key for t in (".norm1.", ".norm2.", ".norm.", ".ln_1.", ".ln_2.", "ln_final.",
                                     "first_conv.1.", "second_conv.1.", "mlp_bns", ".bn", "bn_blocks",
                                     "layer1.1.", "layer2.1."))
    if is_norm and leaf == "weight":
        return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    if leaf in ("bias", "in_proj_bias"):
        return (0.02 * r.standard_normal(shape)).astype(np.float32)
    if key.endswith("cls_token") or key.endswith("cls_pos") or key == "token_embedding.weight" \
            or key.endswith("learnable_tokens"):
        return (0.02 * r.standard_normal(shape)).astype(np.float32)
    if key == "positional_embedding":
        return (0.01 * r.standard_normal(shape)).astype(np.float32)
    if key == "prompt_learner.embedding" or key.startswith("token_embedding.row."):      # default nn.Embedding N(0,1) draws
        return r.standard_normal(shape).astype(np.float32)
    if key in ("text_projection", "pc_projection"):
        return ((512 ** -0.5) * r.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return ((fan_in ** -0.5) * r.standard_normal(shape)).astype(np.float32)


def synth_state_dict(spec, seed=0, as_torch=True):
    out = OrderedDict()
    for key, shape in spec:
        a = synth_tensor(key, shape, seed)
        if as_torch:
            import torch
            a = torch.from_numpy(np.ascontiguousarray(a))
        out[key] = a
    return out


def ulip_pointbert_state_dict(seed=0, with_token_embedding=False, n_ctx=32, as_torch=True):
    """Full synthetic state dict for ULIP_PointBERT.  token_embedding.weight (25 M values, used
    only at construction -- SURVEY Q1) is skipped unless asked for."""
    spec = [(k, (n_ctx, TXT_WIDTH)) if k.endswith("learnable_tokens") else (k, s)
            for k, s in ulip_spec(768, with_token_embedding)]
    return synth_state_dict(spec + pointbert_spec(), seed, as_torch)


def synth_prompt_embedding(num_classes, seed=0, as_torch=True):
    """The frozen [C,77,512] token embeddings PromptLearner caches at construction
    (ULIP_models.py:102; SURVEY Q1: N(0,1) draws, neither parameter nor buffer)."""
    a = synth_tensor("prompt_learner.embedding", (num_classes, CTX_LEN, TXT_WIDTH), seed)
    if as_torch:
        import torch
        a = torch.from_numpy(a)
    return a


def synth_prompt_embedding_from_tokens(tokenized_prompts, seed=0, as_torch=True):
    """The same cache with the structure the reference's really has: ULIP_models.py:102 computes it as
    token_embedding(tokenized_prompts), so two positions holding the SAME token id hold the same row (N(0,1) draws of a
    never-initialised nn.Embedding) -- in particular the start token's row is identical in every prompt.  One
    deterministic row per token id."""
    ids = np.asarray(tokenized_prompts)
    out = np.empty(ids.shape + (TXT_WIDTH,), dtype=np.float32)
    for t in np.unique(ids):
        out[ids == t] = synth_tensor(f"token_embedding.row.{int(t)}", (TXT_WIDTH,), seed)
    if as_torch:
        import torch
        out = torch.from_numpy(out)
    return out


def synth_clouds(B, N, seed=1234, duplicates=False):
    """Synthetic unit-sphere clouds + labels + FPS start indices.
    uniform [-1,1]^3, then per-cloud pc_normalize (centre, scale by max radius --
    data/dataset_3d.py:33-38).  duplicates=True resamples every cloud with replacement
    (ShapeNetPart loader behaviour, dataset_3d.py:752)."""
    r = np.random.default_rng([seed, B, N])
    pc = r.random((B, N, 3), dtype=np.float32) * 2 - 1
    if duplicates:
        sel = r.integers(0, N, size=(B, N))
        pc = np.take_along_axis(pc, sel[:, :, None], axis=1)
    pc = pc - pc.mean(axis=1, keepdims=True)
    pc = pc / np.sqrt((pc ** 2).sum(-1)).max(axis=1)[:, None, None]
    start = r.integers(0, N, size=(B,)).astype(np.int64)
    return np.ascontiguousarray(pc.astype(np.float32)), start


def checkpoint_like(sd, seed=0, weight_gain=3.0, outliers=4, outlier_gain=(5.0, 10.0)):
    """A state dict with the MAGNITUDES of a trained ULIP / SLIP checkpoint instead of the std-0.02 synthetic ones (VERDICT r4
    weak #10; no checkpoint exists offline): every LayerNorm gain is log-normal around 1 (sigma 0.5: 95 % below 2.3) with
    `outliers` channels per LayerNorm at 5-10 (the massive-activation channels every trained transformer has), LayerNorm / BatchNorm
    biases N(0, 0.1), and every Linear / Conv weight matrix `weight_gain` x larger (std 0.06).  Deterministic (name-keyed like the
    rest of this module); applied to a COPY.  Both sides of the checkpoint-like parity fixtures (tests/golden/make_golden.py
    `ckpt`: the reference; tests/test_model_gpu.py: the HIP path) call this with the same seed."""
    import torch
    out = OrderedDict()
    for k, v in sd.items():
        t = v.clone() if torch.is_tensor(v) else np.array(v)
        is_t = torch.is_tensor(t)
        a = t.detach().cpu().numpy().copy() if is_t else t
        r = _rng("ckpt_like:" + k, seed)
        norm_gain = k.endswith(("norm1.weight", "norm2.weight", "ln_1.weight", "ln_2.weight", "ln_final.weight")) or k == "point_encoder.norm.weight"
        norm_bias = k.endswith(("norm1.bias", "norm2.bias", "ln_1.bias", "ln_2.bias", "ln_final.bias")) or k == "point_encoder.norm.bias"
        if norm_gain:
            a = np.exp(r.normal(0.0, 0.5, size=a.shape)).astype(np.float32)
            if not (k.startswith("ln_final") or k == "point_encoder.norm.weight"):
                # (the final LayerNorms feed the projections: outlier gains there would make every feature parallel -- cosines of
                # 1, a saturated loss -- which no trained checkpoint does; the in-block norms are where the massive channels live)
                idx = r.choice(a.size, size=min(outliers, a.size), replace=False)
                a.reshape(-1)[idx] = r.uniform(outlier_gain[0], outlier_gain[1], size=idx.size).astype(np.float32)
        elif norm_bias:
            a = r.normal(0.0, 0.1, size=a.shape).astype(np.float32)
        elif a.ndim >= 2 and a.dtype == np.float32 and k.endswith("weight") and "token_embedding" not in k and "running" not in k:
            a = (a * weight_gain).astype(np.float32)
        out[k] = torch.from_numpy(a) if is_t else a
    return out
