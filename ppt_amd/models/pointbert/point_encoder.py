"""Mirror of models/pointbert/point_encoder.py:14-257 (Mlp, Attention, Block, TransformerEncoder,
PointTransformer): identical constructor signatures, attribute names and state-dict keys; the
forward of PointTransformer is ONE autograd node running ppt_amd.engine's kernel pipeline."""
import os

import torch
import torch.nn as nn

from ... import blocks as _blk
from ... import engine, gradscale, graphs, ops
from .dvae import Encoder, Group


def _prec(m):
    return getattr(m, "precision", None) or _blk.DEFAULT_PRECISION


class Mlp(nn.Module):
    """point_encoder.py:14-30.  Inside PointTransformer the block is computed by engine.vit_block_forward (one autograd node
    per tower); called on its own, forward() runs the same kernels through ppt_amd.blocks (GEMMs with the GELU and its
    derivative in their epilogues, weight gradients included)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        """point_encoder.py:24-30 (x [..., in_features])."""
        if not x.is_cuda:
            raise RuntimeError("ppt_amd modules run on the HIP device only (no CPU fallback)")
        if self.drop.p:
            raise NotImplementedError("Mlp with drop > 0: every PPT configuration builds it with drop = 0 (point_encoder.py:147)")
        act = {nn.GELU: ops.ACT_GELU}.get(type(self.act))
        if act is None:
            raise NotImplementedError(f"Mlp activation {type(self.act).__name__}: the GEMM epilogues implement nn.GELU")
        return _blk.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, act, _prec(self))


class Attention(nn.Module):
    """point_encoder.py:33-58 (qkv_bias=False; scale applied after q@k^T)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        """point_encoder.py:46-58 (x [B, N, C]; heads of 64)."""
        if not x.is_cuda:
            raise RuntimeError("ppt_amd modules run on the HIP device only (no CPU fallback)")
        if self.attn_drop.p or self.proj_drop.p:
            raise NotImplementedError("Attention with dropout: every PPT configuration builds it with p = 0")
        return _blk.self_attention(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.num_heads,
                                   float(self.scale), False, _prec(self))


class DropPath(nn.Module):
    """timm 0.4.12 DropPath semantics: per-sample factor floor(keep + U[0,1)) / keep in train()."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return _blk.drop_path(x, self.drop_prob, self.training)


class Block(nn.Module):
    """point_encoder.py:61-79."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)

    def _norm(self, ln, x):
        return _blk.layer_norm(x, ln.weight, ln.bias, ln.eps) if isinstance(ln, nn.LayerNorm) else ln(x)

    def forward(self, x):
        """point_encoder.py:76-79."""
        x = x + self.drop_path(self.attn(self._norm(self.norm1, x)))
        x = x + self.drop_path(self.mlp(self._norm(self.norm2, x)))
        return x


class TransformerEncoder(nn.Module):
    """point_encoder.py:82-110."""

    def __init__(self, embed_dim=768, depth=4, num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate,
                  drop_path=drop_path_rate[i] if isinstance(drop_path_rate, list) else drop_path_rate)
            for i in range(depth)])

    def forward(self, x, pos, task='cls'):
        """point_encoder.py:99-110: `block(x + pos)` for every block; task='partseg' returns the outputs of
        blocks 3, 7 and 11."""
        feature_list = []
        for i, block in enumerate(self.blocks):
            x = block(x + pos)
            if task == 'partseg' and i in (3, 7, 11):
                feature_list.append(x)
        return feature_list if task == 'partseg' else x


TIER_PARAMS = {   # ULIP_models.py:461-470, cumulative
    1: ["norm2.weight", "norm2.bias", "mlp.fc2.weight", "mlp.fc2.bias"],
    2: ["norm1.weight", "norm1.bias", "mlp.fc1.weight", "mlp.fc1.bias"],
    3: ["attn.qkv.weight", "attn.proj.weight", "attn.proj.bias"],
}


def remap_pointbert_checkpoint(base_model):
    """Key translation of load_model_from_ckpt (point_encoder.py:206-215, 322-331) for a Point-BERT pre-training
    checkpoint's `base_model` table: the DistributedDataParallel prefix goes, `transformer_q.*` (the dVAE-supervised
    query encoder; its cls_head excepted) and `base_model.*` entries keep their remainder as the key, every other entry
    is dropped."""
    out = {}
    for k, v in base_model.items():
        k = k.replace("module.", "")
        if k.startswith("transformer_q") and not k.startswith("transformer_q.cls_head"):
            out[k[len("transformer_q."):]] = v
        elif k.startswith("base_model"):
            out[k[len("base_model."):]] = v
    return out


def _load_model_from_ckpt(self, bert_ckpt_path):
    """point_encoder.py:206-232 / :322-345: load a Point-BERT checkpoint non-strictly, reporting what did not match."""
    ckpt = torch.load(bert_ckpt_path, map_location=torch.device('cpu'))
    incompatible = self.load_state_dict(remap_pointbert_checkpoint(ckpt['base_model']), strict=False)
    if incompatible.missing_keys:
        print('missing_keys\n  ' + '\n  '.join(incompatible.missing_keys))
    if incompatible.unexpected_keys:
        print('unexpected_keys\n  ' + '\n  '.join(incompatible.unexpected_keys))
    print(f'[{type(self).__name__}] Successful Loading the ckpt from {bert_ckpt_path}')
    return incompatible


class _PointEncoderFn(torch.autograd.Function):
    """feat = PointTransformer(pc); gradients only for the un-frozen last-block parameters."""

    @staticmethod
    def forward(ctx, module, pc, fps_start, dp, tier, names, *params):
        sd, cache, cfg, train = module._live_state(), module._cache(), module._cfg(), module.training
        ctx.module, ctx.tier, ctx.names = module, tier, names
        # the power-of-two scale of the last block's 16-bit backward, fixed now (ppt_amd/gradscale.py; a bare PointTransformer
        # under its own criterion: the batch size)
        ctx.grad_scale = gradscale.current(engine._stage_wc(cache, "last_block").dtype, default_rows=pc.shape[0]) if tier > 0 else 1.0
        # a hipGraph bakes in the device pointers of everything it reads, including the WeightCache's operand copies,
        # which are re-made whenever the optimizer changes a parameter: a section that reads a trainable parameter is
        # never captured.  With an un-frozen last block (head_type >= 1) only the frozen prefix (tokenizer + blocks
        # 0 .. depth-2) is replayed, also under no_grad (validate() between two training epochs).
        last_trains, other_trains = module._trainable_split()
        split = tier > 0 or last_trains
        B, N = pc.shape[0], pc.shape[1]
        drawn = fps_start is None            # the caller injected neither RNG draw: they are made here, inside the graph

        def draw():
            """the two RNG draws of the path (misc.py:59 FPS start, timm DropPath factors), on the device"""
            return (torch.randint(0, N, (B,), dtype=torch.long, device=pc.device), module._draw_drop_path(B, pc.device))

        use_graph = pc.is_cuda and module.use_hip_graphs and ops.profiler is None and not other_trains
        # tier 0 (everything frozen, nothing kept for a backward): ~140 launches whose arguments depend on shapes only --
        # replayed from a hipGraph after graphs.WARMUP_CALLS eager calls (ppt_amd/graphs.py; 2 us less per launch)
        if tier == 0 and module.param_gate is not None and pc.is_cuda:      # e.g. an evaluation right behind a training step
            torch.cuda.current_stream().wait_event(module.param_gate)
            module.param_gate = None
        has_dp = (train and module.drop_path_rate > 0) if drawn else dp is not None
        key = ("point_fwd", tuple(pc.shape), has_dp, drawn, train, cache.dtype, ops.get_persistent_occupancy() if pc.is_cuda else 100)
        ahead = module.group_ahead if (use_graph and module._graphs.ready(("group", tuple(pc.shape), drawn))) else None
        if not split and use_graph and module._graphs.ready(key):
            if ahead is not None:
                # FPS + kNN (and, tokenize_ahead, the whole tokenizer) ran -- or run right now -- on the grouping stream,
                # overlapping the previous iteration's blocks
                tk = module.tokenize_ahead
                grouped, slot = module._group_ahead(pc, fps_start, drawn, ahead, tokenize=tk)
                ins = list(grouped) + ([dp] if dp is not None else [])

                def build():
                    def fn(nb_, ce_, *rest):
                        dp_ = module._draw_drop_path(B, pc.device) if drawn else (rest[0] if rest else None)
                        kw = dict(tokens=(nb_, ce_)) if tk else dict(grouped=(nb_, ce_))
                        feat_, _ = engine.point_encoder_forward(sd, "", cache, None, None, dp_, train, 0, cfg, **kw)
                        return (feat_,), None
                    # (the tokens are handed over in place: one tower graph per ping-pong pair instead of 2 x 25 MB of copies)
                    return graphs.GraphedCall(fn, ins, alias_inputs=tk and dp is None)
                (feat,), _ = module._graphs.get(key + ((("tokens", slot) if dp is None else ("tokens",)) if tk else ("grouped",)), build)(*ins)
                module._group_consumed(slot)
                ctx.saved = None
                return feat.clone()
            ins = [pc] if drawn else [pc, fps_start] + ([dp] if dp is not None else [])

            def build():
                def fn(pc_, *rest):
                    start_, dp_ = draw() if drawn else (rest[0], rest[1] if len(rest) > 1 else None)
                    feat_, _ = engine.point_encoder_forward(sd, "", cache, pc_, start_, dp_, train, 0, cfg)
                    return (feat_,), None
                return graphs.GraphedCall(fn, ins)
            (feat,), _ = module._graphs.get(key, build)(*ins)
            ctx.saved = None
            return feat.clone()
        if split and pc.is_cuda:
            # something in the last block trains: everything in front of it is still frozen.  That prefix is replayed
            # from a hipGraph, and only the rest waits for the optimizer of the previous iteration (module.param_gate,
            # set by train.Trainer.step) -- the prefix runs ahead of it like the whole tower does for head_type 0.
            key = ("point_prefix", tuple(pc.shape), has_dp, drawn, train, cache.dtype, ops.get_persistent_occupancy() if pc.is_cuda else 100)
            if use_graph and module._graphs.ready(key):
                slot = None
                if ahead is not None:
                    tk = module.tokenize_ahead
                    grouped, slot = module._group_ahead(pc, fps_start, drawn, ahead, tokenize=tk)
                    ins = list(grouped) + ([dp] if dp is not None else [])

                    def build():
                        def fn(nb_, ce_, *rest):
                            dp_ = module._draw_drop_path(B, pc.device) if drawn else (rest[0] if rest else None)
                            kw = dict(tokens=(nb_, ce_)) if tk else dict(grouped=(nb_, ce_))
                            x2_, pos2_ = engine.point_encoder_forward(sd, "", cache, None, None, dp_, train, 0, cfg,
                                                                      last_block=False, **kw)
                            return (x2_, pos2_) + ((dp_,) if dp_ is not None else ()), None
                        # (NOT handed over in place here: the last block keeps activations for its backward, and they must not
                        # live in the stage's ping-pong buffers -- the stage of a later step, which waits for the forward only,
                        # would overwrite them before that backward has run.  Releasing the pair behind the backward instead was
                        # measured: it costs the whole gain of the stage, C3 6.27 -> 6.59 ms; two 50 MB copies cost 35 us.)
                        return graphs.GraphedCall(fn, ins)
                    key = key + (("tokens",) if tk else ("grouped",))
                else:
                    ins = [pc] if drawn else [pc, fps_start] + ([dp] if dp is not None else [])

                    def build():
                        def fn(pc_, *rest):
                            start_, dp_ = draw() if drawn else (rest[0], rest[1] if len(rest) > 1 else None)
                            x2_, pos2_ = engine.point_encoder_forward(sd, "", cache, pc_, start_, dp_, train, 0, cfg, last_block=False)
                            return (x2_, pos2_) + ((dp_,) if dp_ is not None else ()), None
                        return graphs.GraphedCall(fn, ins)
                g = module._graphs.get(key, build)
                outs, _ = g(*ins)
                if slot is not None:
                    module._group_consumed(slot)
                g.generation = getattr(g, "generation", 0) + 1
                ctx.prefix_graph, ctx.prefix_generation = g, g.generation
                cut, dp = (outs[0], outs[1]), (outs[2] if len(outs) > 2 else None)
            else:
                if drawn:
                    fps_start, dp = draw()
                cut = engine.point_encoder_forward(sd, "", cache, pc, fps_start, dp, train, 0, cfg, last_block=False)
            gate, module.param_gate = module.param_gate, None
            if gate is not None:
                torch.cuda.current_stream().wait_event(gate)
            feat, saved = engine.point_encoder_forward(sd, "", cache, None, None, dp, train, tier, cfg, resume=cut)
        else:
            if drawn:
                fps_start, dp = draw()
            feat, saved = engine.point_encoder_forward(sd, "", cache, pc, fps_start, dp, train, tier, cfg)
        ctx.saved = saved
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        m = ctx.module
        g = getattr(ctx, "prefix_graph", None)
        if g is not None and g.generation != ctx.prefix_generation:
            raise RuntimeError("the point tower's captured activations were overwritten by a later forward; set "
                               "model.point_encoder.use_hip_graphs = False to keep several forwards alive before backward")
        grads = engine.point_encoder_backward(m._live_state(), m._cache(), ctx.saved, dfeat.contiguous().float(), ctx.tier,
                                              grad_scale=ctx.grad_scale)
        out = []
        for n in ctx.names:
            g = grads[n]
            out.append(g.view_as(m._live_state()[n]))
        return (None, None, None, None, None, None) + tuple(out)


class PointTransformer(nn.Module):
    """point_encoder.py:113-257.  `config` carries trans_dim, depth, drop_path_rate, cls_dim,
    num_heads, group_size, num_group, encoder_dims (PointTransformer_8192point.yaml:15-25).

    Extra, build-specific knobs (not in the reference): `precision` (torch.bfloat16 performance mode /
    torch.float32 parity mode) and the injection hooks `fps_start` / `drop_path_factors` used by the
    parity tests (SURVEY.md App. A Q8, §7 'RNG parity')."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        self.args = kwargs.get("args")
        self.trans_dim = config.trans_dim
        self.depth = config.depth
        self.drop_path_rate = config.drop_path_rate
        self.cls_dim = config.cls_dim
        self.num_heads = config.num_heads
        self.group_size = config.group_size
        self.num_group = config.num_group
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.encoder_dims = config.encoder_dims
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.reduce_dim = nn.Linear(self.encoder_dims, self.trans_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, self.trans_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.dpr = dpr
        self._keep = None
        self.blocks = TransformerEncoder(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr,
                                         num_heads=self.num_heads)
        self.norm = nn.LayerNorm(self.trans_dim)
        self.precision = torch.bfloat16
        self._graphs = graphs.GraphCache()
        self.use_hip_graphs = True
        self.param_gate = None           # event after which this iteration may read the last block's parameters (Trainer)
        # stream for the grouping stage (FPS + kNN, a function of the input cloud alone) when the caller vouches that the
        # cloud is complete in memory at the time of the call (train.Trainer.inputs_ready): it then overlaps the previous
        # iteration's transformer blocks instead of heading this one's critical path.  None: grouping stays in the tower.
        self.group_ahead = None
        # ... and with it the rest of the tokenizer (mini-PointNet, reduce_dim, pos_embed: frozen in every PPT configuration, a
        # function of the cloud alone, HBM-bound where the blocks it then runs beside are not).  PPT_TOKENIZE_AHEAD=0: FPS + kNN only.
        self.tokenize_ahead = os.environ.get("PPT_TOKENIZE_AHEAD", "1") != "0"
        self._group_slot = 0
        self._group_free = [None, None]
        self.fps_start = None            # [B] int64: injected FPS start indices (else torch.randint)
        self.drop_path_factors = None    # [depth,2,B] fp32: injected DropPath factors (else drawn on device)
        self._wc = None
        self._sd = None

    # ---- plumbing -------------------------------------------------------------------------
    def _cache(self):
        if self._wc is None or self._wc.dtype != self.precision:
            self._wc = engine.WeightCache(self.precision, self.__dict__.setdefault("demoted", set()))
        if self.precision == torch.float32:
            # (ULIP_WITH_IMAGE.set_precision("split16").  A 16-bit tower has no fp32 GEMM of its own and leaves the switch to the
            # owner's text tower: a mixed-mode model whose text tower runs split16 then sees ONE value in forward and backward)
            ops.set_split16(self.__dict__.get("split16", False))
        return self._wc

    def _live_state(self):
        ref = self.cls_token
        if self._sd is None or self._sd[1] is not ref or self._sd[2] != ref.device:
            self._sd = (self.state_dict(keep_vars=True), ref, ref.device)
        return self._sd[0]

    def _apply(self, fn, *a, **k):
        self._sd = None
        if hasattr(self, "_graphs"):
            self._graphs.clear()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._sd = None
        self._wc = None
        if hasattr(self, "_graphs"):
            self._graphs.clear()
        return super().load_state_dict(*a, **k)

    def _cfg(self):
        return dict(num_group=self.num_group, group_size=self.group_size, trans_dim=self.trans_dim, depth=self.depth,
                    num_heads=self.num_heads)

    def _tier(self):
        """head_type tier implied by the requires_grad flags of the last block (ULIP_models.py:461-470)."""
        sd = dict(self.blocks.blocks[-1].named_parameters())
        tier = 0
        for t in (1, 2, 3):
            if tier == t - 1 and all(sd[n].requires_grad for n in TIER_PARAMS[t]):
                tier = t
        expected = {n for t in (1, 2, 3) if t <= tier for n in TIER_PARAMS[t]}
        actual = {n for n, q in sd.items() if q.requires_grad}
        if actual != expected:
            raise RuntimeError("un-frozen last-block parameters must follow the cumulative head_type tiers of "
                               f"ULIP_models.py:461-470; got {sorted(actual)}")
        return tier

    def _trainable_split(self):
        """(something in the last block trains, something in front of it / behind it trains)."""
        in_last = sum(1 for q in self.blocks.blocks[-1].parameters() if q.requires_grad)
        total = sum(1 for q in self.parameters() if q.requires_grad)
        return in_last > 0, total > in_last

    def _group_ahead(self, pc, fps_start, drawn, side, tokenize=False):
        """Group.forward (dvae.py:159-181) of `pc` on `side`, replayed from one of two hipGraphs with their own output
        buffers (the previous tower may still be reading the other pair).  The caller's stream is made to wait for the
        result; `side` waits for nothing but the tower that last read this pair -- NOT for the caller's stream, which is
        the point: the caller guarantees `pc` (and `fps_start`) are complete in memory.  -> ((nbhd, center), slot).
        tokenize: the stage runs the whole tokenizer (engine.tokenize_points: + mini-PointNet, reduce_dim, pos_embed -- frozen
        weights, train-mode BatchNorm statistics of this batch) -> ((x2, pos2), slot), the blocks' input."""
        main = torch.cuda.current_stream()
        slot = self._group_slot = (self._group_slot + 1) % len(self._group_free)
        B, N = pc.shape[0], pc.shape[1]
        G, n = self.num_group, self.group_size
        ins = [pc] if drawn else [pc, fps_start]
        key = ("group", tuple(pc.shape), drawn, slot)
        if tokenize:
            sd, cache, cfg, train = self._live_state(), self._cache(), self._cfg(), self.training
            key = ("tokens", tuple(pc.shape), drawn, train, cache.dtype, ops.get_persistent_occupancy(), slot)

        def build():
            def fn(pc_, *rest):
                start_ = torch.randint(0, N, (B,), dtype=torch.long, device=pc_.device) if drawn else rest[0]   # misc.py:59
                if tokenize:
                    return tuple(engine.tokenize_points(sd, "", cache, pc_, start_, train, cfg)[:2]), None
                return tuple(engine.group_points(pc_, G, n, start_)), None
            return graphs.GraphedCall(fn, ins)
        with torch.cuda.stream(side):
            if self._group_free[slot] is not None:
                side.wait_event(self._group_free[slot])
            # (a DevicePrefetcher batch: behind its copy; a vouched-for tensor: nothing; a tensor DERIVED from the batch on the
            # caller's stream, which no event covers: behind the caller's stream -- graphs.wait_inputs)
            graphs.wait_inputs(side, ins, main, getattr(self, "inputs_vouched", False))
            grouped, _ = self._graphs.get(key, build)(*ins)
            done = side.record_event()
        for t in ins:
            t.record_stream(side)
        main.wait_event(done)
        return grouped, slot

    def _group_consumed(self, slot):
        """The tower reading pair `slot` has been queued on the current stream: the pair is free once it has run."""
        self._group_free[slot] = torch.cuda.current_stream().record_event()

    def _draw_drop_path(self, B, device):
        """timm DropPath factors for both residual branches of every block (train mode only)."""
        if self.drop_path_factors is not None:
            return self.drop_path_factors.to(device=device, dtype=torch.float32).contiguous()
        if not self.training or self.drop_path_rate <= 0:
            return None
        if self._keep is None or self._keep.device != device:      # (a host->device copy per step would drain the stream)
            self._keep = 1.0 - torch.tensor(self.dpr, dtype=torch.float32, device=device).view(-1, 1, 1)
        keep = self._keep
        u = torch.rand((self.depth, 2, B), dtype=torch.float32, device=device)
        return (torch.floor(keep + u) / keep).contiguous()

    # ---- reference API ---------------------------------------------------------------------
    load_model_from_ckpt = _load_model_from_ckpt

    def forward(self, pts):
        """pts [B,N,3] -> cat(cls, max) features [B, 2*trans_dim] (point_encoder.py:234-257)."""
        pts = pts.contiguous().float()
        B, N, _ = pts.shape
        if self.fps_start is None and self.drop_path_factors is None:
            start = dp = None                    # both draws happen inside _PointEncoderFn (and inside its hipGraph)
        else:
            start = self.fps_start
            if start is None:
                start = torch.randint(0, N, (B,), dtype=torch.long, device=pts.device)   # misc.py:59
            start = start.to(pts.device).contiguous()
            dp = self._draw_drop_path(B, pts.device)
        tier = self._tier() if torch.is_grad_enabled() else 0
        names = []
        for t in (1, 2, 3):
            if tier >= t:
                names += [f"blocks.blocks.{self.depth - 1}.{n}" for n in TIER_PARAMS[t]]
        sd = self._live_state()
        return _PointEncoderFn.apply(self, pts, start, dp, tier, names, *[sd[n] for n in names])


class PointTransformer_partseg(nn.Module):
    """point_encoder.py:260-420.  forward(pts [B,N,3], cls_label one-hot [B,16]) -> per-point features [B,N,128].
    The PointBERT backbone (tokenizer + 12 blocks, frozen in PPT: ULIP_models.py:550-565) runs on the engine
    pipeline; the decoder modules keep the reference names (propagation_{0,1,2}, dgcnn_pro_{1,2}, conv1, bn1, drop1,
    conv2 -- conv2 is unused by forward, as in the reference)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        from .pointnet2_utils import DGCNN_Propagation, PointNetFeaturePropagation
        self.config = config
        self.trans_dim = config.trans_dim
        self.depth = config.depth
        self.drop_path_rate = config.drop_path_rate
        self.cls_dim = config.cls_dim
        self.num_heads = config.num_heads
        self.group_size = config.group_size
        self.num_group = config.num_group
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.encoder_dims = config.encoder_dims
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.reduce_dim = nn.Linear(self.encoder_dims, self.trans_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, self.trans_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.dpr = dpr
        self._keep = None
        self.blocks = TransformerEncoder(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr,
                                         num_heads=self.num_heads)
        self.norm = nn.LayerNorm(self.trans_dim)
        self.propagation_2 = PointNetFeaturePropagation(in_channel=self.trans_dim + 3, mlp=[self.trans_dim * 4, self.trans_dim])
        self.propagation_1 = PointNetFeaturePropagation(in_channel=self.trans_dim + 3, mlp=[self.trans_dim * 4, self.trans_dim])
        self.propagation_0 = PointNetFeaturePropagation(in_channel=self.trans_dim + 3 + 16,
                                                        mlp=[self.trans_dim * 4, self.trans_dim])
        self.dgcnn_pro_1 = DGCNN_Propagation(k=4)
        self.dgcnn_pro_2 = DGCNN_Propagation(k=4)
        self.conv1 = nn.Conv1d(self.trans_dim, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, self.cls_dim, 1)
        self._precision = torch.bfloat16
        self.fps_start = None            # (tokenizer [B], level-1 [B], level-2 [B]) injected FPS starts
        self.drop_path_factors = None    # [depth,2,B]
        self.dropout_mask = None         # [B,N,128] multiplicative Dropout(0.5) factors
        self._wc = None
        self._sd = None
        # The frozen backbone's part of the forward (three FPS, kNN, tokenizer, 12 blocks: ~200 launches) is replayed from a
        # hipGraph after graphs.WARMUP_CALLS eager calls per shape: the part-seg step is ~690 launches of small kernels and
        # was HOST-bound -- 7.5 ... 16 ms of enqueue time per step depending on the host's load for ~6 ms of GPU work
        # (tools/host_time.py).  Only with the RNG draws made on the device (nothing injected).  (The decoder's forward +
        # backward as graphs through torch.cuda.make_graphed_callables was tried: hipStreamEndCapture of the BACKWARD capture
        # segfaults in this ROCm / PyTorch build whenever any eager GPU work preceded it in the process, also for a
        # two-line pure-ATen module.)
        self.use_hip_graphs = True
        self._graphs = graphs.GraphCache()
        self._graph_injected = False      # tests: allow the graph with injected (static) RNG tensors
        self.graph_decoder = os.environ.get("PPT_PARTSEG_GRAPH_DECODER", "1") != "0"
        self.backbone_ahead = os.environ.get("PPT_PARTSEG_BACKBONE_AHEAD", "1") != "0"   # with group_ahead: the whole frozen backbone runs ahead
        self.decoder_gate = None          # event after which this iteration may read the decoder's parameters (train.Trainer)
        self.group_ahead = None           # stream for the grouping stage of a step whose inputs the caller vouches for (Trainer)
        self._ahead = graphs.AheadStage()
        self.precision = torch.bfloat16   # (the setter also hands the decoder modules their operand format)

    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, dtype):
        self._precision = dtype
        self._graphs.clear()
        # the decoder's GEMM operand format: IEEE half in the performance mode (engine.DECODER_F16), else as the backbone
        self._dec_precision = torch.float16 if (dtype == torch.bfloat16 and engine.DECODER_F16 and "decoder" not in self.__dict__.setdefault("demoted", set())) else dtype
        for m in (self.propagation_0, self.propagation_1, self.propagation_2, self.dgcnn_pro_1, self.dgcnn_pro_2):
            m.precision = self._dec_precision

    _cache = PointTransformer._cache
    _cfg = PointTransformer._cfg
    _draw_drop_path = PointTransformer._draw_drop_path

    def _live_state(self):
        return self.state_dict(keep_vars=True)

    def load_state_dict(self, *a, **k):
        self._wc = None
        self._graphs.clear()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._graphs.clear()
        return super()._apply(fn, *a, **k)

    load_model_from_ckpt = _load_model_from_ckpt

    def forward(self, pts, cls_label):

        from ...autograd import conv_bn_relu_rows
        pts = pts.contiguous().float()
        B, N, _ = pts.shape
        dev = pts.device
        def grouping(p):
            """The part of the step that depends on the input cloud alone: three FPS + the kNN grouping."""
            if self.fps_start is not None:
                s0, s1, s2 = (t.to(dev).contiguous() for t in self.fps_start)
            else:                                        # three independent random starts
                s0, s1, s2 = (torch.randint(0, N, (B,), dtype=torch.long, device=dev) for _ in range(3))
            with torch.no_grad():
                # the three farthest-point samplings of the step (group centres, 512 and 256 decoder anchors: point_encoder.py
                # :360-364) are independent walks over the same clouds and each is one serial workgroup per cloud: run as ONE
                # launch over 3B clouds; the 256-point sampling is the first 256 picks of a 512-point one from the same start
                if self.num_group == 512:
                    _, ctr = ops.fps(p.repeat(3, 1, 1), 512, torch.cat([s0, s1, s2]))
                    center, c1, c2 = ctr[:B].contiguous(), ctr[B:2 * B].contiguous(), ctr[2 * B:, :256].contiguous()
                else:
                    _, center = ops.fps(p, self.num_group, s0)
                    _, c1 = ops.fps(p, 512, s1)
                    _, c2 = ops.fps(p, 256, s2)
                _, nbhd = ops.knn_group(p, center, self.group_size, want_idx=False)
            return (center, c1, c2, nbhd), None

        def blocks(nbhd, center):
            dp = self._draw_drop_path(B, dev)
            with torch.no_grad():                        # frozen backbone: features after blocks 3, 7, 11 (+ final LN, cls dropped)
                feats, ctr = engine.point_encoder_forward(self._live_state(), "", self._cache(), None, None, dp, self.training, 0,
                                                          self._cfg(), fetch=(3, 7, 11), grouped=(nbhd, center))
            return (feats[0], feats[1], feats[2], ctr.contiguous()), None

        def backbone(p):
            (center, c1, c2, nbhd), _ = grouping(p)
            (f0_, f1_, f2_, ctr), _ = blocks(nbhd, center)
            return (f0_, f1_, f2_, ctr, c1, c2), None

        injected = self.fps_start is not None or self.drop_path_factors is not None or self.dropout_mask is not None
        key = ("partseg_backbone", (B, N), self.training, self._precision)
        slot = None
        if (pts.is_cuda and self.use_hip_graphs and graphs.enabled and ops.profiler is None and (not injected or self._graph_injected)
                and not torch.cuda.is_current_stream_capturing() and self._graphs.ready(key)):
            if self.group_ahead is not None and self.backbone_ahead:
                # ... and the WHOLE frozen backbone with it (round 3): tokenizer, the 12 blocks and the three feature taps are a
                # function of the cloud, the frozen weights and RNG draws made inside the stage's graph -- nothing the optimizer
                # writes -- so the next step's backbone runs under this step's decoder (hundreds of small kernels that leave
                # most of the chip idle) instead of in front of it
                outs, slot = self._ahead.run(self._graphs, ("partseg_backbone_ahead", (B, N), self.training, self._precision), backbone,
                                             [pts], self.group_ahead, vouched=getattr(self, "inputs_vouched", False))
                f_a, f_b, f_c, center, c1, c2 = (o.clone() for o in outs)
                self._ahead.consumed(slot)
            elif self.group_ahead is not None:
                # the caller vouches that `pts` is complete in memory (train.Trainer.inputs_ready): the grouping stage runs on its
                # own stream as soon as the step is called -- under the previous iteration's decoder backward (0.3 ms of serial
                # FPS walks off the head of the caller's stream) -- and only the blocks wait for it (graphs.AheadStage)
                (center, c1, c2, nbhd), slot = self._ahead.run(self._graphs, ("partseg_group", (B, N)), grouping, [pts], self.group_ahead, vouched=getattr(self, "inputs_vouched", False))
                bkey = ("partseg_blocks", (B, N), self.training, self._precision)
                outs, _ = self._graphs.get(bkey, lambda: graphs.GraphedCall(blocks, [nbhd, center]))(nbhd, center)
                f_a, f_b, f_c, center = (o.clone() for o in outs)
                c1, c2 = c1.clone(), c2.clone()
                self._ahead.consumed(slot)
            else:
                outs, _ = self._graphs.get(key, lambda: graphs.GraphedCall(backbone, [pts]))(pts)
                f_a, f_b, f_c, center, c1, c2 = (o.clone() for o in outs)          # (the decoder's autograd nodes keep them)
        else:
            (f_a, f_b, f_c, center, c1, c2), _ = backbone(pts)
        feats = (f_a, f_b, f_c)
        if self.decoder_gate is not None:        # the optimizer of the previous iteration (on the text stream) has written the decoder
            if pts.is_cuda:
                torch.cuda.current_stream().wait_event(self.decoder_gate)
            self.decoder_gate = None
        dkey = ("partseg_decoder_warm", (B, N), self.training, self._precision)
        if (pts.is_cuda and self.training and torch.is_grad_enabled() and self.use_hip_graphs and self.graph_decoder and graphs.enabled
                and ops.profiler is None and (not injected or self._graph_injected) and not torch.cuda.is_current_stream_capturing()
                and self._graphs.ready(dkey)):
            # the decoder's forward AND backward from hipGraphs, as one autograd node (ppt_amd.autograd._PartsegDecoder)
            from ...autograd import _PartsegDecoder, partseg_decoder_params
            drop = self.dropout_mask.to(dev) if self.dropout_mask is not None else None
            return _PartsegDecoder.apply(self, f_a, f_b, f_c, center, c1, c2, pts, cls_label, drop, *partseg_decoder_params(self))
        f0 = torch.cat([cls_label.float().view(B, 1, 16).expand(-1, N, -1), pts], dim=-1)       # [B,N,19]
        f2 = self.propagation_2.forward_rows(c2, center, c2, feats[1])
        f1 = self.propagation_1.forward_rows(c1, center, c1, feats[0])
        f2 = self.dgcnn_pro_2.forward_rows(center, feats[2], c2, f2)
        f1 = self.dgcnn_pro_1.forward_rows(c2, f2, c1, f1)
        f0 = self.propagation_0.forward_rows(pts, c1, f0, f1)
        y = conv_bn_relu_rows(f0.reshape(B * N, -1), self.conv1, self.bn1, self.training, self._dec_precision)
        y = y.view(B, N, -1)
        if self.dropout_mask is not None:
            y = y * self.dropout_mask.to(dev)
        elif self.training:
            y = self.drop1(y)
        return y
