"""Mirror of models/pointmlp/pointMLP.py: the same module tree and state-dict keys (embedding.net.{0,1},
local_grouper_list.{i}.affine_{alpha,beta}, pre_blocks_list.{i}.transfer / .operation.{j}.net{1,2},
pos_blocks_list.{i}.operation.{j}.net{1,2}, classifier.{0,1,4,5}) and the same constructor signatures; the forward runs
ppt_amd.engine.pointmlp_forward (FPS, kNN, gather and GEMM kernels).  The configurations the reference instantiates are
supported -- pointMLP() (ULIP_PN_MLP, ULIP_models.py:399-400) and pointMLPElite(): ReLU, groups=1, bias=False,
use_xyz=False, normalize="anchor"; anything else raises."""
import torch
import torch.nn as nn

from ... import engine, graphs, ops


def get_activation(activation):
    """pointMLP.py:6-20; only the ReLU both factory functions ask for is on the path."""
    if activation.lower() != "relu":
        raise NotImplementedError(f"activation {activation!r}: pointMLP()/pointMLPElite() use relu (pointMLP.py:359-370)")
    return nn.ReLU(inplace=True)


class LocalGrouper(nn.Module):
    """pointMLP.py:124-150 (parameter container: the affine of the anchor normalisation)."""

    def __init__(self, channel, groups, kneighbors, use_xyz=True, normalize="center", **kwargs):
        super().__init__()
        self.groups, self.kneighbors, self.use_xyz = groups, kneighbors, use_xyz
        self.normalize = normalize.lower() if normalize is not None else None
        if self.normalize != "anchor" or use_xyz:
            raise NotImplementedError("only normalize='anchor', use_xyz=False (pointMLP.py:359-370) is on the PPT path")
        self.affine_alpha = nn.Parameter(torch.ones([1, 1, 1, channel]))
        self.affine_beta = nn.Parameter(torch.zeros([1, 1, 1, channel]))


class ConvBNReLU1D(nn.Module):
    """pointMLP.py:184-196."""

    def __init__(self, in_channels, out_channels, kernel_size=1, bias=True, activation='relu'):
        super().__init__()
        if bias or kernel_size != 1:
            raise NotImplementedError("bias=False, kernel_size=1 (pointMLP.py:359-370) is on the PPT path")
        self.act = get_activation(activation)
        self.net = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size, bias=bias), nn.BatchNorm1d(out_channels), self.act)


class ConvBNReLURes1D(nn.Module):
    """pointMLP.py:199-231 (groups=1 branch)."""

    def __init__(self, channel, kernel_size=1, groups=1, res_expansion=1.0, bias=True, activation='relu'):
        super().__init__()
        if bias or kernel_size != 1 or groups != 1:
            raise NotImplementedError("bias=False, kernel_size=1, groups=1 (pointMLP.py:359-370) is on the PPT path")
        self.act = get_activation(activation)
        mid = int(channel * res_expansion)
        self.net1 = nn.Sequential(nn.Conv1d(channel, mid, kernel_size, groups=groups, bias=bias), nn.BatchNorm1d(mid), self.act)
        self.net2 = nn.Sequential(nn.Conv1d(mid, channel, kernel_size, bias=bias), nn.BatchNorm1d(channel))


class PreExtraction(nn.Module):
    """pointMLP.py:234-262."""

    def __init__(self, channels, out_channels, blocks=1, groups=1, res_expansion=1, bias=True, activation='relu', use_xyz=True):
        super().__init__()
        in_channels = 3 + 2 * channels if use_xyz else 2 * channels
        self.transfer = ConvBNReLU1D(in_channels, out_channels, bias=bias, activation=activation)
        self.operation = nn.Sequential(*[ConvBNReLURes1D(out_channels, groups=groups, res_expansion=res_expansion, bias=bias,
                                                         activation=activation) for _ in range(blocks)])


class PosExtraction(nn.Module):
    """pointMLP.py:265-282."""

    def __init__(self, channels, blocks=1, groups=1, res_expansion=1, bias=True, activation='relu'):
        super().__init__()
        self.operation = nn.Sequential(*[ConvBNReLURes1D(channels, groups=groups, res_expansion=res_expansion, bias=bias,
                                                         activation=activation) for _ in range(blocks)])


class Model(nn.Module):
    """pointMLP.py:285-334.  forward(x [B,N,3]) -> [B,256].

    Build-specific knobs: `precision` (bf16 / fp32 parity), `fps_start` = one start vector [B] per stage and
    `dropout_masks` = (m1 [B,512], m2 [B,256]) to inject the RNG draws of the path in parity tests."""

    chain_priority_hint = 0     # ULIP_WITH_IMAGE.chain_priority: this tower (4.6 ms) is the step's critical path, not the prompt chain

    def __init__(self, points=1024, embed_dim=64, groups=1, res_expansion=1.0, activation="relu", bias=True, use_xyz=True,
                 normalize="center", dim_expansion=[2, 2, 2, 2], pre_blocks=[2, 2, 2, 2], pos_blocks=[2, 2, 2, 2],
                 k_neighbors=[32, 32, 32, 32], reducers=[2, 2, 2, 2], **kwargs):
        super().__init__()
        self.stages = len(pre_blocks)
        self.points = points
        assert len(pre_blocks) == len(k_neighbors) == len(reducers) == len(pos_blocks) == len(dim_expansion), \
            "Please check stage number consistent for pre_blocks, pos_blocks k_neighbors, reducers."
        self.embedding = ConvBNReLU1D(3, embed_dim, bias=bias, activation=activation)
        self.local_grouper_list = nn.ModuleList()
        self.pre_blocks_list = nn.ModuleList()
        self.pos_blocks_list = nn.ModuleList()
        last_channel, anchor_points = embed_dim, points
        for i in range(self.stages):
            out_channel = last_channel * dim_expansion[i]
            anchor_points = anchor_points // reducers[i]
            self.local_grouper_list.append(LocalGrouper(last_channel, anchor_points, k_neighbors[i], use_xyz, normalize))
            self.pre_blocks_list.append(PreExtraction(last_channel, out_channel, pre_blocks[i], groups=groups,
                                                      res_expansion=res_expansion, bias=bias, activation=activation, use_xyz=use_xyz))
            self.pos_blocks_list.append(PosExtraction(out_channel, pos_blocks[i], groups=groups, res_expansion=res_expansion,
                                                      bias=bias, activation=activation))
            last_channel = out_channel
        self.act = get_activation(activation)
        self.classifier = nn.Sequential(nn.Linear(last_channel, 512), nn.BatchNorm1d(512), self.act, nn.Dropout(0.5),
                                        nn.Linear(512, 256), nn.BatchNorm1d(256), self.act, nn.Dropout(0.5))
        self._cfg = dict(points=points, k_neighbors=list(k_neighbors), reducers=list(reducers), pre_blocks=list(pre_blocks),
                         pos_blocks=list(pos_blocks))
        self.precision = torch.bfloat16
        self.fps_start = None
        self.dropout_masks = None
        self._wc = None
        self._sd = None
        self._graphs = graphs.GraphCache()
        self.use_hip_graphs = True
        self.group_ahead = None            # stream for the grouping stage of the next iteration (train.Trainer.inputs_ready)
        self.group_ahead_pays_when_frozen = True
        self._ahead = graphs.AheadStage()

    def _apply(self, fn, *a, **k):
        self._sd = None
        if hasattr(self, "_graphs"):
            self._graphs.clear()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._sd, self._wc = None, None
        if hasattr(self, "_graphs"):
            self._graphs.clear()
        return super().load_state_dict(*a, **k)

    def load_model_from_ckpt(self, ckpt_path):
        """pointMLP.py:336-350."""
        ckpt = torch.load(ckpt_path)
        base_ckpt = {k.replace("module.", ""): v for k, v in ckpt['net'].items()}
        incompatible = self.load_state_dict(base_ckpt, strict=False)
        if incompatible.missing_keys:
            print("incompatible keys:", incompatible.missing_keys)
        if incompatible.unexpected_keys:
            print("unexpected keys:", incompatible.unexpected_keys)
        print("finished loading ckpt from {}".format(ckpt_path))

    def forward(self, x):
        xyz = x.contiguous().float()
        B, N, _ = xyz.shape
        cfg = self._cfg
        if self._wc is None or self._wc.dtype != self.precision:
            self._wc = engine.WeightCache(self.precision)
        if self.precision == torch.float32:
            ops.set_split16(self.__dict__.get("split16", False))
        n_src, s = [N], cfg["points"]                                    # points FPS draws its start from, per stage
        for r in cfg["reducers"][:-1]:
            s //= r
            n_src.append(s)
        if self.fps_start is not None:
            starts = tuple(s.to(xyz.device).contiguous() for s in self.fps_start)
        else:                                                            # furthest_point_sample's draw (pointMLP.py:77)
            starts = tuple(torch.randint(0, n, (B,), dtype=torch.long, device=xyz.device) for n in n_src)
        masks = None
        if self.dropout_masks is not None:
            masks = tuple(m.to(device=xyz.device, dtype=torch.float32).contiguous() for m in self.dropout_masks)
        elif self.training:
            masks = ((torch.rand((B, 512), device=xyz.device) >= 0.5).float() * 2.0,
                     (torch.rand((B, 256), device=xyz.device) >= 0.5).float() * 2.0)
        w0 = self.embedding.net[0].weight
        if self._sd is None or self._sd[1] is not w0:
            self._sd = (self.state_dict(keep_vars=True), w0)
        sd, wc, train = self._sd[0], self._wc, self.training
        ns = len(starts)
        with torch.no_grad():            # every parameter of this encoder is frozen in PPT (ULIP_models.py:425-439)
            key = ("pointmlp", tuple(xyz.shape), masks is not None, train, wc.dtype)
            if xyz.is_cuda and self.use_hip_graphs and ops.profiler is None and self._graphs.ready(key):
                if self.group_ahead is not None:
                    # FPS + kNN of the four stages on the grouping stream (graphs.AheadStage); starts that are not injected
                    # are drawn there too (the ones drawn above sit on the caller's stream, behind the previous step)
                    drawn = self.fps_start is None

                    def gfn(x, *st):
                        st = st if st else tuple(torch.randint(0, n, (B,), dtype=torch.long, device=x.device) for n in n_src)
                        return tuple(engine.pointmlp_group(x, st, cfg)), None
                    grouped, slot = self._ahead.run(self._graphs, ("pointmlp_group", tuple(xyz.shape), drawn), gfn,
                                                    [xyz] + ([] if drawn else list(starts)), self.group_ahead, vouched=getattr(self, "inputs_vouched", False))
                    ng = len(grouped)
                    ins = [xyz] + list(grouped) + (list(masks) if masks is not None else [])

                    def fn2(x_, *a):
                        m = a[ng:]
                        return (engine.pointmlp_forward(sd, "", wc, x_, None, train, tuple(m) if m else None, cfg=cfg,
                                                        grouped=list(a[:ng])),), None
                    (feat,), _ = self._graphs.get(key + ("grouped",), lambda: graphs.GraphedCall(fn2, ins))(*ins)
                    self._ahead.consumed(slot)
                    return feat.clone()
                ins = [xyz] + list(starts) + (list(masks) if masks is not None else [])

                def fn(x_, *rest):
                    m = rest[ns:]
                    return (engine.pointmlp_forward(sd, "", wc, x_, rest[:ns], train, tuple(m) if m else None, cfg=cfg),), None
                (feat,), _ = self._graphs.get(key, lambda: graphs.GraphedCall(fn, ins))(*ins)
                return feat.clone()
            return engine.pointmlp_forward(sd, "", wc, xyz, starts, train, masks, cfg=cfg)


def pointMLP(**kwargs) -> Model:
    """pointMLP.py:359-363."""
    return Model(points=1024, embed_dim=64, groups=1, res_expansion=1.0, activation="relu", bias=False, use_xyz=False,
                 normalize="anchor", dim_expansion=[2, 2, 2, 2], pre_blocks=[2, 2, 2, 2], pos_blocks=[2, 2, 2, 2],
                 k_neighbors=[24, 24, 24, 24], reducers=[2, 2, 2, 2], **kwargs)


def pointMLPElite(**kwargs) -> Model:
    """pointMLP.py:366-370."""
    return Model(points=1024, embed_dim=32, groups=1, res_expansion=0.25, activation="relu", bias=False, use_xyz=False,
                 normalize="anchor", dim_expansion=[2, 2, 2, 1], pre_blocks=[1, 1, 2, 1], pos_blocks=[1, 1, 2, 1],
                 k_neighbors=[24, 24, 24, 24], reducers=[2, 2, 2, 2], **kwargs)
