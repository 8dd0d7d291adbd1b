"""Mirror of models/pointnet2/pointnet2.py:40-73 (Pointnet2_Msg) and of the set-abstraction containers of
models/pointnet2/pointnet2_utils.py:161-266: identical constructor signatures and state-dict keys
(sa1.conv_blocks.{i}.{j}, sa1.bn_blocks.{i}.{j}, sa3.mlp_convs.{j}, sa3.mlp_bns.{j}, fc1, bn1, fc2, bn2); the
forward runs ppt_amd.engine.pointnet2_msg_forward (FPS + ball query + grouped-MLP GEMM kernels)."""
import torch
import torch.nn as nn

from ... import engine, graphs, ops


class PointNetSetAbstractionMsg(nn.Module):
    """pointnet2_utils.py:209-226 (parameter container)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for i in range(len(mlp_list)):
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last_channel = in_channel + 3
            for out_channel in mlp_list[i]:
                convs.append(nn.Conv2d(last_channel, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last_channel = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)


class PointNetSetAbstraction(nn.Module):
    """pointnet2_utils.py:161-175 (parameter container; only group_all=True is on the path)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, remove_last=False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.group_all = group_all
        self.remove_last = remove_last


class Pointnet2_Msg(nn.Module):
    """pointnet2.py:40-73.  forward(xyz [B,N,3]) -> [B,256].

    Build-specific knobs: `precision` (bf16 / fp32 parity), `fps_start` = (start1 [B], start2 [B]) and
    `dropout_masks` = (m1 [B,512], m2 [B,256]) to inject the three RNG draws of the path in parity tests."""
    _engine_forward = staticmethod(engine.pointnet2_msg_forward)
    _group_levels = engine.PN2_MSG_LEVELS
    _graph_tag = "pn2_msg"
    _drop_p = (0.4, 0.5)
    # FPS + ball queries are ~a quarter of this encoder's time and use 32 workgroups: worth running ahead even though the
    # encoder is frozen (train.Trainer consults this when the batch is resident)
    group_ahead_pays_when_frozen = True

    def __init__(self, normal_channel=False):
        super().__init__()
        if normal_channel:
            raise NotImplementedError("normal_channel=True is not on the PPT path (pointnet2.py:6-8 default)")
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0, [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = PointNetSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320,
                                             [[64, 64, 128], [128, 128, 256], [128, 128, 256]])
        self.sa3 = PointNetSetAbstraction(None, None, None, 640 + 3, [256, 512, 1024], True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.5)
        self.precision = torch.bfloat16
        self.fps_start = None
        self.dropout_masks = None
        self._wc = None
        self._sd = None
        self._graphs = graphs.GraphCache()
        self.use_hip_graphs = True
        self.group_ahead = None            # stream for the grouping stage of the next iteration (train.Trainer.inputs_ready)
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

    def forward(self, xyz):
        xyz = xyz.contiguous().float()
        B, N, _ = xyz.shape
        if self._wc is None or self._wc.dtype != self.precision:
            self._wc = engine.WeightCache(self.precision)
        if self.fps_start is not None:
            starts = tuple(s.to(xyz.device).contiguous() for s in self.fps_start)
        else:
            starts = (torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device),
                      torch.randint(0, 512, (B,), dtype=torch.long, device=xyz.device))
        masks = None
        if self.dropout_masks is not None:
            masks = tuple(m.to(device=xyz.device, dtype=torch.float32).contiguous() for m in self.dropout_masks)
        elif self.training:
            p1, p2 = self._drop_p
            masks = ((torch.rand((B, 512), device=xyz.device) >= p1).float() / (1.0 - p1),
                     (torch.rand((B, 256), device=xyz.device) >= p2).float() / (1.0 - p2))
        if self._sd is None or self._sd[1] is not self.fc1.weight:
            self._sd = (self.state_dict(keep_vars=True), self.fc1.weight)
        sd, wc, train = self._sd[0], self._wc, self.training
        with torch.no_grad():            # every parameter of this encoder is frozen in PPT (ULIP_models.py:372-389)
            # ~90 shape-static launches: replayed from a hipGraph after graphs.WARMUP_CALLS eager calls (ppt_amd/graphs.py)
            key = (self._graph_tag, tuple(xyz.shape), masks is not None, train, wc.dtype)
            fwd = self._engine_forward
            if xyz.is_cuda and self.use_hip_graphs and ops.profiler is None and self._graphs.ready(key):
                if self.group_ahead is not None:
                    # FPS + ball queries of both levels on the grouping stream (graphs.AheadStage), the rest from here
                    levels = self._group_levels
                    drawn = self.fps_start is None
                    n1 = levels[0][0]

                    def gfn(x, *st):
                        # not injected: the two FPS starts are drawn here, on the grouping stream (the ones drawn above sit on
                        # the caller's stream, behind the previous tower)
                        s0, s1 = st if st else (torch.randint(0, N, (B,), dtype=torch.long, device=x.device),
                                                torch.randint(0, n1, (B,), dtype=torch.long, device=x.device))
                        return tuple(engine.pointnet2_group(x, (s0, s1), levels)), None
                    grouped, slot = self._ahead.run(self._graphs, ("pn2_group", tuple(xyz.shape), drawn), gfn,
                                                    [xyz] + ([] if drawn else [starts[0], starts[1]]), self.group_ahead, vouched=getattr(self, "inputs_vouched", False))
                    ng = len(grouped)
                    ins = list(grouped) + (list(masks) if masks is not None else [])

                    def fn2(*a):
                        m = a[ng:]
                        return (fwd(sd, "", wc, None, None, train, tuple(m) if m else None, grouped=list(a[:ng])),), None
                    (feat,), _ = self._graphs.get(key + ("grouped",), lambda: graphs.GraphedCall(fn2, ins))(*ins)
                    self._ahead.consumed(slot)
                    return feat.clone()
                ins = [xyz, starts[0], starts[1]] + (list(masks) if masks is not None else [])

                def fn(x, s0, s1, *m):
                    return (fwd(sd, "", wc, x, (s0, s1), train, tuple(m) if m else None),), None
                (feat,), _ = self._graphs.get(key, lambda: graphs.GraphedCall(fn, ins))(*ins)
                return feat.clone()
            return fwd(sd, "", wc, xyz, starts, train, masks)


class Pointnet2_Ssg(Pointnet2_Msg):
    """pointnet2.py:6-38: single-scale set abstractions (512 x r0.2 x 32, 128 x r0.4 x 64, group_all), the same FC head
    with Dropout(0.4) twice.  forward(xyz [B,N,3]) -> [B,256]; same knobs as Pointnet2_Msg."""
    _engine_forward = staticmethod(engine.pointnet2_ssg_forward)
    _group_levels = engine.PN2_SSG_LEVELS
    _graph_tag = "pn2_ssg"
    _drop_p = (0.4, 0.4)

    def __init__(self, normal_channel=False):
        nn.Module.__init__(self)
        if normal_channel:
            raise NotImplementedError("normal_channel=True is not on the PPT path (pointnet2.py:7-9 default)")
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=3, mlp=[64, 64, 128], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256],
                                          group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024],
                                          group_all=True, remove_last=True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.precision = torch.bfloat16
        self.fps_start = None
        self.dropout_masks = None
        self._wc = None
        self._sd = None
        self._graphs = graphs.GraphCache()
        self.use_hip_graphs = True
        self.group_ahead = None            # stream for the grouping stage of the next iteration (train.Trainer.inputs_ready)
        self._ahead = graphs.AheadStage()
