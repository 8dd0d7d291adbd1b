"""GPU: every HIP kernel, called through the C ABI (ppt_amd.ops -> libppt_hip.so), against the
oracle (index work: bit-exact) or plain fp32 torch-CPU math (floating point: tolerance stated per test)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from ppt_amd import weights as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ppt_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,N,M,dup", [(4, 1024, 512, False), (2, 2048, 512, True), (1, 8192, 512, False),
                                        (3, 512, 128, False), (2, 100, 100, False), (2, 64, 70, True),
                                        (1, 16384, 64, False), (5, 1000, 333, False)])
def test_fps_bit_exact(ops, B, N, M, dup):
    pc, start = W.synth_clouds(B, N, seed=1234, duplicates=dup)
    ref = O.fps(pc, M, start)
    idx, ctr = ops.fps(dev(pc), M, dev(start))
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(ctr.cpu().numpy(), np.take_along_axis(pc, ref[:, :, None], axis=1))


def test_fps_golden(ops):
    g = np.load(os.path.join(G, "g_index.npz"))
    for tag, (B, N, dup) in {"a": (4, 1024, False), "b": (2, 2048, True), "c": (1, 8192, False)}.items():
        pc, start = W.synth_clouds(B, N, seed=1234, duplicates=dup)
        idx, _ = ops.fps(dev(pc), 512, dev(start))
        assert np.array_equal(idx.cpu().numpy(), g[f"fps_{tag}_idx"].astype(np.int64))


def test_fps_all_duplicate_points(ops):
    pc = np.tile(np.float32([[0.25, -0.5, 0.125]]), (2, 300, 1))
    start = np.array([7, 299], np.int64)
    idx, _ = ops.fps(dev(pc), 20, dev(start))
    assert np.array_equal(idx.cpu().numpy(), O.fps(pc, 20, start))


# ------------------------------------------------------------------ kNN / group
@pytest.mark.parametrize("B,N,G_,k,dup", [(4, 1024, 512, 32, False), (2, 2048, 512, 32, True), (1, 8192, 512, 32, False),
                                           (2, 2048, 256, 4, True), (3, 200, 50, 8, False), (2, 64, 64, 64, False),
                                           (1, 33, 7, 5, False)])
def test_knn_group_bit_exact(ops, B, N, G_, k, dup):
    pc, start = W.synth_clouds(B, N, seed=99, duplicates=dup)
    cidx = O.fps(pc, G_, start)
    nbr_ref, nb_ref, ce_ref = O.group(pc, cidx, k)
    idx, nb = ops.knn_group(dev(pc), dev(ce_ref), k)
    assert np.array_equal(idx.cpu().numpy(), nbr_ref)        # same total order (distance, index)
    assert np.array_equal(nb.cpu().numpy(), nb_ref)


def test_knn_golden_sets(ops):
    g = np.load(os.path.join(G, "g_index.npz"))
    pc, _ = W.synth_clouds(4, 1024, seed=1234)
    cidx = g["fps_a_idx"].astype(np.int64)
    center = np.take_along_axis(pc, cidx[:, :, None], axis=1)
    idx, _ = ops.knn_group(dev(pc), dev(center), 32)
    assert np.array_equal(np.sort(idx.cpu().numpy(), -1), g["knn_a_k32"].astype(np.int64))


def test_knn_degenerate_overflow_path(ops):
    """hundreds of exact ties overflow the survivor list -> exact fallback path."""
    rng = np.random.default_rng(3)
    pc = rng.random((2, 1024, 3), dtype=np.float32)
    pc[0, 100:900] = pc[0, 100]           # 800 copies of one point
    pc[1, :] = pc[1, 0]                   # every point identical
    center = pc[:, [100, 5, 950, 0]].copy()
    nbr_ref, _ = O.knn(pc, center, 32)
    idx, nb = ops.knn_group(dev(pc), dev(center), 32)
    assert np.array_equal(idx.cpu().numpy(), nbr_ref)


@pytest.mark.parametrize("N,r,K", [(1024, 0.1, 16), (1024, 0.2, 32), (1024, 0.4, 128), (8192, 0.2, 32), (8192, 0.4, 128)])
def test_ball_query(ops, N, r, K):
    pc, start = W.synth_clouds(2, N, seed=5)
    cidx = O.fps(pc, 128, start)
    center = np.take_along_axis(pc, cidx[:, :, None], axis=1)
    ref = O.ball_query(pc, center, r, K)
    out = ops.ball_query(dev(pc), dev(center), r, K)
    assert np.array_equal(out.cpu().numpy(), ref)


# ------------------------------------------------------------------ GEMM
def _gemm_case(ops, dtype, M, N, K, tol, **kw):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    a, b = dev(A, dtype), dev(Bm, dtype)
    ref = a.float().cpu() @ b.float().cpu().t()
    out = ops.gemm(a, b, out_dtype=torch.float32, **kw)
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err < tol, err
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 384), (513, 1152, 384), (1000, 40, 512), (32, 512, 768),
                                    (77, 2048, 512), (130, 130, 136), (4104, 384, 1536)])
def test_gemm_bf16_plain(ops, M, N, K):
    # operands are exactly representable (bf16-rounded before the fp32 reference) -> only fp32
    # accumulation-order error remains: 1e-3 absolute on O(1) outputs is generous
    _gemm_case(ops, torch.bfloat16, M, N, K, 2e-3)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (513, 384, 384), (77, 512, 2048), (130, 136, 132)])
def test_gemm_f32_plain(ops, M, N, K):
    _gemm_case(ops, torch.float32, M, N, K, 2e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_epilogues(ops, dtype):
    rng = np.random.default_rng(11)
    M, N, K = 640, 384, 256
    A = dev(rng.standard_normal((M, K)).astype(np.float32), dtype)
    Bm = dev((rng.standard_normal((N, K)) / 16).astype(np.float32), dtype)
    bias = dev(rng.standard_normal(N).astype(np.float32))
    res = dev(rng.standard_normal((M, N)).astype(np.float32))
    res2 = dev(rng.standard_normal((M, N)).astype(np.float32))
    rs = dev(rng.random(M // 64).astype(np.float32))
    gadd = dev(rng.standard_normal((M // 32, N)).astype(np.float32))
    base = A.float().cpu() @ Bm.float().cpu().t()
    tol = 3e-3 if dtype == torch.bfloat16 else 5e-5
    for act, fn in ((ops.ACT_NONE, lambda x: x), (ops.ACT_RELU, torch.relu), (ops.ACT_GELU, O.gelu_erf),
                    (ops.ACT_QUICKGELU, O.quick_gelu)):
        out = ops.gemm(A, Bm, out_dtype=torch.float32, bias=bias, act=act, row_scale=rs, row_scale_rows=64,
                       residual=res, residual2=res2, group_add=gadd, group_rows=32)
        ref = fn(base + bias.cpu() + gadd.cpu().repeat_interleave(32, 0)) * rs.cpu().repeat_interleave(64)[:, None] \
            + res.cpu() + res2.cpu()
        assert (out.cpu() - ref).abs().max().item() < tol, act
    # derivative epilogue
    pre = dev(rng.standard_normal((M, N)).astype(np.float32), dtype)
    for act in (ops.ACT_GELU, ops.ACT_QUICKGELU, ops.ACT_RELU):
        out = ops.gemm(A, Bm, out_dtype=torch.float32, act=act, dact_pre=pre)
        x = pre.float().cpu().requires_grad_(True)
        f = {ops.ACT_GELU: O.gelu_erf, ops.ACT_QUICKGELU: O.quick_gelu, ops.ACT_RELU: torch.relu}[act](x)
        (d,) = torch.autograd.grad(f.sum(), x)
        assert (out.cpu() - base * d).abs().max().item() < tol * 2, act
    # second output, pooled max, column statistics
    out2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    pool = torch.empty((M // 32, N), dtype=torch.float32, device="cuda")
    cs = torch.empty((M // 32, N), dtype=torch.float32, device="cuda")
    cq = torch.empty_like(cs)
    out = ops.gemm(A, Bm, out_dtype=torch.float32, bias=bias, out2=out2, pool_max=pool, col_stats=(cs, cq))
    ref = base + bias.cpu()
    assert (out.cpu() - ref).abs().max().item() < tol
    assert (out2.float().cpu() - ref).abs().max().item() < 2e-2
    assert (pool.cpu() - ref.view(M // 32, 32, N).max(1)[0]).abs().max().item() < tol
    assert (cs.cpu().sum(0) - ref.sum(0)).abs().max().item() < tol * M
    chunks = ref.view(M // 32, 32, N)
    m2 = ((chunks - chunks.mean(1, keepdim=True)) ** 2).sum(1)
    assert (cq.cpu() - m2).abs().max().item() < tol * 32
    g1 = dev(np.ones(N, np.float32)); b0 = dev(np.zeros(N, np.float32))
    sc, sh = ops.bn_finalize(g1, b0, True, partials=(cs, cq), rows_per_partial=32, count=M, update_running=False)
    assert (sc.cpu() - 1 / torch.sqrt(ref.var(0, unbiased=False) + 1e-5)).abs().max().item() < 1e-3


@pytest.mark.parametrize("M,N,K,group_rows,pool_rows,stats,act", [
    (640, 384, 256, 0, 0, False, "gelu"),          # 64x64 LDS-DMA tiles, bias + GELU
    (1000, 136, 64, 16, 0, False, "none"),         # ragged M and N, 16-row groups
    (650, 256, 128, 32, 32, True, "none"),         # register-staged 128x128: group term + statistics + pool, ragged M
    (1040, 128, 64, 64, 16, True, "relu"),         # 16-row pools (max and min)
    (1088, 128, 64, 0, 64, False, "quickgelu"),    # 64-row pools
    (12800, 1024, 96, 0, 0, False, "gelu"),        # >= 768 tiles of 128x128: half-slab LDS-DMA kernel
    (25000, 512, 64, 32, 32, True, "none"),        # the same kernel with the conv3 epilogue, ragged M
])
def test_gemm_register_epilogue(ops, M, N, K, group_rows, pool_rows, stats, act):
    """bf16 C (or none) with per-column / per-row-group epilogue terms: the MFMA-layout epilogue of gemm.hip."""
    rng = np.random.default_rng(M + N + K)
    A = dev(rng.standard_normal((M, K)).astype(np.float32), torch.bfloat16)
    Bm = dev((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32), torch.bfloat16)
    bias = dev(rng.standard_normal(N).astype(np.float32))
    kw = dict(bias=bias)
    ref = A.float().cpu() @ Bm.float().cpu().t() + bias.cpu()
    if group_rows:
        ng = (M + group_rows - 1) // group_rows
        gadd = dev(rng.standard_normal((ng, N)).astype(np.float32))
        kw.update(group_add=gadd, group_rows=group_rows)
        ref = ref + gadd.cpu().repeat_interleave(group_rows, 0)[:M]
    pre = ref
    code, fn = {"none": (ops.ACT_NONE, lambda x: x), "relu": (ops.ACT_RELU, torch.relu), "gelu": (ops.ACT_GELU, O.gelu_erf),
                "quickgelu": (ops.ACT_QUICKGELU, O.quick_gelu)}[act]
    ref = fn(ref)
    if stats:
        nchunk = (M + 31) // 32
        cs = torch.full((nchunk, N), float("nan"), device="cuda"); cq = torch.full_like(cs, float("nan"))
        kw["col_stats"] = (cs, cq)
    if pool_rows:
        npool = (M + pool_rows - 1) // pool_rows
        pmax = torch.full((npool, N), float("nan"), device="cuda"); pmin = torch.full_like(pmax, float("nan"))
        kw.update(pool_max=pmax, pool_min=pmin, pool_rows=pool_rows)
    out = ops.gemm(A, Bm, out_dtype=torch.bfloat16, act=code, **kw)
    torch.cuda.synchronize()
    assert torch.allclose(out.float().cpu(), ref, rtol=1e-2, atol=1e-2)
    if stats:                                        # statistics are taken before the activation, per 32-row chunk
        for c0 in (0, nchunk // 2, nchunk - 1):
            rows = pre[c0 * 32:(c0 + 1) * 32]
            assert torch.allclose(cs[c0].cpu(), rows.sum(0), rtol=1e-4, atol=1e-3)
            assert torch.allclose(cq[c0].cpu(), ((rows - rows.mean(0)) ** 2).sum(0), rtol=1e-3, atol=1e-3)
        nfull = M // 32                               # every whole chunk (a rare wrong lane group must not hide)
        chunks = pre[:nfull * 32].view(nfull, 32, N)
        assert torch.allclose(cs[:nfull].cpu(), chunks.sum(1), rtol=1e-4, atol=2e-3)
        assert torch.allclose(cq[:nfull].cpu(), ((chunks - chunks.mean(1, keepdim=True)) ** 2).sum(1), rtol=1e-3, atol=2e-3)
        # the launch is bit-reproducible
        cs2 = torch.empty_like(cs); cq2 = torch.empty_like(cq)
        kw3 = dict(kw); kw3["col_stats"] = (cs2, cq2)
        for _ in range(3):
            out_b = ops.gemm(A, Bm, out_dtype=torch.bfloat16, act=code, **kw3)
            torch.cuda.synchronize()
            assert torch.equal(out_b, out) and torch.equal(cs2, cs) and torch.equal(cq2, cq)
    if pool_rows:
        for g0 in (0, npool // 2, npool - 1):
            rows = ref[g0 * pool_rows:(g0 + 1) * pool_rows]
            assert torch.allclose(pmax[g0].cpu(), rows.max(0)[0], rtol=1e-5, atol=1e-5)
            assert torch.allclose(pmin[g0].cpu(), rows.min(0)[0], rtol=1e-5, atol=1e-5)
        full = (M // pool_rows) * pool_rows
        assert torch.allclose(pmax[:M // pool_rows].cpu(), ref[:full].view(-1, pool_rows, N).max(1)[0], rtol=1e-5, atol=1e-5)
    # the same launch without a C: pooled / statistics-only GEMMs (conv4)
    if pool_rows:
        pm2 = torch.empty_like(pmax)
        kw2 = dict(kw); kw2.update(pool_max=pm2, pool_min=None); kw2.pop("col_stats", None)
        ops.gemm(A, Bm, act=code, want_out=False, **kw2)
        torch.cuda.synchronize()
        assert torch.equal(pm2, pmax)


@pytest.mark.parametrize("dtype,M,N,K", [(torch.bfloat16, 14080, 384, 96), (torch.float32, 14100, 392, 48),
                                          (torch.bfloat16, 16416, 384, 1536)])
def test_gemm_residual_epilogue_full_size(ops, dtype, M, N, K):
    """proj / fc2 at full size: bias, DropPath row scale, fp32 residual stream in and out, the next block's + pos
    (the LDS-walk epilogue of the 64x64 LDS-DMA kernel), and the GELU-derivative epilogue; bit-reproducible."""
    rng = np.random.default_rng(M + N)
    A = dev(rng.standard_normal((M, K)).astype(np.float32), dtype)
    Bm = dev((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32), dtype)
    bias = dev(rng.standard_normal(N).astype(np.float32))
    res = dev(rng.standard_normal((M, N)).astype(np.float32))
    res2 = dev(rng.standard_normal((M, N)).astype(np.float32))
    rows = 470
    rs = dev(rng.random((M + rows - 1) // rows).astype(np.float32))
    out = ops.gemm(A, Bm, out_dtype=torch.float32, bias=bias, row_scale=rs, row_scale_rows=rows, residual=res, residual2=res2)
    ref = (A.float().cpu() @ Bm.float().cpu().t() + bias.cpu()) * rs.cpu().repeat_interleave(rows)[:M, None] + res.cpu() + res2.cpu()
    tol = 3e-3 if dtype == torch.bfloat16 else 5e-5
    assert (out.cpu() - ref).abs().max().item() < tol * (1 if K < 512 else 4)
    out_b = ops.gemm(A, Bm, out_dtype=torch.float32, bias=bias, row_scale=rs, row_scale_rows=rows, residual=res, residual2=res2)
    torch.cuda.synchronize()
    assert torch.equal(out, out_b)
    # derivative epilogue with a bf16 / fp32 saved pre-activation
    pre = dev(rng.standard_normal((M, N)).astype(np.float32), dtype)
    d = ops.gemm(A, Bm, out_dtype=dtype, act=ops.ACT_GELU, dact_pre=pre)
    x = pre.float().cpu().requires_grad_(True)
    (g,) = torch.autograd.grad(O.gelu_erf(x).sum(), x)
    want = (A.float().cpu() @ Bm.float().cpu().t()) * g
    assert torch.allclose(d.float().cpu(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("S,K", [(4, 2048), (3, 1536)])
def test_splitk_linear_with_layernorm_consumers(ops, S, K):
    """The prompt chain's skinny long-K linears as split-K launches (ops.gemm_splitk: S fp32 partial products from ONE batched
    ppt_gemm) whose reduction rides in the LayerNorm that reads the result: ppt_layernorm_fwd_sum (x + bias + sum of the slices ->
    LayerNorm, the summed row written back) and ppt_layernorm_bwd_sum (dy = sum of the slices).  Against the single-launch linear +
    the plain LayerNorm kernels: the same values up to the fp32 summation order; bit-reproducible."""
    M, D = 817, 512
    rng = np.random.default_rng(S * 7 + K)
    A = dev(rng.standard_normal((M, K)).astype(np.float32), torch.float16)
    Wt = dev((rng.standard_normal((D, K)) / math.sqrt(K)).astype(np.float32), torch.float16)
    bias = dev(rng.standard_normal(D).astype(np.float32))
    x = dev(rng.standard_normal((M, D)).astype(np.float32))
    g = dev(1.0 + 0.1 * rng.standard_normal(D).astype(np.float32))
    b = dev(0.1 * rng.standard_normal(D).astype(np.float32))
    parts = ops.gemm_splitk(A, Wt, S)
    assert parts.shape == (S, M, D)
    full = ops.gemm(A, Wt, out_dtype=torch.float32)
    assert torch.allclose(parts.sum(0), full, rtol=1e-5, atol=2e-5)
    for z in range(S):                                  # every slice is the product over ITS K range
        ref = A[:, z * (K // S):(z + 1) * (K // S)].float() @ Wt[:, z * (K // S):(z + 1) * (K // S)].float().t()
        assert torch.allclose(parts[z], ref, rtol=1e-4, atol=1e-4)
    # forward consumer
    xs = torch.empty_like(x)
    y, mean, rstd = ops.layernorm_fwd_sum(x, bias, parts, g, b, torch.float16, write_xs=xs, save_stats=True)
    want = x + bias
    for z in range(S):
        want = want + parts[z]
    assert torch.equal(xs, want)                        # the summation order is the documented one
    y_ref, mean_ref, rstd_ref = ops.layernorm_fwd(want, g, b, torch.float16, save_stats=True)
    assert torch.equal(y, y_ref) and torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)
    one = ops.gemm(A, Wt, out_dtype=torch.float32, bias=bias, residual=x)            # the single-launch linear it replaces
    assert torch.allclose(xs, one, rtol=1e-5, atol=3e-5)
    xs2 = torch.empty_like(x)
    y2, _, _ = ops.layernorm_fwd_sum(x, bias, parts, g, b, torch.float16, write_xs=xs2, save_stats=True)
    assert torch.equal(y, y2) and torch.equal(xs, xs2)
    # backward consumer
    dx0 = dev(rng.standard_normal((M, D)).astype(np.float32))
    dx_a, cp_a = ops.layernorm_bwd_sum(parts, want, g, mean, rstd, dx0.clone(), accumulate=True, copy_dtype=torch.float16)
    dy = parts[0].clone()
    for z in range(1, S):
        dy = dy + parts[z]
    dx_b, _, _, cp_b = ops.layernorm_bwd(dy, want, g, mean, rstd, dx=dx0.clone(), accumulate=True, copy_dtype=torch.float16)
    assert torch.equal(dx_a, dx_b) and torch.equal(cp_a, cp_b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,kind", [
    (16416, 1152, 384, "plain"),        # qkv of a C2 batch: ragged M (64.125 row tiles), ragged N for 256-wide tiles (4.5)
    (16416, 1536, 384, "gelu"),         # fc1: bias + GELU, 16-bit C
    (16416, 384, 1536, "residual"),     # fc2: fp32 residual stream in place, DropPath row scale, + pos; 256 x 128 tiles
    (8200, 384, 384, "residual"),       # proj-shaped, ragged last row tile of 8 rows
    (9000, 640, 64, "dact"),            # derivative epilogue with a saved pre-activation + second (pre) output; K = 2 half-slabs
    (16384, 512, 256, "stats"),         # conv3-shaped: per-group term + BatchNorm chunk statistics
    (4096, 4096, 4096, "plain"),        # square, 256 tiles = one per CU
    (3000, 256, 96, "batched"),         # blockIdx.z batches with strides
])
def test_gemm256_core_is_the_tile_loop_bit_for_bit(ops, dtype, M, N, K, kind):
    """csrc/gemm256.hip (256 x 256 / 256 x 128 macro-tiles, 8 waves, LDS-DMA ring) against the 64 x 64 / 128 x 128 tile loops of
    gemm.hip on the same launch: both walk K in ascending 16-wide MFMA steps into fp32 accumulators and share the epilogue code, so
    the results are the SAME BITS; plus an fp32 reference on the operands' 16-bit values; plus run-to-run reproducibility."""
    from ppt_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    A = dev(rng.standard_normal((M, K)).astype(np.float32), dtype)
    Bm = dev((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32), dtype)
    bias = dev(rng.standard_normal(N).astype(np.float32))
    kw, out_dtype = {}, dtype
    base = None
    if kind == "gelu":
        kw = dict(bias=bias, act=ops.ACT_GELU)
    elif kind == "residual":
        res = dev(rng.standard_normal((M, N)).astype(np.float32))
        res2 = dev(rng.standard_normal((M, N)).astype(np.float32))
        rs = dev(rng.random((M + 512) // 513).astype(np.float32))
        kw = dict(bias=bias, row_scale=rs, row_scale_rows=513, residual=res, residual2=res2)
        out_dtype = torch.float32
    elif kind == "dact":
        pre = dev(rng.standard_normal((M, N)).astype(np.float32), dtype)
        kw = dict(act=ops.ACT_GELU, dact_pre=pre)
    elif kind == "stats":
        gadd = dev(rng.standard_normal((M // 32, N)).astype(np.float32))
        kw = dict(bias=bias, group_add=gadd, group_rows=32)

    def launch(core):
        k2 = dict(kw)
        extra = {}
        if kind == "stats":
            cs = torch.full(((M + 31) // 32, N), float("nan"), device="cuda"); cq = torch.full_like(cs, float("nan"))
            k2["col_stats"] = (cs, cq)
            extra = dict(cs=cs, cq=cq)
        if kind == "dact":
            o2 = torch.empty((M, N), dtype=torch.float32, device="cuda")
            k2.update(out2=o2)
            extra = dict(o2=o2)
        if kind == "batched":
            Mb = M // 3
            out = torch.full((3 * Mb, N), float("nan"), dtype=out_dtype, device="cuda")
            ops.gemm(A[:Mb], Bm, out=out, M=Mb, batch=3, strideA=Mb * K, strideB=0, strideC=Mb * N, core=core, **k2)
        else:
            out = ops.gemm(A, Bm, out_dtype=out_dtype, core=core, **k2)
        torch.cuda.synchronize()
        return out, extra
    lib.ppt_set_gemm256(0)
    try:
        old, old_x = launch(None)
        new, new_x = launch("256")
        again, _ = launch("256")
    finally:
        lib.ppt_set_gemm256(-1)
    assert torch.equal(new, again), "not reproducible"
    assert torch.equal(old, new), (old.float() - new.float()).abs().max().item()
    for k in old_x:
        assert torch.equal(old_x[k], new_x[k]), k
    auto, _ = launch(None)                               # ... and ppt_gemm routes problems of this size here by itself
    assert torch.equal(auto, new)
    # fp32 reference on a window of rows (first tile, the ragged last tile)
    for rs_ in (slice(0, 512), slice(max(0, (M // 3 if kind == "batched" else M) - 300), M // 3 if kind == "batched" else M)):
        ref = A[rs_].float() @ Bm.float().t()
        if kind in ("gelu", "residual", "stats"):
            ref = ref + bias
        if kind == "stats":
            ref = ref + gadd.repeat_interleave(32, 0)[rs_]
        if kind == "gelu":
            ref = torch.nn.functional.gelu(ref)
        if kind == "residual":
            ref = ref * rs.repeat_interleave(513)[:M][rs_, None] + res[rs_] + res2[rs_]
        if kind == "dact":
            x = pre[rs_].float().requires_grad_(True)
            (gd,) = torch.autograd.grad(torch.nn.functional.gelu(x).sum(), x)
            ref = ref * gd
        tol = 4e-2 if dtype == torch.bfloat16 else 1e-2
        assert torch.allclose(new[rs_].float(), ref, rtol=tol, atol=tol * (2 if K > 1024 else 1)), (new[rs_].float() - ref).abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_prologues(ops, dtype):
    rng = np.random.default_rng(12)
    M, N, K = 512, 256, 128
    Bm = dev((rng.standard_normal((N, K)) / 11).astype(np.float32), dtype)
    scale = dev((1 + 0.2 * rng.standard_normal(K)).astype(np.float32))
    shift = dev((0.3 * rng.standard_normal(K)).astype(np.float32))
    tol = 2e-2 if dtype == torch.bfloat16 else 5e-5
    # affine + relu on A (BatchNorm -> ReLU fused into the consumer GEMM)
    A = dev(rng.standard_normal((M, K)).astype(np.float32), dtype)
    out = ops.gemm(A, Bm, out_dtype=torch.float32, a_mode=ops.A_AFFINE_RELU, a_scale=scale, a_shift=shift)
    a = torch.relu(A.float().cpu() * scale.cpu() + shift.cpu())
    if dtype == torch.bfloat16:
        a = a.to(torch.bfloat16).float()
    ref = a @ Bm.float().cpu().t()
    assert (out.cpu() - ref).abs().max().item() < tol
    # conv1 (K=3) + BN + ReLU producer
    pts = dev(rng.standard_normal((M, 3)).astype(np.float32))
    w1 = dev(rng.standard_normal((K, 3)).astype(np.float32))
    b1 = dev(rng.standard_normal(K).astype(np.float32))
    out = ops.gemm(None, Bm, out_dtype=torch.float32, a_mode=ops.A_CONV1, pts=pts, w1=w1, b1=b1, a_scale=scale,
                   a_shift=shift)
    a = torch.relu((pts.cpu() @ w1.cpu().t() + b1.cpu()) * scale.cpu() + shift.cpu())
    if dtype == torch.bfloat16:
        a = a.to(torch.bfloat16).float()      # the producer rounds the operand to bf16, as the kernel does
    ref = a @ Bm.float().cpu().t()
    assert (out.cpu() - ref).abs().max().item() < tol


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,D", [(513 * 2, 384), (77 * 3, 512), (5, 64), (9, 1000), (64 * 513, 384)])
def test_layernorm_fwd_bwd(ops, M, D):
    rng = np.random.default_rng(M + D)
    x = rng.standard_normal((M, D)).astype(np.float32) * 2 + 0.5
    add = rng.standard_normal((M, D)).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    dy = rng.standard_normal((M, D)).astype(np.float32)
    xs_t = (torch.from_numpy(x) + torch.from_numpy(add)).requires_grad_(True)
    wt, bt = torch.from_numpy(w).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    ref = O.layer_norm(xs_t, wt, bt)
    ref.backward(torch.from_numpy(dy))
    xd = dev(x)
    y, mean, rstd = ops.layernorm_fwd(xd, dev(w), dev(b), torch.float32, add=dev(add), write_xs=xd, save_stats=True)
    assert (y.cpu() - ref.detach()).abs().max().item() < 2e-5
    assert (xd.cpu() - xs_t.detach()).abs().max().item() == 0
    dx0 = dev(np.ones((M, D), np.float32))
    dx, dw, db = ops.layernorm_bwd(dev(dy), xd, dev(w), mean, rstd, dx=dx0, accumulate=True, want_wgrad=True,
                                   partial_rows=64)
    assert (dx.cpu() - 1 - xs_t.grad).abs().max().item() < 5e-5
    assert (dw.cpu() - wt.grad).abs().max().item() < 1e-3 * max(1.0, M / 1000)
    assert (db.cpu() - bt.grad).abs().max().item() < 1e-3 * max(1.0, M / 1000)
    # default partial rows (what the engine uses) + the 16-bit operand copy of dx that the next GEMM reads
    dx1, dw1, db1, cp = ops.layernorm_bwd(dev(dy), xd, dev(w), mean, rstd, want_wgrad=True, copy_dtype=torch.float16)
    assert (dx1.cpu() - xs_t.grad).abs().max().item() < 5e-5 and torch.equal(cp, dx1.half())
    assert ((dw1.cpu() - wt.grad).norm() / wt.grad.norm()).item() < 1e-5 and ((db1.cpu() - bt.grad).norm() / bt.grad.norm()).item() < 1e-5
    # bf16 output + broadcast positional table
    tab = rng.standard_normal((7, D)).astype(np.float32)
    y2, _, _ = ops.layernorm_fwd(dev(x), dev(w), dev(b), torch.bfloat16, add=dev(tab), add_rows=7)
    ref2 = O.layer_norm(torch.from_numpy(x) + torch.from_numpy(tab).repeat((M + 6) // 7, 1)[:M], wt.detach(), bt.detach())
    assert (y2.float().cpu() - ref2).abs().max().item() < 3e-2


# ------------------------------------------------------------------ attention
def _attn_ref(qkv, Bt, T, H, scale, causal):
    q, k, v = qkv.view(Bt, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = (q @ k.transpose(-2, -1)) * scale
    if causal:
        a = a + torch.full((T, T), float("-inf")).triu(1)
    a = torch.softmax(a, -1)
    return (a @ v).transpose(1, 2).reshape(Bt * T, H * 64)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("Bt,T,H,causal", [(2, 513, 6, False), (3, 77, 8, True), (1, 64, 1, False), (2, 100, 2, True)])
def test_attention_fwd_bwd(ops, dtype, tol, Bt, T, H, causal):
    rng = np.random.default_rng(T + H)
    qkv = rng.standard_normal((Bt * T, 3 * H * 64)).astype(np.float32)
    qd = dev(qkv, dtype)
    q32 = qd.float().cpu().requires_grad_(True)
    ref = _attn_ref(q32, Bt, T, H, 0.125, causal)
    out, lse = ops.attention_fwd(qd, Bt, T, H, 0.125, causal)
    assert (out.float().cpu() - ref.detach()).abs().max().item() < tol
    dout = rng.standard_normal((Bt * T, H * 64)).astype(np.float32)
    dd = dev(dout, dtype)
    ref.backward(dd.float().cpu())
    dqkv = ops.attention_bwd(qd, out, dd, lse, Bt, T, H, 0.125, causal)
    err = (dqkv.float().cpu() - q32.grad).abs().max().item()
    assert err < tol * 4, err


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("Bt,T,H", [(22, 513, 6), (11, 385, 12)])
def test_attention_fwd_lds_resident_vit_kernel(ops, dtype, tol, Bt, T, H):
    """attn_fwd_resident (T = 64 n + 1, >= 128 (batch, head) pairs: K / V of a pair resident in LDS, the class-token query row
    on the vector ALU) against the fp32 reference -- every row, the last one in particular -- with the softmax statistics, and
    against the streaming kernel the other shapes use (PPT_ATTN_RESIDENT=0 cannot be flipped in-process: a sub-batch below
    the 128-pair threshold takes the streaming kernel)."""
    rng = np.random.default_rng(T + H)
    qkv = rng.standard_normal((Bt * T, 3 * H * 64)).astype(np.float32)
    qkv[:, :H * 64] *= 2.0                                                # sharper softmax than unit-variance scores
    qd = dev(qkv, dtype)
    out, lse = ops.attention_fwd(qd, Bt, T, H, 0.125, False)
    q32 = qd.float().cpu()
    ref = _attn_ref(q32, Bt, T, H, 0.125, False)
    err = (out.float().cpu() - ref).abs()
    assert err.max().item() < tol, err.max().item()
    assert err.view(Bt, T, -1)[:, T - 1].max().item() < tol                # the vector-ALU row
    q, k, _ = q32.view(Bt, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp((q @ k.transpose(-2, -1)) * 0.125, -1)       # [Bt, H, T]
    assert (lse.cpu().view(Bt, H, T) - lse_ref).abs().max().item() < 2e-3
    nb = 120 // H                                                          # < 128 pairs: the streaming kernel, same rows
    out_s, lse_s = ops.attention_fwd(qd[:nb * T].contiguous(), nb, T, H, 0.125, False)
    assert (out[:nb * T].float() - out_s.float()).abs().max().item() < tol / 2
    assert (lse.view(Bt, H, T)[:nb] - lse_s.view(nb, H, T)).abs().max().item() < 1e-4


# ------------------------------------------------------------------ small ops
def test_conv1_stats_bn_finalize(ops):
    rng = np.random.default_rng(21)
    M, C = 2048 * 3 + 100, 128
    pts = rng.standard_normal((M, 3)).astype(np.float32) * 0.1
    w1 = rng.standard_normal((C, 3)).astype(np.float32)
    b1 = rng.standard_normal(C).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    be = (0.1 * rng.standard_normal(C)).astype(np.float32)
    rm = rng.standard_normal(C).astype(np.float32)
    rv = (1 + rng.random(C)).astype(np.float32)
    y = torch.from_numpy(pts) @ torch.from_numpy(w1).t() + torch.from_numpy(b1)
    mean, var = y.mean(0), y.var(0, unbiased=False)
    ps, pq, rpp = ops.conv1_stats(dev(pts), dev(w1), dev(b1))
    rmd, rvd = dev(rm), dev(rv)
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    sc, sh = ops.bn_finalize(dev(g), dev(be), True, partials=(ps, pq), rows_per_partial=rpp, count=M, running_mean=rmd, running_var=rvd,
                             num_batches_tracked=nbt)
    sc_ref = torch.from_numpy(g) / torch.sqrt(var + 1e-5)
    assert (sc.cpu() - sc_ref).abs().max().item() < 1e-4 * sc_ref.abs().max().item()
    sh_ref = torch.from_numpy(be) - mean * sc_ref
    assert (sh.cpu() - sh_ref).abs().max().item() < 1e-4 * sh_ref.abs().max().item()
    assert (rmd.cpu() - (0.9 * torch.from_numpy(rm) + 0.1 * mean)).abs().max().item() < 1e-5
    assert (rvd.cpu() - (0.9 * torch.from_numpy(rv) + 0.1 * y.var(0, unbiased=True))).abs().max().item() < 1e-5
    assert nbt.item() == 1
    sc2, sh2 = ops.bn_finalize(dev(g), dev(be), False, running_mean=dev(rm), running_var=dev(rv))
    ref = torch.from_numpy(g) / torch.sqrt(torch.from_numpy(rv) + 1e-5)
    assert (sc2.cpu() - ref).abs().max().item() < 1e-6
    assert (sh2.cpu() - (torch.from_numpy(be) - torch.from_numpy(rm) * ref)).abs().max().item() < 1e-6


@pytest.mark.parametrize("P,C,rpp,tail", [(2048, 512, 32, 0), (5000, 96, 32, 7), (300, 512, 32, 0), (2500, 40, 64, 13)])
def test_bn_finalize_many_partials(ops, P, C, rpp, tail):
    """(sum, M2) chunk partials -> scale / shift / running statistics: the two-stage fold (>= 2048 partials, workspace)
    and the single-kernel one give the statistics of the underlying rows; large mean / small spread included."""
    rng = np.random.default_rng(P + C)
    count = P * rpp - (rpp - tail if tail else 0)
    rows = (rng.standard_normal((count, C)) * rng.random(C) * 3 + rng.standard_normal(C) * 50).astype(np.float32)
    x = torch.from_numpy(rows).double()
    ps = torch.zeros((P, C), dtype=torch.float32); pq = torch.zeros((P, C), dtype=torch.float32)
    for i in range(P):
        ch = x[i * rpp:min(count, (i + 1) * rpp)]
        ps[i] = ch.sum(0).float(); pq[i] = ((ch - ch.mean(0)) ** 2).sum(0).float()
    g = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32); be = (0.1 * rng.standard_normal(C)).astype(np.float32)
    rm = dev(rng.standard_normal(C).astype(np.float32)); rv = dev((1 + rng.random(C)).astype(np.float32))
    rm0, rv0 = rm.cpu().clone(), rv.cpu().clone()
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    sc, sh = ops.bn_finalize(dev(g), dev(be), True, partials=(ps.cuda(), pq.cuda()), rows_per_partial=rpp, count=count,
                             running_mean=rm, running_var=rv, num_batches_tracked=nbt)
    mean, var = x.mean(0), x.var(0, unbiased=False)
    sc_ref = torch.from_numpy(g).double() / torch.sqrt(var + 1e-5)
    assert torch.allclose(sc.cpu().double(), sc_ref, rtol=2e-5, atol=1e-6)
    assert torch.allclose(sh.cpu().double(), torch.from_numpy(be).double() - mean * sc_ref, rtol=2e-5, atol=2e-4)
    assert torch.allclose(rm.cpu().double(), 0.9 * rm0.double() + 0.1 * mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rv.cpu().double(), 0.9 * rv0.double() + 0.1 * x.var(0, unbiased=True), rtol=1e-5, atol=1e-5)
    assert nbt.item() == 1
    sc_b, sh_b = ops.bn_finalize(dev(g), dev(be), True, partials=(ps.cuda(), pq.cuda()), rows_per_partial=rpp, count=count)
    assert torch.equal(sc_b, sc) and torch.equal(sh_b, sh)


@pytest.mark.parametrize("M,C,training", [(1000, 96, True), (4096, 384, True), (130, 40, False)])
def test_batch_norm_relu_rows_matches_torch(ops, M, C, training):
    """relu(BatchNorm1d(x)) over rows with trainable gamma / beta: forward, running statistics, dx, d gamma, d beta."""
    from ppt_amd.autograd import batch_norm_relu_rows
    torch.manual_seed(M + C)
    x = (torch.randn(M, C) * 2 + torch.randn(C) * 3).cuda().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C).cuda()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(C)); bn.bias.copy_(0.3 * torch.randn(C))
        bn.running_mean.copy_(torch.randn(C)); bn.running_var.copy_(1 + torch.rand(C))
    ref = torch.nn.BatchNorm1d(C).cuda()
    ref.load_state_dict(bn.state_dict())
    bn.train(training); ref.train(training)
    w = torch.randn(M, C, device="cuda")
    y = batch_norm_relu_rows(x, bn, training)
    (y * w).sum().backward()
    xr = x.detach().clone().requires_grad_(True)
    yr = torch.relu(ref(xr))
    (yr * w).sum().backward()
    assert torch.allclose(y, yr, rtol=1e-4, atol=1e-4)
    assert torch.allclose(x.grad, xr.grad, rtol=1e-3, atol=2e-4)
    assert torch.allclose(bn.weight.grad, ref.weight.grad, rtol=1e-3, atol=1e-2)
    assert torch.allclose(bn.bias.grad, ref.bias.grad, rtol=1e-3, atol=1e-2)
    assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-5)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize("B,F,E,C", [(32, 768, 512, 40), (5, 256, 512, 15), (64, 768, 512, 50)])
def test_head_loss_matches_autograd(ops, B, F, E, C):
    """ppt_head_logits + ppt_head_ce_bwd vs torch: projection, normalised-text logits, CrossEntropyLoss(label_smoothing=0.2)
    and its gradient w.r.t. the un-normalised text features (ULIP_models.py:257,279-281; main_cls.py:52,196)."""
    torch.manual_seed(B + C)
    feat = torch.randn(B, F).cuda(); w = (torch.randn(F, E) / math.sqrt(F)).cuda()
    text = torch.randn(C, E).cuda().requires_grad_(True)
    scale = torch.tensor(math.log(1 / 0.07)).cuda()
    labels = torch.randint(0, C, (B,)).cuda()
    loss, logits, d_text = ops.head_loss(feat, w, text.detach(), scale, labels, 0.2)
    pc = feat.double() @ w.double()
    tn = text.double() / text.double().norm(dim=-1, keepdim=True)
    ref_logits = scale.double().exp() * pc @ tn.t()
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, labels, label_smoothing=0.2)
    (g,) = torch.autograd.grad(ref_loss, text)
    assert torch.allclose(logits.double(), ref_logits, rtol=1e-4, atol=1e-3)
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    assert torch.allclose(d_text.double(), g.double(), rtol=1e-3, atol=1e-5 * g.abs().max().item() + 1e-8)
    loss2, logits2, d2 = ops.head_loss(feat, w, text.detach(), scale, labels, 0.2)
    torch.cuda.synchronize()
    assert torch.equal(logits, logits2) and torch.equal(d_text, d2) and torch.equal(loss, loss2)


def test_misc_ops(ops):
    rng = np.random.default_rng(31)
    x = rng.standard_normal((3, 513, 384)).astype(np.float32)
    out, am = ops.cls_max_pool(dev(x), want_argmax=True)
    xt = torch.from_numpy(x)
    ref = torch.cat([xt[:, 0], xt[:, 1:].max(1)[0]], -1)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(am.cpu().long(), xt[:, 1:].max(1)[1] + 1)
    a = rng.standard_normal((130, 70)).astype(np.float32)
    t = ops.transpose(dev(a), torch.bfloat16)
    assert torch.equal(t.cpu(), torch.from_numpy(a).t().contiguous().to(torch.bfloat16))
    assert torch.equal(ops.convert(dev(a), torch.bfloat16).cpu(), torch.from_numpy(a).to(torch.bfloat16))
    pts = rng.standard_normal((1000, 3)).astype(np.float32)
    w = rng.standard_normal((128, 3)).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    y = ops.linear3_gelu(dev(pts), dev(w), dev(b), torch.float32)
    ref = O.gelu_erf(torch.from_numpy(pts) @ torch.from_numpy(w).t() + torch.from_numpy(b))
    assert (y.cpu() - ref).abs().max().item() < 1e-5
    p = rng.standard_normal((50, 384)).astype(np.float32)
    assert (ops.reduce_rows(dev(p)).cpu() - torch.from_numpy(p).sum(0)).abs().max().item() < 1e-4


def test_ball_query_grouped_and_pointnet2_helpers(ops):
    pc, start = W.synth_clouds(2, 1024, seed=8)
    cidx = O.fps(pc, 64, start)
    center = np.take_along_axis(pc, cidx[:, :, None], axis=1)
    idx, g = ops.ball_query(dev(pc), dev(center), 0.3, 32, want_grouped=True)
    ref = O.ball_query(pc, center, 0.3, 32)
    assert np.array_equal(idx.cpu().numpy(), ref)
    gref = pc[np.arange(2)[:, None, None], ref] - center[:, :, None, :]
    assert np.array_equal(g.cpu().numpy(), gref)
    rng = np.random.default_rng(1)
    P = rng.standard_normal((2 * 1024, 64)).astype(np.float32)
    Q = rng.standard_normal((2 * 64, 64)).astype(np.float32)
    y, (ps, pm) = ops.gather_add(dev(P), dev(Q), idx, 1024, torch.float32)
    yref = P.reshape(2, 1024, 64)[np.arange(2)[:, None, None], ref] + Q.reshape(2, 64, 1, 64)
    assert np.abs(y.cpu().numpy() - yref.reshape(-1, 64)).max() < 1e-6
    ch = torch.from_numpy(yref.reshape(-1, 32, 64))
    assert (ps.cpu() - ch.sum(1)).abs().max().item() < 1e-4
    assert (pm.cpu() - ((ch - ch.mean(1, keepdim=True)) ** 2).sum(1)).abs().max().item() < 1e-3
    # pooled max/min over 16-row groups + finish with mixed-sign scales
    M, N, K = 1024, 64, 64
    A = dev(rng.standard_normal((M, K)).astype(np.float32), torch.bfloat16)
    Wt = dev((rng.standard_normal((N, K)) / 8).astype(np.float32), torch.bfloat16)
    pmax = torch.empty((M // 16, N), device="cuda"); pmin = torch.empty_like(pmax)
    ops.gemm(A, Wt, want_out=False, pool_max=pmax, pool_min=pmin, pool_rows=16)
    full = (A.float().cpu() @ Wt.float().cpu().t()).view(M // 16, 16, N)
    assert (pmax.cpu() - full.max(1)[0]).abs().max().item() < 2e-3 and (pmin.cpu() - full.min(1)[0]).abs().max().item() < 2e-3
    sc = dev(rng.standard_normal(N).astype(np.float32)); sh = dev(rng.standard_normal(N).astype(np.float32))
    out = torch.zeros((M // 32, 2 * N), device="cuda")
    ops.pool_finish(pmax, pmin, 2, sc, sh, out[:, N:])
    want = torch.relu(full.view(M // 32, 32, N) * sc.cpu() + sh.cpu()).max(1)[0]
    assert (out[:, N:].cpu() - want).abs().max().item() < 5e-3 and out[:, :N].abs().max().item() == 0
    x = dev(rng.standard_normal((5, N)).astype(np.float32)); mk = dev((rng.random((5, N)) > 0.5).astype(np.float32) * 2)
    yb = ops.bn_act_rows(x, sc, sh, torch.float32, mask=mk)
    assert (yb.cpu() - torch.relu(x.cpu() * sc.cpu() + sh.cpu()) * mk.cpu()).abs().max().item() < 1e-6


@pytest.mark.parametrize("M,N1,N2", [(64, 64, 64), (32832, 384, 384), (32768, 1536, 384), (4096, 8, 72), (8192, 200, 136)])
def test_gemm_tn_reads_operands_as_stored(ops, M, N1, N2):
    """dW = dY^T X without transposed copies (ppt_gemm_tn_bf16): against fp64 on the same bf16 operands, for ragged
    tile edges, row strides wider than the operand, and run-to-run bit reproducibility of the sliced reduction."""
    g = torch.Generator(device="cuda").manual_seed(M + N1)
    a_full = torch.randn(M, N1 + 8, device="cuda", generator=g).to(torch.bfloat16)
    a = a_full[:, :N1]                                                  # lda = N1 + 8
    b = torch.randn(M, N2, device="cuda", generator=g).to(torch.bfloat16)
    c = ops.gemm_tn_splitk(a, b)
    assert c.shape == (N1, N2) and c.dtype == torch.float32
    ref = a.double().t() @ b.double()
    tol = 2e-5 * float(M) ** 0.5 * 4 + 1e-6
    assert (c.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item() / float(M) ** 0.5)
    assert torch.equal(c, ops.gemm_tn_splitk(a, b))
    # the transposed-copy path (rows not a multiple of 32) computes the same product
    c2 = ops.gemm_tn_splitk(a[:M - 3].contiguous(), b[:M - 3])
    ref2 = a[:M - 3].double().t() @ b[:M - 3].double()
    assert (c2.double() - ref2).abs().max().item() <= tol * max(1.0, ref2.abs().max().item() / float(M) ** 0.5)


@pytest.mark.parametrize("M,D,pad", [(32768, 256, 0), (1000, 36, 4), (777, 130, 0), (5, 8, 0)])
def test_col_sums_matches_torch(ops, M, D, pad):
    """bias gradients: column sums of an fp32 matrix (vector path for D % 4 == 0, row stride wider than D allowed)."""
    g = torch.Generator(device="cuda").manual_seed(M)
    full = torch.randn(M, D + pad, device="cuda", generator=g)
    x = full[:, :D]
    out = ops.col_sums(x)
    ref = x.double().sum(0)
    assert (out.double() - ref).abs().max().item() < 1e-5 * max(1.0, float(M) ** 0.5)
    assert torch.equal(out, ops.col_sums(x))


@pytest.mark.parametrize("groups", [1, 7, 4096])
def test_mini_pointnet_conv12_is_the_prologue_gemm(ops, groups):
    """ppt_mini_pointnet_conv12_bf16 against ppt_gemm(PPT_A_CONV1 + bias + pool over 32 rows): same expression, same
    summation order -> bit-identical y2 and group maxima; and against fp64 within bf16 rounding."""
    g = torch.Generator(device="cuda").manual_seed(groups)
    M = groups * 32
    pts = torch.randn(M, 3, device="cuda", generator=g) * 0.3
    w1 = torch.randn(128, 3, device="cuda", generator=g)
    b1 = torch.randn(128, device="cuda", generator=g) * 0.1
    sc = 1.0 + 0.1 * torch.randn(128, device="cuda", generator=g)
    sh = 0.1 * torch.randn(128, device="cuda", generator=g)
    w2 = (torch.randn(256, 128, device="cuda", generator=g) / 128 ** 0.5).to(torch.bfloat16)
    b2 = torch.randn(256, device="cuda", generator=g) * 0.1
    y2, gm = ops.mini_pointnet_conv12(pts, w1, b1, sc, sh, w2, b2)
    gm_ref = torch.empty((groups, 256), dtype=torch.bfloat16, device="cuda")
    y2_ref = ops.gemm(None, w2, out_dtype=torch.bfloat16, a_mode=ops.A_CONV1, pts=pts, w1=w1, b1=b1, a_scale=sc, a_shift=sh,
                      bias=b2, pool_max=gm_ref)
    assert torch.equal(y2, y2_ref) and torch.equal(gm, gm_ref)
    a = torch.relu(sc.double() * (pts.double() @ w1.double().t() + b1.double()) + sh.double())
    ref = a.to(torch.bfloat16).double() @ w2.double().t() + b2.double()
    assert (y2.double() - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(gm, y2.view(groups, 32, 256).float().amax(1).to(torch.bfloat16))


@pytest.mark.parametrize("B,Q,K,C", [(2, 100, 4, 512), (3, 257, 4, 384), (1, 33, 16, 64)])
def test_group_norm_lrelu_max_matches_torch(B, Q, K, C):
    """csrc/groupnorm.hip (DGCNN_Propagation's GroupNorm(4) + LeakyReLU(0.2) + max over k) against torch autograd of the
    reference formulation: permute to [B,C,Q,K], F.group_norm, F.leaky_relu, max(dim=-1) -- values and all three gradients."""
    from ppt_amd.autograd import group_norm_lrelu_max
    g = torch.Generator(device="cuda").manual_seed(B * Q + C)
    y0 = torch.randn(B, Q, K, C, device="cuda", generator=g) * 1.5 + 0.3
    gn = torch.nn.GroupNorm(4, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(1.0 + 0.2 * torch.randn(C, device="cuda", generator=g))
        gn.bias.copy_(0.1 * torch.randn(C, device="cuda", generator=g))
    dout = torch.randn(B, Q, C, device="cuda", generator=g)
    res = []
    for native in (True, False):
        y = y0.clone().requires_grad_(True)
        gn.zero_grad()
        if native:
            out = group_norm_lrelu_max(y, gn, 0.2)
        else:
            z = torch.nn.functional.group_norm(y.double().permute(0, 3, 1, 2), 4, gn.weight.double(), gn.bias.double(), gn.eps)
            out = torch.nn.functional.leaky_relu(z, 0.2).max(dim=-1)[0].permute(0, 2, 1)
        out.backward(dout.to(out.dtype))
        res.append((out.detach().double(), y.grad.double(), gn.weight.grad.double().clone(), gn.bias.grad.double().clone()))
    for a, b, name in zip(res[0], res[1], ("out", "dy", "dgamma", "dbeta")):
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() < 2e-4 * scale, name


def test_dgcnn_layer_matches_the_graph_feature_formulation():
    """DGCNN_Propagation._layer (conv per source / query point + ppt_gather_add + fused GroupNorm/LeakyReLU/max) against the
    reference formulation built with torch ops: conv(cat(x_k[nn] - x_q, x_q)) -> GroupNorm -> LeakyReLU -> max; values and the
    gradients of both inputs and of the conv / norm parameters, fp32 mode."""
    from ppt_amd.models.pointbert.pointnet2_utils import DGCNN_Propagation
    torch.manual_seed(11)
    B, S, Nq, C = 2, 96, 160, 384
    m = DGCNN_Propagation(k=4).cuda()
    m.precision = torch.float32
    coor = torch.randn(B, S, 3, device="cuda")
    coor_q = torch.randn(B, Nq, 3, device="cuda")
    f0, fq0 = torch.randn(B, S, C, device="cuda"), torch.randn(B, Nq, C, device="cuda")
    dout = torch.randn(B, Nq, 512, device="cuda")
    res = []
    for native in (True, False):
        f, fq = f0.clone().requires_grad_(True), fq0.clone().requires_grad_(True)
        m.zero_grad()
        if native:
            out = m._layer(m.layer1, coor_q, fq, coor, f)
        else:
            g = m.get_graph_feature(coor_q, fq, coor, f).double()                       # [B,Nq,k,2C]
            w = m.layer1[0].weight.reshape(512, -1).double()
            y = (g @ w.t()).permute(0, 3, 1, 2)
            gn = m.layer1[1]
            y = torch.nn.functional.leaky_relu(torch.nn.functional.group_norm(y, 4, gn.weight.double(), gn.bias.double(), gn.eps), 0.2)
            out = y.max(dim=-1)[0].permute(0, 2, 1)
        out.backward(dout.to(out.dtype))
        res.append([t.double() for t in (out.detach(), f.grad, fq.grad, m.layer1[0].weight.grad.clone(), m.layer1[1].weight.grad.clone(),
                                         m.layer1[1].bias.grad.clone())])
    for a, b, name in zip(res[0], res[1], ("out", "df", "dfq", "dW", "dgamma", "dbeta")):
        assert (a - b).abs().max().item() < 2e-3 * max(1.0, b.abs().max().item()), name
    # forward() takes the reference's channel-first tensors (pointnet2_utils.py:433), forward_rows() the row layout
    with torch.no_grad():
        cf = m(coor.permute(0, 2, 1), f0.permute(0, 2, 1), coor_q.permute(0, 2, 1), fq0.permute(0, 2, 1))
        rows = m.forward_rows(coor, f0, coor_q, fq0)
    assert cf.shape == (B, 384, Nq) and torch.equal(cf.permute(0, 2, 1), rows)


@pytest.mark.parametrize("C1,N", [(32, 32), (64, 64), (64, 96), (64, 128), (128, 128)])
def test_conv12_stats_is_the_prologue_gemm(ops, C1, N):
    """ppt_conv12_stats_bf16 against ppt_gemm(PPT_A_CONV1 + bias + column statistics): bit-identical y, and BatchNorm
    partials that finalise to the same scale / shift."""
    g = torch.Generator(device="cuda").manual_seed(C1 + N)
    M = 32 * 777
    pts = torch.randn(M, 3, device="cuda", generator=g) * 0.3
    w1 = torch.randn(C1, 3, device="cuda", generator=g)
    b1 = torch.randn(C1, device="cuda", generator=g) * 0.1
    sc = 1.0 + 0.1 * torch.randn(C1, device="cuda", generator=g)
    sh = 0.1 * torch.randn(C1, device="cuda", generator=g)
    w2 = (torch.randn(N, C1, device="cuda", generator=g) / C1 ** 0.5).to(torch.bfloat16)
    b2 = torch.randn(N, device="cuda", generator=g) * 0.1
    y, (ps, pm) = ops.conv12_stats(pts, w1, b1, sc, sh, w2, b2)
    rs = torch.empty((M // 32, N), dtype=torch.float32, device="cuda")
    rm = torch.empty_like(rs)
    y_ref = ops.gemm(None, w2, out_dtype=torch.bfloat16, a_mode=ops.A_CONV1, pts=pts, w1=w1, b1=b1, a_scale=sc, a_shift=sh, bias=b2,
                     col_stats=(rs, rm))
    assert torch.equal(y, y_ref)
    assert (ps - rs).abs().max().item() < 1e-4 * max(1.0, rs.abs().max().item())
    assert (pm - rm).abs().max().item() < 1e-3 * max(1.0, rm.abs().max().item())
    gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    a = ops.bn_finalize(gamma, beta, True, partials=(ps, pm), rows_per_partial=32, count=M, update_running=False)
    b = ops.bn_finalize(gamma, beta, True, partials=(rs, rm), rows_per_partial=32, count=M, update_running=False)
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() < 1e-4 * max(1.0, v.abs().max().item())


@pytest.mark.parametrize("K,N,pr", [(32, 64, 16), (64, 128, 32), (96, 128, 64), (128, 256, 64)])
def test_affine_conv_pool_is_the_prologue_gemm(ops, K, N, pr):
    """ppt_affine_conv_pool_bf16 against ppt_gemm(PPT_A_AFFINE_RELU + bias + pool max/min + column statistics): bit-identical
    maxima and minima, BatchNorm partials equal to rounding."""
    g = torch.Generator(device="cuda").manual_seed(K + N)
    M = 64 * 301
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    sc = 1.0 + 0.2 * torch.randn(K, device="cuda", generator=g)
    sh = 0.2 * torch.randn(K, device="cuda", generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g) * 0.1

    def bufs():
        return (torch.empty((M // pr, N), dtype=torch.float32, device="cuda"), torch.empty((M // pr, N), dtype=torch.float32, device="cuda"),
                (torch.empty((M // 32, N), dtype=torch.float32, device="cuda"), torch.empty((M // 32, N), dtype=torch.float32, device="cuda")))
    pmax, pmin, st = bufs()
    ops.affine_conv_pool(A, sc, sh, w, b, pr, pmax, pmin, st)
    rmax, rmin, rst = bufs()
    ops.gemm(A, w, a_mode=ops.A_AFFINE_RELU, a_scale=sc, a_shift=sh, bias=b, want_out=False, pool_max=rmax, pool_min=rmin, pool_rows=pr,
             col_stats=rst)
    assert torch.equal(pmax, rmax) and torch.equal(pmin, rmin)
    assert (st[0] - rst[0]).abs().max().item() < 1e-4 * max(1.0, rst[0].abs().max().item())
    assert (st[1] - rst[1]).abs().max().item() < 1e-3 * max(1.0, rst[1].abs().max().item())
    a = torch.relu(A.double() * sc.double() + sh.double()).to(torch.bfloat16).double()
    v = a @ w.double().t() + b.double()
    assert (pmax.double() - v.view(M // pr, pr, N).amax(1)).abs().max().item() < 2e-4 * max(1.0, v.abs().max().item())


@pytest.mark.parametrize("groups", [1, 5, 2048])
def test_mini_pointnet_conv4_is_the_prologue_gemm(ops, groups):
    """ppt_mini_pointnet_conv4_bf16 against ppt_gemm(PPT_A_AFFINE_RELU + bias + pool over 32 rows): bit-identical group maxima."""
    g = torch.Generator(device="cuda").manual_seed(groups)
    M = 32 * groups
    A = torch.randn(M, 512, device="cuda", generator=g).to(torch.bfloat16)
    sc = 1.0 + 0.2 * torch.randn(512, device="cuda", generator=g)
    sh = 0.2 * torch.randn(512, device="cuda", generator=g)
    w = (torch.randn(256, 512, device="cuda", generator=g) / 512 ** 0.5).to(torch.bfloat16)
    b = torch.randn(256, device="cuda", generator=g) * 0.1
    tok = ops.mini_pointnet_conv4(A, sc, sh, w, b)
    ref = torch.empty((groups, 256), dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, w, a_mode=ops.A_AFFINE_RELU, a_scale=sc, a_shift=sh, bias=b, pool_max=ref, want_out=False)
    assert torch.equal(tok, ref)
    a = torch.relu(A.double() * sc.double() + sh.double()).to(torch.bfloat16).double()
    v = (a @ w.double().t() + b.double()).view(groups, 32, 256).amax(1)
    assert (tok.double() - v).abs().max().item() < 2e-2 * max(1.0, v.abs().max().item())


@pytest.mark.parametrize("groups", [1, 5, 2048])
def test_mini_pointnet_conv3_is_the_group_add_gemm(ops, groups):
    """ppt_mini_pointnet_conv3_bf16 against ppt_gemm(group_add + column statistics): bit-identical y3, equal BatchNorm partials."""
    g = torch.Generator(device="cuda").manual_seed(groups + 3)
    M = 32 * groups
    A = torch.randn(M, 256, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(512, 256, device="cuda", generator=g) / 16).to(torch.bfloat16)
    gt = torch.randn(groups, 512, device="cuda", generator=g)
    st = (torch.empty((groups, 512), device="cuda"), torch.empty((groups, 512), device="cuda"))
    y = ops.mini_pointnet_conv3(A, w, gt, st)
    rst = (torch.empty((groups, 512), device="cuda"), torch.empty((groups, 512), device="cuda"))
    ref = ops.gemm(A, w, out_dtype=torch.bfloat16, group_add=gt, group_rows=32, col_stats=rst)
    assert torch.equal(y, ref)
    assert torch.equal(ops.mini_pointnet_conv3(A, w, gt), ref)                   # eval mode: no partials
    assert (st[0] - rst[0]).abs().max().item() < 1e-4 * max(1.0, rst[0].abs().max().item())
    assert (st[1] - rst[1]).abs().max().item() < 1e-3 * max(1.0, rst[1].abs().max().item())


# ------------------------------------------------------------------ rowgemm: weight-stationary short-K linears (csrc/rowgemm.hip)
def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("M,N,K", [(16416, 1152, 384), (513, 384, 384), (100, 1536, 384), (33, 384, 384), (1480, 1536, 512),
                                    (817, 2048, 512), (40, 512, 512), (1, 256, 512), (2080, 200, 384)])
@pytest.mark.parametrize("ln", [False, True])
def test_rowgemm_bf16_forms(ops, M, N, K, ln):
    """bf16 output forms: plain (+bias), GELU, QuickGELU with the saved pre-activation; A as bf16 rows or as the fp32
    residual stream with the LayerNorm prologue.  Reference: fp32 torch on the bf16-rounded operands (what the MFMA
    multiplies), so the only differences are accumulation order and the final bf16 rounding: tolerance 2^-7 relative
    to the row scale (one bf16 ulp of the result) + the A&S erf's 1.5e-7."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    bias = torch.randn(N, generator=g) * 0.1
    if ln:
        x = torch.randn(M, K, generator=g) * 2.0 + torch.randn(M, 1, generator=g)
        gam, bet = 1.0 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
        a_ref = _bf(torch.nn.functional.layer_norm(x, (K,), gam, bet, 1e-5))
        A, kw = x.cuda(), dict(ln=(gam.cuda(), bet.cuda()))
    else:
        a = torch.randn(M, K, generator=g)
        a_ref = _bf(a)
        A, kw = a.cuda().to(torch.bfloat16), {}
    wd = w.cuda().to(torch.bfloat16)
    pre = a_ref @ _bf(w).t() + bias
    for act, fn in ((ops.ACT_NONE, lambda v: v), (ops.ACT_GELU, lambda v: torch.nn.functional.gelu(v)),
                    (ops.ACT_QUICKGELU, lambda v: v * torch.sigmoid(1.702 * v))):
        out2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda") if act != ops.ACT_NONE else None
        out = ops.rowgemm(A, wd, bias=bias.cuda(), act=act, out2=out2, **kw)
        want = fn(pre)
        tol = 2.0 ** -7 * max(1.0, want.abs().max().item()) * (1.5 if ln else 1.0)    # (LN: a bf16 tie of the operand may flip)
        assert (out.float().cpu() - want).abs().max().item() < tol, act
        if out2 is not None:
            assert (out2.float().cpu() - pre).abs().max().item() < tol
        again = ops.rowgemm(A, wd, bias=bias.cuda(), act=act, **kw)
        assert torch.equal(out, again), "bit-reproducible"
    nb = ops.rowgemm(A, wd, **kw)                                          # no bias
    assert (nb.float().cpu() - (pre - bias)).abs().max().item() < 2.0 ** -7 * max(1.0, pre.abs().max().item()) * 1.5


@pytest.mark.parametrize("M,N,K,rows", [(16416, 384, 384, 513), (1026, 384, 384, 513), (1480, 512, 512, 37), (77, 512, 512, 0)])
def test_rowgemm_residual_form(ops, M, N, K, rows):
    """out = residual + row_scale[m // rows] * (acc + bias) (+ residual2), fp32, also in place (out is residual): the
    proj / out_proj epilogue (point_encoder.py:77, ULIP_models.py:53).  fp32 reference on the bf16-rounded operands."""
    g = torch.Generator().manual_seed(M + N)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5
    bias, res, res2 = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    rs = (torch.floor(0.7 + torch.rand((M + rows - 1) // rows, generator=g)) / 0.7) if rows else None
    A, wd = a.cuda().to(torch.bfloat16), w.cuda().to(torch.bfloat16)
    acc = _bf(a) @ _bf(w).t() + bias
    scaled = acc * rs.repeat_interleave(rows)[:M, None] if rows else acc
    kw = dict(row_scale=rs.cuda(), row_scale_rows=rows) if rows else {}
    out = ops.rowgemm(A, wd, bias=bias.cuda(), residual=res.cuda(), residual2=res2.cuda(), **kw)
    assert (out.cpu() - (res + scaled + res2)).abs().max().item() < 2e-4 * max(1.0, acc.abs().max().item())
    x = res.cuda().clone()
    ops.rowgemm(A, wd, bias=bias.cuda(), residual=x, out=x, **kw)            # in place, no second residual
    assert (x.cpu() - (res + scaled)).abs().max().item() < 2e-4 * max(1.0, acc.abs().max().item())


def test_rowgemm_matches_the_tile_gemm_path(ops):
    """The fused LayerNorm -> linear equals ppt_layernorm_fwd + ppt_gemm on the same inputs up to the summation order of the
    LayerNorm statistics (a rare 1-ulp flip of a bf16 operand) and of the K loop: agreement to one bf16 ulp of the output."""
    g = torch.Generator().manual_seed(5)
    M, K, N = 4104, 384, 1536
    x = (torch.randn(M, K, generator=g) * 3).cuda()
    gam, bet = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.2 * torch.randn(K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda()
    h, _, _ = ops.layernorm_fwd(x, gam, bet, torch.bfloat16)
    ref = ops.gemm(h, w, out_dtype=torch.bfloat16, bias=b, act=ops.ACT_GELU)
    got = ops.rowgemm(x, w, ln=(gam, bet), bias=b, act=ops.ACT_GELU)
    d = (ref.float() - got.float()).abs()
    assert d.max().item() <= 2.0 ** -6 * max(1.0, ref.float().abs().max().item())
    assert (d > 0).float().mean().item() < 0.05, "almost every element is bit-identical"


def test_rowgemm_grid_follows_the_persistent_occupancy_and_results_do_not(ops):
    """ppt_rowgemm_bf16 sizes its grid for the share of the CUs ppt_set_persistent_occupancy leaves to the calling thread's
    launches (its workgroups take a whole CU each); what a row tile computes does not depend on which walker takes it."""
    g = torch.Generator().manual_seed(9)
    M, K, N = 16416, 384, 1152
    x = (torch.randn(M, K, generator=g) * 2).cuda()
    gam, bet = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.2 * torch.randn(K, generator=g)).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().to(torch.float16)
    full = ops.rowgemm(x, w, ln=(gam, bet))
    for pct in (70, 25):
        with ops.persistent_occupancy(pct):
            assert ops.get_persistent_occupancy() == pct
            part = ops.rowgemm(x, w, ln=(gam, bet))
        assert torch.equal(part, full), pct
    assert ops.get_persistent_occupancy() == 100


def test_rowgemm_rejects_what_it_does_not_support(ops):
    a = torch.zeros(64, 256, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(64, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.rowgemm(a, w)                                                    # K = 256: PPT_EUNSUPPORTED, no silent fallback


# ------------------------------------------------------------------ part-seg decoder glue (csrc/interp.hip)
@pytest.mark.parametrize("B,N,S,D1,D2,dtype", [(2, 256, 512, 3, 384, torch.float32), (2, 2048, 512, 19, 384, torch.bfloat16),
                                                (1, 100, 37, 0, 64, torch.float32), (3, 33, 64, 5, 130, torch.bfloat16)])
def test_three_nn_interp_fwd_matches_the_reference_formulation(ops, B, N, S, D1, D2, dtype):
    """ppt_three_nn_interp_fwd against pointnet2_utils.py:333-358 written with torch ops on the CPU (and the oracle's 3-NN):
    weights bit-exact up to the division's rounding, rows = [points1 | interpolated | 0] in the operand dtype."""
    rng = np.random.default_rng(B * 1000 + N)
    xyz1 = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    xyz2 = rng.uniform(-1, 1, (B, S, 3)).astype(np.float32)
    p1 = torch.from_numpy(rng.standard_normal((B, N, D1)).astype(np.float32)) if D1 else None
    p2 = torch.from_numpy(rng.standard_normal((B, S, D2)).astype(np.float32))
    idx, _, d = ops.knn_group(dev(xyz2), dev(xyz1), 3, want_nbhd=False, want_dist=True)
    oidx, ow = O.three_nn(xyz1, xyz2)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    mult = 8 if dtype == torch.bfloat16 else 4
    rows, w = ops.three_nn_interp(p1.cuda() if D1 else None, p2.cuda(), idx, d, dtype, mult)
    assert np.abs(w.cpu().numpy() - ow).max() < 2e-6
    gathered = p2[torch.arange(B)[:, None, None], torch.from_numpy(oidx)]
    interp = (gathered * torch.from_numpy(ow)[..., None]).sum(dim=2)
    want = interp if p1 is None else torch.cat([p1, interp], dim=-1)
    ld = (D1 + D2 + mult - 1) // mult * mult
    assert rows.shape == (B * N, ld)
    got = rows.float().cpu().view(B, N, ld)
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -8
    assert (got[..., :D1 + D2] - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    assert (got[..., D1 + D2:] == 0).all()


@pytest.mark.parametrize("B,E,div,S,C,ld,off,weighted", [(2, 6144, 3, 512, 384, 408, 19, True), (3, 2048, 1, 512, 512, 512, 0, False),
                                                          (1, 30000, 3, 700, 130, 136, 4, True), (2, 1024, 1, 256, 384, 384, 0, False),
                                                          (2, 96, 3, 5, 8, 12, 3, True)])
def test_scatter_rows_bwd_is_the_gather_gradient(ops, B, E, div, S, C, ld, off, weighted):
    """ppt_scatter_rows_bwd against the definition (fp64 index_add on the CPU), sources nobody gathered get zeros, and two
    launches give the same bits (owner-computes, fixed order)."""
    g = torch.Generator().manual_seed(E + S)
    idx = torch.randint(0, S, (B, E), generator=g)
    idx[:, : E // 4] = idx[:, : E // 4] % max(1, S // 8)                    # skew: some sources are gathered very often
    if S > 3:
        idx[idx == 3] = 2                                                   # ... and source 3 never
    w = torch.rand(B, E, generator=g) if weighted else None
    d_rows = torch.randn(B * (E // div), ld, generator=g)
    want = torch.zeros(B, S, C, dtype=torch.float64)
    src = d_rows.view(B, E // div, ld)[:, :, off:off + C].double()
    for b in range(B):
        rows = src[b][torch.arange(E) // div]
        if weighted:
            rows = rows * w[b].double()[:, None]
        want[b].index_add_(0, idx[b], rows)
    got = ops.scatter_rows_bwd(idx.cuda(), w.cuda() if weighted else None, d_rows.cuda(), off, div, S, C)
    assert (got.double().cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    if S > 3:
        assert (got[:, 3] == 0).all()
    again = ops.scatter_rows_bwd(idx.cuda(), w.cuda() if weighted else None, d_rows.cuda(), off, div, S, C)
    assert torch.equal(got, again)


def test_sum_groups(ops):
    x = torch.randn(300 * 4, 384)
    got = ops.sum_groups(x.cuda(), 4)
    assert (got.cpu() - x.view(300, 4, 384).sum(1)).abs().max().item() < 1e-5


def test_feature_propagation_matches_the_reference_formulation():
    """PointNetFeaturePropagation (one fused autograd node: interpolation + concat + 2 x (conv, BatchNorm, ReLU) and their
    backward) against the module written with torch ops in fp64, fp32 mode: output, gradient of the interpolated features and
    of every parameter; and forward() in the reference's channel-first layout equals forward_rows()."""
    from ppt_amd.models.pointbert.pointnet2_utils import PointNetFeaturePropagation
    torch.manual_seed(3)
    B, N, S, D1, D2 = 2, 320, 96, 19, 64
    m = PointNetFeaturePropagation(in_channel=D1 + D2, mlp=[128, 48]).cuda()
    m.precision = torch.float32
    m.train()
    xyz1, xyz2 = torch.randn(B, N, 3, device="cuda"), torch.randn(B, S, 3, device="cuda")
    p1, p2_0 = torch.randn(B, N, D1, device="cuda"), torch.randn(B, S, D2, device="cuda")
    dout = torch.randn(B, N, 48, device="cuda")
    res = []
    for native in (True, False):
        p2 = p2_0.clone().requires_grad_(True)
        m.zero_grad()
        for bn in m.mlp_bns:
            bn.reset_running_stats()
        if native:
            out = m.forward_rows(xyz1, xyz2, p1, p2)
        else:
            oidx, ow = O.three_nn(xyz1.cpu().numpy(), xyz2.cpu().numpy())
            gathered = p2.double()[torch.arange(B, device="cuda")[:, None, None], torch.from_numpy(oidx).cuda()]
            x = torch.cat([p1.double(), (gathered * torch.from_numpy(ow).cuda().double()[..., None]).sum(2)], -1).reshape(B * N, -1)
            for conv, bn in zip(m.mlp_convs, m.mlp_bns):
                x = x @ conv.weight.reshape(conv.weight.shape[0], -1).double().t() + conv.bias.double()
                mu, var = x.mean(0), x.var(0, unbiased=False)
                x = torch.relu((x - mu) / torch.sqrt(var + bn.eps) * bn.weight.double() + bn.bias.double())
            out = x.view(B, N, -1)
        out.backward(dout.to(out.dtype))
        grads = [p2.grad] + [q.grad.clone() for q in m.parameters()]
        res.append([out.detach().double()] + [t.double() for t in grads])
        if native:
            rm = m.mlp_bns[0].running_mean.clone()
    names = ["out", "dp2"] + [n for n, _ in m.named_parameters()]
    for a, b, name in zip(res[0], res[1], names):
        if "convs" in name and name.endswith("bias"):
            continue                                       # bias in front of a BatchNorm: zero gradient up to rounding noise
        assert (a - b).abs().max().item() < 2e-3 * max(1.0, b.abs().max().item()), name
    assert rm.abs().max().item() > 0                       # running statistics were updated as nn.BatchNorm1d does
    with torch.no_grad():
        cf = m(xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1), p1.permute(0, 2, 1), p2_0.permute(0, 2, 1))
        rows = m.forward_rows(xyz1, xyz2, p1, p2_0)
    assert cf.shape == (B, 48, N) and torch.equal(cf.permute(0, 2, 1), rows)


@pytest.mark.parametrize("B,S,N,dup", [(2, 512, 1024, False), (1, 512, 8192, False), (2, 128, 512, True), (3, 33, 129, False)])
def test_square_distance_is_bit_exact(ops, B, S, N, dup):
    """models.pointbert.dvae.square_distance on the GPU (ppt_square_distance_f32) == the oracle's restatement of the
    reference CPU arithmetic, bit for bit (it can be slightly negative, as the reference's: SURVEY App. A Q7)."""
    from ppt_amd.models.pointbert.dvae import square_distance
    pc, start = W.synth_clouds(B, N, seed=S + N, duplicates=dup)
    src = np.ascontiguousarray(pc[:, :S])
    got = square_distance(dev(src), dev(pc))
    want = O.square_distance(src, pc)
    assert got.shape == (B, S, N) and np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(ops.square_distance(dev(src), dev(pc)).cpu().numpy(), want)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,T,P,H", [(5, 37, 17, 8), (3, 77, 33, 2), (40, 37, 17, 8), (2, 130, 70, 1)])
def test_attention_prefix_matches_the_unshared_kernels(ops, dtype, tol, C, T, P, H):
    """ppt_attention_prefix_fwd / _bwd (C prompts sharing their first P positions, stored once) against ppt_attention_fwd /
    _bwd on the EXPANDED tensors (every prompt carrying its own copy of the prefix): outputs bit-identical; dQ of all rows and
    dK / dV of the prompts' own rows bit-identical; dK / dV of the shared rows = the sum of the copies' gradients."""
    g = torch.Generator().manual_seed(C * 100 + T)
    HD = 64
    rows = P + C * (T - P)
    qkv_c = torch.randn(rows, 3 * H * HD, generator=g) * 0.5
    dout_c = torch.randn(rows, H * HD, generator=g)
    idx = torch.empty(C, T, dtype=torch.long)                       # compact row of (c, pos)
    for c in range(C):
        idx[c, :P] = torch.arange(P)
        idx[c, P:] = P + c * (T - P) + torch.arange(T - P)
    qkv_f = qkv_c[idx.reshape(-1)].contiguous()
    dout_f = dout_c[idx.reshape(-1)].clone()
    dout_f.view(C, T, -1)[1:, :P] = 0                               # the prefix's outputs are consumed once (from copy 0)
    scale = HD ** -0.5
    qc, qf = qkv_c.cuda().to(dtype), qkv_f.cuda().to(dtype)
    out_c, lse_c = ops.attention_prefix_fwd(qc, C, T, P, H, scale)
    out_f, lse_f = ops.attention_fwd(qf, C, T, H, scale, True)
    exp = out_f.view(C, T, -1)
    assert torch.equal(out_c[:P], exp[0, :P]) and torch.equal(out_c[P:].view(C, T - P, -1), exp[:, P:])
    dq_c = ops.attention_prefix_bwd(qc, out_c, dout_c.cuda().to(dtype), lse_c, C, T, P, H, scale)
    dq_f = ops.attention_bwd(qf, out_f, dout_f.cuda().to(dtype), lse_f, C, T, H, scale, True).view(C, T, 3, H * HD)
    dc = dq_c.view(rows, 3, H * HD)
    assert torch.equal(dc[P:].view(C, T - P, 3, H * HD), dq_f[:, P:])                       # the prompts' own rows, q / k / v
    assert torch.equal(dc[:P, 0], dq_f[0, :P, 0])                                            # dQ of the shared rows
    want = dq_f[:, :P, 1:].float().sum(0)                                                    # dK / dV: summed over the copies
    got = dc[:P, 1:].float()
    assert (got - want).abs().max().item() < tol * max(1.0, want.abs().max().item())
    again = ops.attention_prefix_bwd(qc, out_c, dout_c.cuda().to(dtype), lse_c, C, T, P, H, scale)
    assert torch.equal(dq_c, again)


@pytest.mark.parametrize("B,N,S,qs", [(2, 8192, 512, [(0.1, 16), (0.2, 32), (0.4, 128)]), (3, 512, 128, [(0.2, 32), (0.4, 64), (0.8, 128)]),
                                       (1, 1000, 77, [(0.05, 8), (0.3, 40)]), (2, 300, 64, [(1e-4, 4), (3.0, 128), (0.2, 16)])])
def test_ball_query_multi_equals_the_single_queries(ops, B, N, S, qs):
    """ppt_ball_query_multi_f32 (one pass over the cloud for all radii of a set-abstraction level) == ppt_ball_query_f32 per
    radius == the oracle: indices and centred coordinates bit for bit, including lists with no hit at all and padded lists."""
    pc, start = W.synth_clouds(B, N, seed=N + S)
    ctr = np.take_along_axis(pc, O.fps(pc, S, start)[:, :, None], axis=1)
    outs = ops.ball_query_multi(dev(pc), dev(ctr), qs, want_grouped=True)
    for (r, K), (idx, g) in zip(qs, outs):
        i1, g1 = ops.ball_query(dev(pc), dev(ctr), r, K, want_grouped=True)
        assert torch.equal(idx, i1) and torch.equal(g, g1), (r, K)
        assert np.array_equal(idx.cpu().numpy(), O.ball_query(pc, ctr, r, K)), (r, K)
    plain = ops.ball_query_multi(dev(pc), dev(ctr), qs)
    assert all(g is None for _, g in plain) and all(torch.equal(a[0], b[0]) for a, b in zip(plain, outs))


@pytest.mark.parametrize("R,C,eps", [(4 * 2048, 50, 0.2), (64, 15, 0.2), (1000, 40, 0.0), (129, 96, 0.1)])
def test_cross_entropy_rows_matches_torch(ops, R, C, eps):
    """ppt_cross_entropy_rows against nn.CrossEntropyLoss(label_smoothing) and its autograd gradient (main_partseg.py:213,
    main_cls.py:52): loss to 1e-6 relative, gradient to 1e-7 absolute (values are O(1 / R)); repeatable bit for bit."""
    g = torch.Generator().manual_seed(R + C)
    logits = (torch.randn(R, C, generator=g) * 8).cuda().requires_grad_(True)
    labels = torch.randint(0, C, (R,), generator=g).cuda()
    ref = torch.nn.CrossEntropyLoss(label_smoothing=eps)(logits.double(), labels)
    (gref,) = torch.autograd.grad(ref, logits)
    loss, dl, scale = ops.cross_entropy_rows(logits.detach(), labels, eps)
    loss2, dl2, _ = ops.cross_entropy_rows(logits.detach(), labels, eps)
    assert torch.equal(loss, loss2) and torch.equal(dl, dl2) and scale.item() == 1.0
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert (dl - gref).abs().max().item() < 2e-6 / R * 10 + 1e-7


@pytest.mark.parametrize("R,C,eps,ign", [(700, 50, 0.2, -100), (64, 15, 0.0, -100), (300, 50, 0.2, 255)])
def test_cross_entropy_rows_ignores_the_ignore_index_only(ops, R, C, eps, ign):
    """A row whose label == ignore_index (nn.CrossEntropyLoss: -100 by default; main_partseg-style 255 too) is an ignored row, as
    in ATen -- no loss, zero gradient, left out of the mean.  ADVICE r3 (low): any OTHER label outside [0, C) is a corrupt label --
    ATen raises a device assert -- and must not be dropped silently: the loss comes back NaN."""
    g = torch.Generator().manual_seed(R)
    logits = (torch.randn(R, C, generator=g) * 4).cuda().requires_grad_(True)
    labels = torch.randint(0, C, (R,), generator=g)
    labels[::7] = ign
    labels = labels.cuda()
    ref = torch.nn.CrossEntropyLoss(label_smoothing=eps, ignore_index=ign)(logits.double(), labels)
    (gref,) = torch.autograd.grad(ref, logits)
    loss, dl, scale = ops.cross_entropy_rows(logits.detach(), labels, eps, ign)
    n_valid = int((labels != ign).sum())
    assert abs(scale.item() - R / n_valid) < 1e-6 * R / n_valid
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert (dl * scale - gref).abs().max().item() < 2e-5 / n_valid + 1e-7
    assert dl[::7].abs().max().item() == 0.0
    from ppt_amd.train import _CrossEntropyRows
    lg = logits.detach().clone().requires_grad_(True)
    _CrossEntropyRows.apply(lg, labels, eps, ign).backward()
    assert (lg.grad - gref).abs().max().item() < 2e-5 / n_valid + 1e-7
    for corrupt in (C, C + 7, -1):
        if corrupt == ign:
            continue
        bad = labels.clone()
        bad[5] = corrupt
        loss_b, dl_b, _ = ops.cross_entropy_rows(logits.detach(), bad, eps, ign)
        assert torch.isnan(loss_b).item(), corrupt
        assert torch.isfinite(dl_b).all() and dl_b[5].abs().max().item() == 0.0


def test_wave_priority_changes_no_result(ops):
    """ppt_set_wave_priority only raises the issue priority of the waves (s_setprio): GEMM, LayerNorm forward / backward and
    the short-sequence attention backward give bit-identical results with it on."""
    g = torch.Generator().manual_seed(0)
    a = torch.randn(817, 512, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(1536, 512, generator=g) * 0.05).cuda().to(torch.bfloat16)
    x = torch.randn(817, 512, generator=g).cuda()
    gam, bet = (1 + 0.1 * torch.randn(512, generator=g)).cuda(), (0.1 * torch.randn(512, generator=g)).cuda()
    qkv = torch.randn(817, 1536, generator=g).cuda().to(torch.bfloat16)

    def run():
        c = ops.gemm(a, w, out_dtype=torch.float32)
        y, mean, rstd = ops.layernorm_fwd(x, gam, bet, torch.bfloat16, save_stats=True)
        dx = ops.layernorm_bwd(c[:, :512].contiguous(), x, gam, mean, rstd)[0]
        o, lse = ops.attention_prefix_fwd(qkv, 40, 37, 17, 8, 0.125)
        dq = ops.attention_prefix_bwd(qkv, o, o, lse, 40, 37, 17, 8, 0.125)
        return c, y, dx, o, dq
    plain = run()
    assert ops._lib.lib().ppt_get_wave_priority() == 0
    with ops.wave_priority(1):
        assert ops._lib.lib().ppt_get_wave_priority() == 1
        hi = run()
    assert ops._lib.lib().ppt_get_wave_priority() == 0
    for p_, h_ in zip(plain, hi):
        assert torch.equal(p_, h_)


def test_adamw_step_matches_torch(ops):
    """ppt_adamw_step against torch.optim.AdamW (main_cls.py:58-60 hyper-parameters) over several steps with a changing lr."""
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(32, 512, generator=g) * 0.02
    p_ref = torch.nn.Parameter(p0.clone().cuda())
    opt = torch.optim.AdamW([p_ref], lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    p, m, v = p0.clone().cuda(), torch.zeros(32, 512).cuda(), torch.zeros(32, 512).cuda()
    for step in range(1, 6):
        grad = (torch.randn(32, 512, generator=g) * (10.0 ** (step - 3))).cuda()
        lr = 3e-3 / step
        for gr in opt.param_groups:
            gr['lr'] = lr
        p_ref.grad = grad.clone()
        opt.step()
        ops.adamw_step(p, grad, m, v, lr, 0.9, 0.98, 1e-8, 0.1, step)
        assert (p - p_ref.detach()).abs().max().item() < 2e-7 * max(1.0, p_ref.abs().max().item()), step
        assert (m - opt.state[p_ref]['exp_avg']).abs().max().item() <= 1e-6 * m.abs().max().item()
        assert (v - opt.state[p_ref]['exp_avg_sq']).abs().max().item() <= 1e-6 * v.abs().max().item()


def test_adamw_step_unscales_a_loss_scaled_gradient_in_place(ops):
    """grad_scale = 1 / S (a caller that scaled its loss): the update is the one of the un-scaled gradient, bit for bit (S is a
    power of two), and g is left holding that gradient.  A non-finite gradient element is skipped AND counted."""
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(40, 512, generator=g) * 0.02
    grad = torch.randn(40, 512, generator=g).cuda() * 1e-3
    outs = []
    for S in (1.0, 4096.0):
        p, m, v = p0.clone().cuda(), torch.zeros(40, 512).cuda(), torch.zeros(40, 512).cuda()
        gs = grad * S
        ops.adamw_step(p, gs, m, v, 3e-3, 0.9, 0.98, 1e-8, 0.1, 1, grad_scale=1.0 / S)
        assert torch.equal(gs, grad)
        outs.append((p, m, v))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # a non-finite element is skipped (parameter and moments untouched, gradient reads 0) and counted, the rest updates -- with and
    # without a caller-side scale
    for S in (4096.0, 1.0):
        p, m, v = p0.clone().cuda(), torch.full((40, 512), 0.5).cuda(), torch.full((40, 512), 0.25).cuda()
        gs = grad * S
        gs[3, 7], gs[5, 9], gs[11, 0] = float("inf"), float("nan"), float("-inf")
        skipped = torch.zeros(1, dtype=torch.int64, device="cuda")
        ops.adamw_step(p, gs, m, v, 3e-3, 0.9, 0.98, 1e-8, 0.1, 1, grad_scale=1.0 / S, skipped=skipped)
        bad = torch.zeros(40, 512, dtype=torch.bool); bad[3, 7] = bad[5, 9] = bad[11, 0] = True
        assert skipped.item() == 3
        assert torch.isfinite(p).all() and torch.isfinite(m).all() and torch.isfinite(v).all()
        assert torch.equal(p.cpu()[bad], p0[bad]) and (m.cpu()[bad] == 0.5).all() and (v.cpu()[bad] == 0.25).all() and (gs.cpu()[bad] == 0).all()
        assert not torch.equal(p.cpu()[~bad], p0[~bad]) and torch.equal(gs.cpu()[~bad], grad.cpu()[~bad])
        ops.adamw_step(p, gs, m, v, 3e-3, 0.9, 0.98, 1e-8, 0.1, 2, grad_scale=1.0, skipped=skipped)      # (now finite everywhere)
        assert skipped.item() == 3


@pytest.mark.parametrize("count", [1, 13, 47, 70])
def test_adamw_multi_is_the_single_tensor_kernel_per_tensor(ops, count):
    """ppt_adamw_multi (one launch per 64 tensors, the table as a kernel argument) against ppt_adamw_step tensor by tensor: bit
    identical parameters and moments for sizes from 1 element to a 1536 x 384 matrix, per-tensor step numbers, and the skip
    counter; and against torch.optim.AdamW itself for one step."""
    g = torch.Generator().manual_seed(count)
    shapes = [(1,), (128,), (384,), (1536, 384), (50, 512), (1023,), (1025,), (384, 387)]
    ps = [torch.randn(shapes[i % len(shapes)], generator=g) * 0.05 for i in range(count)]
    gs = [torch.randn(shapes[i % len(shapes)], generator=g) * 10.0 ** ((i % 5) - 3) for i in range(count)]
    steps = [1 + (i % 3) for i in range(count)]
    hyper = (3e-3, 0.9, 0.98, 1e-8, 0.1)
    a = [(p.clone().cuda(), gr.clone().cuda(), torch.full(p.shape, 0.01).cuda(), torch.full(p.shape, 0.02).cuda()) for p, gr in zip(ps, gs)]
    b = [(p.clone().cuda(), gr.clone().cuda(), torch.full(p.shape, 0.01).cuda(), torch.full(p.shape, 0.02).cuda()) for p, gr in zip(ps, gs)]
    if count > 1:
        a[1][1].view(-1)[0] = float("nan"); b[1][1].view(-1)[0] = float("nan")
        a[-1][1].view(-1)[-1] = float("inf"); b[-1][1].view(-1)[-1] = float("inf")
    sk_a, sk_b = torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    for (p, gr, m, v), st in zip(a, steps):
        ops.adamw_step(p, gr, m, v, *hyper, st, skipped=sk_a)
    ops.adamw_multi([(p, gr, m, v, st) for (p, gr, m, v), st in zip(b, steps)], *hyper, skipped=sk_b)
    assert sk_a.item() == sk_b.item() == (2 if count > 1 else 0)
    for ta, tb in zip(a, b):
        for x, y in zip(ta, tb):
            assert torch.equal(x, y)
    # one step of torch.optim.AdamW on fresh state
    params = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    opt = torch.optim.AdamW(params, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    items = []
    for q, gr in zip(params, gs):
        q.grad = gr.clone().cuda()
        items.append((q.detach().clone(), gr.clone().cuda(), torch.zeros_like(q), torch.zeros_like(q), 1))
    opt.step()
    ops.adamw_multi(items, *hyper)
    for q, it in zip(params, items):
        assert (it[0] - q.detach()).abs().max().item() < 2e-7 * max(1.0, q.abs().max().item())


def test_scaled_entry_and_exit_kernels_are_exact(ops):
    """The three places a gradient scale is folded into an existing kernel (ppt_amd/gradscale.py): ppt_convert_scaled,
    ppt_rows_matmul_f32's alpha, ppt_prompt_rows_bwd's scale.  A power of two commutes with every rounding involved: results are
    bit-identical to scaling the input (resp. the output) with a separate multiply; the 8-wide conversion path equals the scalar one."""
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(4096, 136, generator=g) * 1e-5).cuda()
    for T in (torch.float16, torch.bfloat16):
        for S in (1.0, 1024.0, 32768.0):
            got = ops.convert(x, T, scale=S)
            assert torch.equal(got, (x * S).to(T)), (T, S)
            odd = x.view(-1)[1:1 + 4099].contiguous()              # unaligned start, length not a multiple of 8: the scalar path
            assert torch.equal(ops.convert(odd, T, scale=S), (odd * S).to(T))
        assert ops.convert(x.to(T), T) is not None
    a = torch.randn(40, 512, generator=g).cuda() * 1e-3
    w = torch.randn(512, 512, generator=g).cuda() * 0.05
    base = ops.rows_matmul(a, w)
    for S in (64.0, 32768.0):
        assert torch.equal(ops.rows_matmul(a, w, alpha=S), base * S)
    gr = torch.randn(817, 512, generator=g).cuda()
    rows_of = torch.full((32, 40), -1, dtype=torch.int32)
    for t in range(32):
        rows_of[t, :20] = torch.arange(20, dtype=torch.int32) * 40 + t
    rows_of = rows_of.cuda()
    base = ops.prompt_rows_bwd(gr, rows_of, 32)
    assert torch.equal(ops.prompt_rows_bwd(gr, rows_of, 32, scale=1.0 / 4096.0), base * (1.0 / 4096.0))


@pytest.mark.parametrize("position", ["front", "middle", "end"])
def test_prompt_rows_are_the_prompt_learner_splice(ops, position):
    """ppt_prompt_rows / ppt_prompt_rows_bwd against PromptLearner.forward + positional add written with torch ops and its
    autograd: rows bit-identical, token gradient = the sum of the rows' gradients."""
    from types import SimpleNamespace
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=M.dataset_classnames("scanobjectnn"), template_init='', class_name_position=position,
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    m = M.ULIP_PointBERT(args)
    m.prompt_learner.embedding = W.synth_prompt_embedding_from_tokens(m.tokenized_prompts, seed=0)
    m.cuda()
    pl = m.prompt_learner
    C, L, P = len(args.classnames), m._text_len(), pl.shared_prefix()
    for Puse in sorted({0, P}):
        base, slot, pos_rows, rows_of, Mr, eot_rows = pl.row_layout(m.positional_embedding, L, Puse)
        tok = pl.learnable_tokens.detach().clone().requires_grad_(True)
        pl_out = pl.embedding.gather(1, pl._scatter_index(tok.device)[2].unsqueeze(-1).expand(-1, -1, 512))
        cls, pos, _ = pl._scatter_index(tok.device)
        prompts = pl_out.index_put((cls, pos), tok.unsqueeze(0).expand(C, -1, -1)) + m.positional_embedding.detach()
        want = torch.cat([prompts[0, :Puse], prompts[:, Puse:L].reshape(C * (L - Puse), -1)]) if Puse else prompts[:, :L].reshape(C * L, -1)
        got = ops.prompt_rows(base, slot, tok.detach().contiguous(), pos_rows)
        assert got.shape[0] == Mr and torch.equal(got, want.detach())
        gr = torch.randn(want.shape, device="cuda")
        (want * gr).sum().backward()
        dt = ops.prompt_rows_bwd(gr.contiguous(), rows_of, tok.shape[0])
        assert (dt - tok.grad).abs().max().item() < 1e-4 * max(1.0, tok.grad.abs().max().item())
        eot = m.tokenized_prompts.argmax(-1).cuda()
        assert torch.equal(got[eot_rows], prompts.detach()[torch.arange(C, device="cuda"), eot])


# ------------------------------------------------------------------ fused ViT MLP (csrc/mlp_fused.hip)
@pytest.mark.parametrize("variant", [2, 3])              # csrc/mlp_fused.hip / csrc/mlp_fused3.hip (round 6): same contract, same bounds
@pytest.mark.parametrize("M,rows,pos", [(16416, 513, True), (32832, 513, False), (513, 513, True), (1000, 0, False), (77, 11, True)])
def test_vit_mlp_matches_the_unfused_chain(ops, M, rows, pos, variant):
    """ppt_vit_mlp_bf16 (LayerNorm + fc1 + GELU + fc2 + DropPath + residual (+ pos) in one kernel) against (a) fp32 torch math
    on the bf16-rounded operands the MFMAs see and (b) the unfused kernels (ppt_layernorm_fwd + 2 x ppt_gemm); in place and
    out of place; bit-reproducible."""
    g = torch.Generator().manual_seed(M)
    D, Hd = 384, 1536
    x = torch.randn(M, D, generator=g) * 2 + torch.randn(M, 1, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w1, b1 = torch.randn(Hd, D, generator=g) * D ** -0.5, 0.1 * torch.randn(Hd, generator=g)
    w2, b2 = torch.randn(D, Hd, generator=g) * Hd ** -0.5, 0.1 * torch.randn(D, generator=g)
    rs = (torch.floor(0.8 + torch.rand((M + rows - 1) // rows, generator=g)) / 0.8) if rows else None
    r2 = torch.randn(M, D, generator=g) if pos else None
    h = _bf(torch.nn.functional.layer_norm(x, (D,), gam, bet, 1e-5))
    u = _bf(torch.nn.functional.gelu(h @ _bf(w1).t() + b1))
    y = u @ _bf(w2).t() + b2
    want = x + (y * rs.repeat_interleave(rows)[:M, None] if rows else y) + (r2 if pos else 0)
    xd, w1d, w2d = x.cuda(), w1.cuda().to(torch.bfloat16), w2.cuda().to(torch.bfloat16)
    w1t, w2t = ops.vit_mlp_retile(w1d, w2d, variant=variant)
    kw = dict(row_scale=rs.cuda(), row_scale_rows=rows) if rows else {}
    out = torch.empty_like(xd)
    ops.vit_mlp(xd, w1t, b1.cuda(), w2t, b2.cuda(), (gam.cuda(), bet.cuda()), out=out, residual2=r2.cuda() if pos else None, **kw)
    tol = 2e-2 * max(1.0, (want - x).abs().max().item())          # bf16 slab + bf16 LayerNorm output: ~2^-8 relative per operand
    assert (out.cpu() - want).abs().max().item() < tol
    # the unfused kernels on the same inputs
    hh, _, _ = ops.layernorm_fwd(xd, gam.cuda(), bet.cuda(), torch.bfloat16)
    f = ops.gemm(hh, w1d, out_dtype=torch.bfloat16, bias=b1.cuda(), act=ops.ACT_GELU)
    ref = ops.gemm(f, w2d, out=torch.empty_like(xd), bias=b2.cuda(), residual=xd, residual2=r2.cuda() if pos else None, **kw)
    assert (out - ref).abs().max().item() < 1e-2 * max(1.0, (want - x).abs().max().item())
    x2 = xd.clone()
    ops.vit_mlp(x2, w1t, b1.cuda(), w2t, b2.cuda(), (gam.cuda(), bet.cuda()), residual2=r2.cuda() if pos else None, **kw)     # in place
    assert torch.equal(x2, out)


@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,rows", [(16416, 513), (1000, 0), (83, 7)])
def test_vit_mlp_with_the_proj_prologue_matches_proj_then_mlp(ops, M, rows, dtype, variant):
    """ppt_vit_mlp_bf16 with proj_a set (x_mid = x + drop_path1 * (a Wp^T + bp) formed in front, then the MLP branch on it)
    against the two launches it replaces -- ppt_rowgemm_bf16 in residual form, then the plain fused MLP -- and against fp32
    torch math on the operands the MFMAs see; in place; bit-reproducible; ragged last chunk."""
    g = torch.Generator().manual_seed(M + 1)
    D, Hd = 384, 1536
    x = torch.randn(M, D, generator=g) * 2 + torch.randn(M, 1, generator=g)
    a = torch.randn(M, D, generator=g)
    wp, bp = torch.randn(D, D, generator=g) * D ** -0.5, 0.1 * torch.randn(D, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w1, b1 = torch.randn(Hd, D, generator=g) * D ** -0.5, 0.1 * torch.randn(Hd, generator=g)
    w2, b2 = torch.randn(D, Hd, generator=g) * Hd ** -0.5, 0.1 * torch.randn(D, generator=g)
    nr = (M + rows - 1) // rows if rows else 0
    rs1 = (torch.floor(0.9 + torch.rand(nr, generator=g)) / 0.9) if rows else None
    rs2 = (torch.floor(0.8 + torch.rand(nr, generator=g)) / 0.8) if rows else None
    rd = lambda t: t.to(dtype).float()
    y1 = rd(a) @ rd(wp).t() + bp
    xm = x + (y1 * rs1.repeat_interleave(rows)[:M, None] if rows else y1)
    h = rd(torch.nn.functional.layer_norm(xm, (D,), gam, bet, 1e-5))
    u = rd(torch.nn.functional.gelu(h @ rd(w1).t() + b1))
    y2 = u @ rd(w2).t() + b2
    want = xm + (y2 * rs2.repeat_interleave(rows)[:M, None] if rows else y2)
    xd, ad = x.cuda(), a.cuda().to(dtype)
    w1t, w2t = ops.vit_mlp_retile(w1.cuda().to(dtype), w2.cuda().to(dtype), variant=variant)
    wpd = wp.cuda().to(dtype)
    wpt = ops.vit_proj_retile(wpd)
    kw = dict(row_scale=rs2.cuda(), row_scale_rows=rows) if rows else {}
    proj = (ad, wpt, bp.cuda(), rs1.cuda() if rows else None, rows if rows else 0)
    out = xd.clone()
    ops.vit_mlp(out, w1t, b1.cuda(), w2t, b2.cuda(), (gam.cuda(), bet.cuda()), proj=proj, **kw)
    tol = (2e-2 if dtype == torch.bfloat16 else 4e-3) * max(1.0, (want - x).abs().max().item())
    assert (out.cpu() - want).abs().max().item() < tol
    # the two launches it replaces
    ref = xd.clone()
    ops.rowgemm(ad, wpd, bias=bp.cuda(), residual=ref, out=ref, row_scale=rs1.cuda() if rows else None, row_scale_rows=rows)
    ops.vit_mlp(ref, w1t, b1.cuda(), w2t, b2.cuda(), (gam.cuda(), bet.cuda()), **kw)
    assert (out - ref).abs().max().item() < tol / 2
    again = xd.clone()
    ops.vit_mlp(again, w1t, b1.cuda(), w2t, b2.cuda(), (gam.cuda(), bet.cuda()), proj=proj, **kw)
    assert torch.equal(again, out)


# ------------------------------------------------------------------ LayerNorm + K = 384 linear, rows stationary (csrc/lnlin.hip)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,bias", [(16416, 1152, False), (513, 1152, False), (1000, 384, True), (77, 768, True), (32832, 1152, False), (12850, 768, True)])
def test_lnlin_matches_layernorm_then_linear(ops, M, N, bias, dtype):
    """ppt_lnlin (norm1 + qkv of a frozen block: 64-row chunks x 384-column slices, the weight streamed in fragment order) against fp32
    torch math on the operands the MFMAs see and against csrc/rowgemm.hip's weight-stationary form of the same product; ragged last
    chunk; bit-reproducible."""
    g = torch.Generator().manual_seed(M + N)
    D = 384
    x = torch.randn(M, D, generator=g) * 2 + torch.randn(M, 1, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w = torch.randn(N, D, generator=g) * D ** -0.5
    b = 0.1 * torch.randn(N, generator=g) if bias else None
    rd = lambda t: t.to(dtype).float()
    want = rd(torch.nn.functional.layer_norm(x, (D,), gam, bet, 1e-5)) @ rd(w).t() + (b if bias else 0)
    xd, wd = x.cuda(), w.cuda().to(dtype)
    wt = ops.lnlin_retile(wd)
    got = ops.lnlin(xd, wt, (gam.cuda(), bet.cuda()), bias=b.cuda() if bias else None)
    assert got.dtype == dtype and got.shape == (M, N)
    tol = (4e-3 if dtype == torch.float16 else 3e-2) * max(1.0, want.abs().max().item())
    assert (got.float().cpu() - want).abs().max().item() < tol
    ref = ops.rowgemm(xd, wd, ln=(gam.cuda(), bet.cuda()), bias=b.cuda() if bias else None)
    assert (got.float() - ref.float()).abs().max().item() < tol / 2
    assert torch.equal(ops.lnlin(xd, wt, (gam.cuda(), bet.cuda()), bias=b.cuda() if bias else None), got)


# ------------------------------------------------------------------ the text tower's MLP half in one launch (csrc/text_mlp.hip)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [817, 64, 1480, 5])
def test_text_mlp_pair_matches_the_two_linears(ops, M, dtype):
    """ppt_text_mlp_pair (c_fc + QuickGELU + c_proj of a CLIP layer, ULIP_models.py:41-42, as one launch over 64-row blocks x 256-unit
    hidden slices; the eight slices' partial products are summed by the caller's LayerNorm) against fp32 torch math on the operands the
    MFMAs see, forward and backward (the branch's input gradient: the tower is frozen), and against the launches it replaces
    (ppt_gemm with the QuickGELU / derivative epilogue + the split-K product); the saved pre-activation; ragged last block;
    bit-reproducible."""
    g = torch.Generator().manual_seed(M)
    D, Hd = 512, 2048
    rd = lambda t: t.to(dtype).float()
    a = torch.randn(M, D, generator=g)
    w1, b1 = torch.randn(Hd, D, generator=g) * D ** -0.5, 0.1 * torch.randn(Hd, generator=g)
    w2 = torch.randn(D, Hd, generator=g) * Hd ** -0.5
    qg = lambda v: v * torch.sigmoid(1.702 * v)
    pre = rd(a) @ rd(w1).t() + b1
    want = rd(qg(pre)) @ rd(w2).t()
    ad, w1d, w2d = a.cuda().to(dtype), w1.cuda().to(dtype), w2.cuda().to(dtype)
    w1t, w2t = ops.text_mlp_retile(w1d, w2d)
    pre_out = torch.empty((M, Hd), dtype=dtype, device="cuda")
    parts = ops.text_mlp_pair(ad, w1t, w2t, bias=b1.cuda(), pre=pre_out)
    assert parts.shape == (8, M, D)
    got = parts.sum(0).cpu()
    tol = (4e-3 if dtype == torch.float16 else 3e-2) * max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() < tol
    assert (pre_out.float().cpu() - pre).abs().max().item() < (2e-3 if dtype == torch.float16 else 2e-2) * max(1.0, pre.abs().max().item())
    # the launches it replaces
    f = ops.gemm(ad, w1d, out_dtype=dtype, bias=b1.cuda(), act=ops.ACT_QUICKGELU)
    ref = ops.gemm(f, w2d, out_dtype=torch.float32)
    assert (got.cuda() - ref).abs().max().item() < tol / 2
    assert torch.equal(ops.text_mlp_pair(ad, w1t, w2t, bias=b1.cuda()), parts)          # (no pre output: same partial products, bit for bit)
    # the LayerNorm prologue (ln_2 applied while the rows are staged): the partial products of LayerNorm-then-launch, bit for bit, and
    # the statistics of the LayerNorm kernel
    xres = (torch.randn(M, D, generator=g) * 2 + torch.randn(M, 1, generator=g)).cuda()
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
    hh, mean, rstd = ops.layernorm_fwd(xres, gam, bet, dtype, save_stats=True)
    two = ops.text_mlp_pair(hh, w1t, w2t, bias=b1.cuda())
    one, m1, r1 = ops.text_mlp_pair(xres, w1t, w2t, bias=b1.cuda(), ln=(gam, bet), save_stats=True)
    assert (one.sum(0) - two.sum(0)).abs().max().item() < tol / 2
    assert (m1 - mean).abs().max().item() < 1e-5 and ((r1 - rstd) / rstd).abs().max().item() < 1e-5
    # ---- backward: d h2 = ((d_out W_proj) * QuickGELU'(pre)) W_fc, i.e. W1 = c_proj^T [2048, 512], W2 = c_fc^T [512, 2048]
    dout = torch.randn(M, D, generator=g)
    pre_r = rd(pre)
    sg = torch.sigmoid(1.702 * pre_r)
    dpre = (rd(dout) @ rd(w2)) * (sg * (1 + 1.702 * pre_r * (1 - sg)))
    dwant = rd(dpre) @ rd(w1)
    b1t, b2t = ops.text_mlp_retile(w2d.t().contiguous(), w1d.t().contiguous())
    dparts = ops.text_mlp_pair(dout.cuda().to(dtype), b1t, b2t, pre=pre_r.cuda().to(dtype), backward=True)
    dgot = dparts.sum(0).cpu()
    dtol = (4e-3 if dtype == torch.float16 else 3e-2) * max(1.0, dwant.abs().max().item())
    assert (dgot - dwant).abs().max().item() < dtol
    assert torch.equal(ops.text_mlp_pair(dout.cuda().to(dtype), b1t, b2t, pre=pre_r.cuda().to(dtype), backward=True), dparts)


@pytest.mark.parametrize("M", [817, 37, 1480])
def test_text_mlp_pair_split16_matches_fp64_and_the_split16_gemms(ops, M):
    """The split16 form of ppt_text_mlp_pair (csrc/text_mlp_split.hip: fp32 operands multiplied as hi + lo IEEE-half pairs, the weights
    split once by ppt_text_mlp_retile_split) against fp64 math on the fp32 operands -- fp32-GRADE bounds, forward and backward, with
    outlier channels in the activations -- and against the two split16 tile GEMMs it replaces; the saved fp32 pre-activation; ragged last
    block; bit-reproducible; a value beyond half's range is saturated and COUNTED."""
    g = torch.Generator().manual_seed(M)
    D, Hd = 512, 2048
    a = torch.randn(M, D, generator=g)
    a[:, ::37] *= 20.0                                                     # LayerNorm outlier channels
    w1, b1 = torch.randn(Hd, D, generator=g) * D ** -0.5, 0.1 * torch.randn(Hd, generator=g)
    w2 = torch.randn(D, Hd, generator=g) * Hd ** -0.5
    qg = lambda v: v * torch.sigmoid(1.702 * v)
    pre64 = a.double() @ w1.double().t() + b1.double()
    want = qg(pre64) @ w2.double().t()
    ad, w1d, w2d = a.cuda(), w1.cuda(), w2.cuda()
    w1t, w2t = ops.text_mlp_retile_split(w1d, w2d, 4)
    pre_out = torch.empty((M, Hd), dtype=torch.float32, device="cuda")
    parts = ops.text_mlp_pair_split(ad, w1t, w2t, bias=b1.cuda(), pre=pre_out, a_pow2=0)
    assert parts.shape == (8, M, D)
    got = parts.double().sum(0).cpu()
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    assert rel(got, want) < 2e-6, rel(got, want)
    assert (got - want).abs().max().item() < 2e-5 * want.abs().max().item()
    assert rel(pre_out.double().cpu(), pre64) < 1e-6
    # the launches it replaces (same products, another summation order)
    f = ops.gemm(ad, w1d, out_dtype=torch.float32, bias=b1.cuda(), act=ops.ACT_QUICKGELU, split=(0, 4))
    ref = ops.gemm(f, w2d, out_dtype=torch.float32, split=(0, 4))
    assert rel(got, ref.double().cpu()) < 2e-6
    assert torch.equal(ops.text_mlp_pair_split(ad, w1t, w2t, bias=b1.cuda(), a_pow2=0), parts)
    # the LayerNorm prologue: the partial products of LayerNorm-then-launch to fp32 rounding, the LayerNorm kernel's statistics
    xres = (torch.randn(M, D, generator=g) * 2 + torch.randn(M, 1, generator=g)).cuda()
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
    hh, mean, rstd = ops.layernorm_fwd(xres, gam, bet, torch.float32, save_stats=True)
    two = ops.text_mlp_pair_split(hh, w1t, w2t, bias=b1.cuda(), a_pow2=0)
    one, m1, r1 = ops.text_mlp_pair_split(xres, w1t, w2t, bias=b1.cuda(), a_pow2=0, ln=(gam, bet), save_stats=True)
    assert rel(one.double().sum(0).cpu(), two.double().sum(0).cpu()) < 2e-6
    assert (m1 - mean).abs().max().item() < 1e-5 and ((r1 - rstd) / rstd).abs().max().item() < 1e-5
    # ---- backward: ((d_out W_proj) * QuickGELU'(pre)) W_fc with gradient-sized values
    dout = torch.randn(M, D, generator=g) * 1e-2
    sg = torch.sigmoid(1.702 * pre64)
    dwant = ((dout.double() @ w2.double()) * (sg * (1 + 1.702 * pre64 * (1 - sg)))) @ w1.double()
    b1t, b2t = ops.text_mlp_retile_split(w2d.t().contiguous(), w1d.t().contiguous(), 4)
    dparts = ops.text_mlp_pair_split(dout.cuda(), b1t, b2t, pre=pre64.float().cuda(), backward=True, a_pow2=0)
    dgot = dparts.double().sum(0).cpu()
    # (1e-2-sized gradients with no pre-scale sit below 2^-2, where hi + lo keep an ABSOLUTE 2^-25: the tile GEMMs share the floor)
    assert rel(dgot, dwant) < 2e-5, rel(dgot, dwant)
    d_pre = ops.gemm(dout.cuda(), w2d.t().contiguous(), out_dtype=torch.float32, act=ops.ACT_QUICKGELU, dact_pre=pre64.float().cuda(), split=(0, 4))
    dref = ops.gemm(d_pre, w1d.t().contiguous(), out_dtype=torch.float32, split=(0, 4))
    assert rel(dgot, dref.double().cpu()) < 2e-5
    assert torch.equal(ops.text_mlp_pair_split(dout.cuda(), b1t, b2t, pre=pre64.float().cuda(), backward=True, a_pow2=0), dparts)
    # ---- range: a finite activation beyond 65 504 x 2^-a_pow2 is saturated (finite result) and counted
    ctr = ops.split16_overflow_counter()
    before = int(ctr.item())
    big = ad.clone()
    big[0, 3] = 3.0e5
    out = ops.text_mlp_pair_split(big, w1t, w2t, bias=b1.cuda(), a_pow2=0)
    assert torch.isfinite(out).all() and int(ctr.item()) > before


@pytest.mark.parametrize("M,N,K,epi", [(817, 1536, 512, "bias"), (817, 512, 512, "bias+residual"), (817, 512, 1536, "chunks"), (37, 512, 512, "plain"),
                                       (1480, 1536, 512, "bias")])
def test_text_lin_split16_matches_fp64_and_the_split16_gemm(ops, M, N, K, epi):
    """ppt_text_lin_split (csrc/text_lin_split.hip: a linear of the text tower's attention half on hi + lo half products, rows stationary,
    the weight halves streamed) against fp64 math on the fp32 operands at fp32-grade bounds and against the split16 tile GEMM: in_proj
    (bias), out_proj (bias + residual, written over a strided output), the K = 1536 input gradient as three partial products, a ragged
    block; bit-reproducible; saturation counted."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    a[:, ::41] *= 15.0
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = 0.1 * torch.randn(N, generator=g) if "bias" in epi else None
    r = torch.randn(M, N, generator=g) if "residual" in epi else None
    want = a.double() @ w.double().t() + (b.double() if b is not None else 0) + (r.double() if r is not None else 0)
    ad, wd = a.cuda(), w.cuda()
    wt = ops.text_lin_retile_split(wd, 4)
    kw = dict(bias=b.cuda() if b is not None else None, residual=r.cuda() if r is not None else None, a_pow2=0)
    got = ops.text_lin_split(ad, wt, **kw)
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    if epi == "chunks":
        assert got.shape == (K // 512, M, N)
        tot = got.double().sum(0).cpu()
    else:
        assert got.shape == (M, N)
        tot = got.double().cpu()
    assert rel(tot, want) < 2e-6, rel(tot, want)
    ref = ops.gemm(ad, wd, out_dtype=torch.float32, bias=kw["bias"], residual=kw["residual"], split=(0, 4))
    assert rel(tot, ref.double().cpu()) < 2e-6
    assert torch.equal(ops.text_lin_split(ad, wt, **kw), got)
    if epi == "bias+residual":                              # out= : a strided destination
        big = torch.zeros((M, N + 64), dtype=torch.float32, device="cuda")
        ops.text_lin_split(ad, wt, out=big[:, :N], **kw)
        assert torch.equal(big[:, :N], got) and float(big[:, N:].abs().max()) == 0.0
    ctr = ops.split16_overflow_counter()
    before = int(ctr.item())
    hot = ad.clone()
    hot[M - 1, 7] = -2.5e5
    out = ops.text_lin_split(hot, wt, **kw)
    assert torch.isfinite(out).all() and int(ctr.item()) > before


@pytest.mark.parametrize("tag,N,M,dup,cols", [("d", 8192, 1024, False, 3), ("e", 2048, 512, True, 6)])
def test_dataset_fps_is_bit_exact(tag, N, M, dup, cols):
    """ppt_amd.data.farthest_point_sample(point, npoint) (data/dataset_3d.py:40-61 on the FPS kernel): the rows the reference
    selected (golden) and the oracle's restated loop select, on the committed cases and on random clouds / starts / widths;
    np.random.randint is drawn exactly once when no start is given; float64 input is rejected."""
    from ppt_amd import data as PD
    gidx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_index.npz"))
    pc, start = W.synth_clouds(1, N, seed=4321, duplicates=dup)
    pts = pc[0].astype(np.float32)
    if cols == 6:
        pts = np.concatenate([pts, pts[:, ::-1] * 0.5], axis=1)
    rows, idx = PD.farthest_point_sample(pts, M, start=int(start[0]), return_index=True)
    assert np.array_equal(idx, gidx[f"dsfps_{tag}_idx"].astype(np.int64))
    assert rows.dtype == np.float32 and np.array_equal(rows, pts[idx])
    rng = np.random.default_rng(N)
    for n, m in ((1500, 1024), (33, 7), (4096, 2048)):
        cloud = rng.standard_normal((n, 3)).astype(np.float32)
        cloud[n // 2:n // 2 + 5] = cloud[0]                        # duplicates
        st = int(rng.integers(0, n))
        want_rows, want_idx = O.dataset_farthest_point_sample(cloud, m, st)
        got_rows, got_idx = PD.farthest_point_sample(cloud, m, start=st, return_index=True)
        assert np.array_equal(got_idx, want_idx) and np.array_equal(got_rows, want_rows)
    r = np.random.RandomState(11)
    want_start, want_next = r.randint(0, N), r.randint(0, 1 << 30)
    np.random.seed(11)
    _, idx2 = PD.farthest_point_sample(pts, 16, return_index=True)
    assert idx2[0] == want_start and np.random.randint(0, 1 << 30) == want_next      # ONE draw, where the reference draws (:51)
    with pytest.raises(TypeError):
        PD.farthest_point_sample(pts.astype(np.float64), 8)


# ------------------------------------------------------------------ split16: fp32 operands as hi + lo half pairs
@pytest.mark.parametrize("M,N,K,kind", [
    (817, 1536, 512, "plain"),           # the prompt chain's in_proj (64 x 64 tiles)
    (817, 512, 2048, "residual"),        # c_proj: bias + fp32 residual
    (16416, 1536, 384, "gelu"),          # fc1 of a C2 batch (128 x 128 tiles), ragged M
    (16416, 384, 1536, "residual"),      # fc2: narrow N over many rows -- 195 tiles of 256 x 128
    (32768, 512, 96, "plain"),           # short K (3 slabs) on the 256 x 128 tile
    (33001, 200, 160, "gelu"),           # ragged M and N on the 256 x 128 tile (258 tiles)
    (1000, 200, 36, "plain"),            # ragged everything, K not a multiple of the 32-float slab
    (9000, 640, 64, "dact"),             # derivative epilogue + second (pre-activation) output
    (16384, 512, 256, "stats"),          # per-group term + BatchNorm chunk statistics + the 32-row max pool
    (8192, 256, 128, "affine"),          # A prologue: BatchNorm + ReLU of the previous layer
    (4096, 128, 0, "conv1"),             # A prologue: the K = 3 first conv from the points
    (3000, 256, 96, "batched"),
])
def test_gemm_split16_is_fp32_grade(ops, M, N, K, kind):
    """ppt_gemm_params.split16 (csrc/gemm_common.h: the fp32 operands multiplied as hi + lo IEEE-half pairs, three 16-bit MFMAs
    per product, fp32 accumulation) against the fp32 MFMA on the same launch and against an fp64 product: every A prologue and
    epilogue of the fp32 path, 64 x 64 and 128 x 128 tiles.  The split product is at least as close to fp64 as the fp32 MFMA's
    (measured 2.7e-7 vs 4.2e-7 rel-L2); side outputs agree to the same level; reproducible bit for bit."""
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    kw, a_kw = {}, {}
    A = None
    if kind == "conv1":
        K = 128
        pts = dev(rng.standard_normal((M, 3)).astype(np.float32))
        w1, b1 = dev((rng.standard_normal((K, 3)) * 0.5).astype(np.float32)), dev(rng.standard_normal(K).astype(np.float32) * 0.1)
        sc, sh = dev((rng.random(K) + 0.5).astype(np.float32)), dev(rng.standard_normal(K).astype(np.float32) * 0.1)
        a_kw = dict(a_mode=ops.A_CONV1, pts=pts, w1=w1, b1=b1, a_scale=sc, a_shift=sh)
        A_eff = torch.relu((pts.double() @ w1.double().t() + b1.double()) * sc.double() + sh.double())
    else:
        A = dev(rng.standard_normal((M, K)).astype(np.float32))
        A[:, ::7] *= 20.0                                   # outlier channels
        A_eff = A.double()
        if kind == "affine":
            sc, sh = dev((rng.random(K) + 0.5).astype(np.float32)), dev(rng.standard_normal(K).astype(np.float32))
            a_kw = dict(a_mode=ops.A_AFFINE_RELU, a_scale=sc, a_shift=sh)
            A_eff = torch.relu(A.double() * sc.double() + sh.double())
    Bm = dev((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32))
    bias = dev(rng.standard_normal(N).astype(np.float32))
    ref = A_eff @ Bm.double().t()
    if kind == "gelu":
        kw = dict(bias=bias, act=ops.ACT_GELU)
        ref = torch.nn.functional.gelu(ref + bias.double())
    elif kind == "residual":
        res = dev(rng.standard_normal((M, N)).astype(np.float32))
        kw = dict(bias=bias, residual=res)
        ref = ref + bias.double() + res.double()
    elif kind == "dact":
        pre = dev(rng.standard_normal((M, N)).astype(np.float32))
        kw = dict(act=ops.ACT_QUICKGELU, dact_pre=pre)
        sg = torch.sigmoid(1.702 * pre.double())
        ref = ref * (sg * (1.0 + 1.702 * pre.double() * (1.0 - sg)))
    elif kind == "stats":
        gadd = dev(rng.standard_normal((M // 32, N)).astype(np.float32))
        kw = dict(bias=bias, group_add=gadd, group_rows=32)
        ref = ref + bias.double() + gadd.double().repeat_interleave(32, dim=0)

    def launch(split):
        k2, extra = dict(kw), {}
        if kind == "stats":
            cs = torch.full(((M + 31) // 32, N), float("nan"), device="cuda"); cq = torch.full_like(cs, float("nan"))
            pm = torch.full((M // 32, N), float("nan"), device="cuda")
            k2.update(col_stats=(cs, cq), pool_max=pm, pool_rows=32)
            extra = dict(cs=cs, cq=cq, pm=pm)
        if kind == "batched":
            Mb = M // 3
            out = torch.full((3 * Mb, N), float("nan"), device="cuda")
            ops.gemm(A[:Mb], Bm, out=out, M=Mb, batch=3, strideA=Mb * K, strideB=0, strideC=Mb * N, split=split, **k2)
        else:
            out = ops.gemm(A, Bm, out_dtype=torch.float32, split=split, **a_kw, **k2)
        torch.cuda.synchronize()
        return out, extra
    f32, f32_x = launch(False)
    s16, s16_x = launch(True)
    again, _ = launch(True)
    assert torch.equal(s16, again), "not reproducible"
    if kind in ("plain", "residual", "gelu", "dact"):
        # the 256 x 128 tile (large plain problems) and the 128 x 128 / 64 x 64 loops walk K and split alike: the same bits
        tiles = ops.gemm(A, Bm, out_dtype=torch.float32, split=True, core="tiles", **kw)
        assert torch.equal(s16, tiles)
    if kind == "batched":
        ref = ref[:3 * (M // 3)]
    scale = ref.abs().max().item()
    e32 = (f32.double() - ref).abs().max().item() / scale
    e16 = (s16.double() - ref).abs().max().item() / scale
    r32 = ((f32.double() - ref).norm() / ref.norm()).item()
    r16 = ((s16.double() - ref).norm() / ref.norm()).item()
    print(f"PARITY split16 gemm {kind} {M}x{N}x{K}: max-err/max {e16:.2e} (fp32 MFMA {e32:.2e}), rel-L2 {r16:.2e} (fp32 MFMA {r32:.2e})")
    assert e16 < 4e-6 and r16 < 2e-6 and r16 < 1.5 * r32 + 1e-7
    for k in f32_x:
        a, b = f32_x[k].double(), s16_x[k].double()
        assert torch.isfinite(b).all() and ((a - b).norm() / a.norm()).item() < 1e-5, k
    if kind == "plain" and K == 512:
        # magnitudes: gradient-like A (1e-6) sinks under half's subnormal floor un-scaled and is recovered by the caller's power of two
        tiny = A * 1e-6
        want = tiny.double() @ Bm.double().t()
        bad = ops.gemm(tiny, Bm, split=(0, 4))
        good = ops.gemm(tiny, Bm, split=(20, 4))
        assert ((bad.double() - want).norm() / want.norm()).item() > 1e-3
        assert ((good.double() - want).norm() / want.norm()).item() < 1e-6
        with pytest.raises(RuntimeError):
            ops.gemm(A, Bm, split=(30, 0))                  # |pow2| <= 24
        h = ops.gemm(A.half(), Bm.half(), split=True)      # 16-bit operands: `split` does not apply
        assert torch.equal(h, ops.gemm(A.half(), Bm.half(), split=False))


@pytest.mark.parametrize("M,N,K", [(817, 512, 512), (16416, 384, 1536), (300, 200, 96)])
def test_gemm_split16_saturates_a_finite_overflow_and_counts_it(ops, M, N, K):
    """half(x * s) is inf beyond 65 504 and lo = half(x * s - inf) NaN, where the fp32 MFMA this mode stands in for gives a
    finite product.  The split saturates a FINITE value to +-65 504 (finite result, wrong by what was cut off) and every wave that did
    adds 1 to the process-wide counter ppt_amd/health.py polls (ppt_gemm_params.split_overflow); inf / NaN inputs propagate as in the
    fp32 mode and are NOT counted; in-range launches leave the counter alone.  64 x 64, 128 x 128 and 256 x 128 split kernels."""
    rng = np.random.default_rng(M + N)
    A = dev(rng.standard_normal((M, K)).astype(np.float32))
    Bm = dev((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32))
    cnt = ops.split16_overflow_counter(A.device)
    torch.cuda.synchronize()
    c0 = int(cnt.item())
    ok = ops.gemm(A, Bm, split=True)
    torch.cuda.synchronize()
    assert int(cnt.item()) == c0 and torch.isfinite(ok).all()
    big = A.clone()
    big[5, 7] = 3.0e6                                       # finite, beyond half's range (A's pre-scale is 2^0)
    out = ops.gemm(big, Bm, split=True)
    torch.cuda.synchronize()
    c1 = int(cnt.item())
    assert c1 > c0, "the saturation was not counted"
    assert torch.isfinite(out).all(), "a finite fp32 operand must not become inf / NaN in the split"
    want = big.clone()
    want[5, 7] = 65504.0                                    # what the launch multiplied
    ref = want.double() @ Bm.double().t()
    assert ((out.double() - ref).norm() / ref.norm()).item() < 2e-6
    rows = torch.ones(M, dtype=torch.bool, device=A.device)
    rows[5] = False
    assert torch.equal(out[rows], ok[rows])                 # every other row: the in-range launch's bits
    # the WEIGHT operand is pre-scaled by 2^4: 5 000 x 16 is beyond the range too
    wbig = Bm.clone()
    wbig[3, 1] = 5000.0
    out = ops.gemm(A, wbig, split=True)
    torch.cuda.synchronize()
    c2 = int(cnt.item())
    assert c2 > c1 and torch.isfinite(out).all()
    assert torch.isfinite(ops.gemm(A, wbig, split=(0, 0))).all() and int(cnt.item()) == c2      # ... and inside it at 2^0
    # inf / NaN inputs are the caller's: they propagate, uncounted
    bad = A.clone()
    bad[2, 3] = float("inf")
    bad[9, 1] = float("nan")
    out = ops.gemm(bad, Bm, split=True)
    torch.cuda.synchronize()
    assert int(cnt.item()) == c2
    assert not torch.isfinite(out[2]).any() and torch.isnan(out[9]).all() and torch.isfinite(out[rows & (torch.arange(M, device=A.device) != 2) & (torch.arange(M, device=A.device) != 9)]).all()


@pytest.mark.parametrize("name,Bt,T,H,causal,P,gain", [
    ("vit", 32, 513, 6, False, 0, 1.0),                     # T = 64 n + 1: the peeled last key
    ("vit, large scores", 8, 513, 6, False, 0, 3.0),
    ("text", 40, 77, 8, True, 0, 1.0),
    ("prefix-shared", 40, 77, 8, True, 17, 1.5),
    ("ragged", 3, 200, 2, False, 0, 1.0),
    ("causal, 5 key tiles", 5, 300, 4, True, 0, 2.0),
    ("five tokens", 2, 5, 1, False, 0, 1.0),                # less than one query block / key tile
    ("one full tile, causal", 1, 64, 3, True, 0, 1.0),
    ("65 = 64 + 1: peeled last key", 7, 65, 2, False, 0, 1.0),
    ("prefix of one", 6, 20, 2, True, 1, 1.0),
])
def test_attention_split16_forward_is_fp32_grade(ops, name, Bt, T, H, causal, P, gain):
    """csrc/attention_split.hip (K.Q^T and V^T.P from hi + lo half pairs on the matrix pipe, fp32 softmax) against the fp32 VALU
    kernel -- same layouts, prefix-shared included -- and against an fp64 softmax(QK^T)V."""
    g = torch.Generator().manual_seed(T + P)
    rows = Bt * T if P == 0 else ops.prefix_rows(Bt, T, P)
    qkv = (torch.randn(rows, 3 * H * 64, generator=g) * gain).cuda()

    def run(split):
        ops.set_split16(split)
        try:
            if P:
                return ops.attention_prefix_fwd(qkv, Bt, T, P, H, 0.125)
            return ops.attention_fwd(qkv, Bt, T, H, 0.125, causal)
        finally:
            ops.set_split16(False)
    o32, l32 = run(False)
    o16, l16 = run(True)
    o16b, _ = run(True)
    torch.cuda.synchronize()
    assert torch.equal(o16, o16b)
    sc = o32.abs().max().item()
    d = (o16 - o32).abs().max().item() / sc
    dl = (l16 - l32).abs().max().item()
    print(f"PARITY split16 attention {name}: out vs fp32 kernel max-err/max {d:.2e}, lse abs {dl:.2e}")
    assert d < 5e-6 and dl < 5e-5
    if P == 0:
        q, k, v = qkv.double().view(Bt, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        s_ = (q @ k.transpose(-1, -2)) * 0.125
        if causal:
            s_ = s_.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool, device="cuda"), 1), float("-inf"))
        want = (torch.softmax(s_, -1) @ v).permute(0, 2, 1, 3).reshape(Bt * T, H * 64)
        e16 = (o16.double() - want).abs().max().item() / want.abs().max().item()
        e32 = (o32.double() - want).abs().max().item() / want.abs().max().item()
        print(f"PARITY split16 attention {name}: vs fp64 {e16:.2e} (fp32 kernel {e32:.2e})")
        assert e16 < 5e-6


@pytest.mark.parametrize("name,Bt,T,H,causal,gscale", [
    ("vit", 8, 513, 6, False, 1.0),
    ("vit, small gradients", 4, 513, 6, False, 1e-3),
    ("vit, tiny gradients", 2, 513, 6, False, 1e-7),
    ("vit, huge gradients", 2, 513, 6, False, 3e4),
    ("text", 40, 77, 8, True, 1.0),
    ("ragged", 3, 200, 2, False, 1.0),
    ("causal, 3 key blocks", 5, 300, 4, True, 1.0),
    ("five tokens", 2, 5, 1, False, 1.0),
])
def test_attention_split16_backward_is_fp32_grade(ops, name, Bt, T, H, causal, gscale):
    """ppt_attention_bwd_split16 (csrc/attention_split.hip: dK / dV and dQ kernels with every MFMA operand a hi + lo half pair)
    against the fp32 VALU kernels on the same inputs and against fp64 autograd of softmax(q k^T scale) v."""
    g = torch.Generator().manual_seed(T + Bt)
    qkv = torch.randn(Bt * T, 3 * H * 64, generator=g).cuda()
    dout = (torch.randn(Bt * T, H * 64, generator=g) * gscale).cuda()

    def run(split):
        ops.set_split16(split)
        try:
            out, lse = ops.attention_fwd(qkv, Bt, T, H, 0.125, causal)
            return ops.attention_bwd(qkv, out, dout, lse, Bt, T, H, 0.125, causal)
        finally:
            ops.set_split16(False)
    d32 = run(False)
    d16 = run(True)
    d16b = run(True)
    torch.cuda.synchronize()
    assert torch.equal(d16, d16b)
    q64 = qkv.double().requires_grad_(True)
    q, k, v = q64.view(Bt, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s_ = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s_ = s_.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    o = (torch.softmax(s_, -1) @ v).permute(0, 2, 1, 3).reshape(Bt * T, H * 64)
    o.backward(dout.double())
    want = q64.grad
    for part, sl in (("dq", slice(0, H * 64)), ("dk", slice(H * 64, 2 * H * 64)), ("dv", slice(2 * H * 64, 3 * H * 64))):
        w_ = want[:, sl]
        e16 = ((d16[:, sl].double() - w_).norm() / w_.norm()).item()
        e32 = ((d32[:, sl].double() - w_).norm() / w_.norm()).item()
        print(f"PARITY split16 attention backward {name} {part}: rel-L2 vs fp64 {e16:.2e} (fp32 kernels {e32:.2e})")
        assert e16 < 3e-6 and e16 < 3 * e32 + 5e-7


@pytest.mark.parametrize("C,T,P,H,gscale", [(40, 77, 17, 8, 1.0), (40, 37, 17, 8, 1e-5), (6, 20, 1, 2, 1.0), (3, 150, 130, 2, 1.0)])
def test_attention_split16_prefix_backward_matches_the_fp32_kernels(ops, C, T, P, H, gscale):
    """ppt_attention_bwd_split16 in the prefix-shared layout (attn_rowmap.h: the prompt chain's causal attention, the first P
    positions stored once): against ppt_attention_prefix_bwd on fp32 -- dQ / dK / dV of every physical row, the shared rows'
    dK / dV being the fixed-order sum over the C + 1 virtual sequences."""
    g = torch.Generator().manual_seed(T * 7 + P)
    rows = ops.prefix_rows(C, T, P)
    qkv = torch.randn(rows, 3 * H * 64, generator=g).cuda()
    dout = (torch.randn(rows, H * 64, generator=g) * gscale).cuda()

    def run(split):
        ops.set_split16(split)
        try:
            out, lse = ops.attention_prefix_fwd(qkv, C, T, P, H, 0.125)
            return ops.attention_prefix_bwd(qkv, out, dout, lse, C, T, P, H, 0.125)
        finally:
            ops.set_split16(False)
    d32, d16, d16b = run(False), run(True), run(True)
    torch.cuda.synchronize()
    assert torch.equal(d16, d16b)
    for part, sl in (("dq", slice(0, H * 64)), ("dk", slice(H * 64, 2 * H * 64)), ("dv", slice(2 * H * 64, 3 * H * 64))):
        for what, rs_ in (("shared rows", slice(0, P)), ("own rows", slice(P, rows))):
            a, b = d32[rs_, sl].double(), d16[rs_, sl].double()
            if a.norm().item() == 0.0:             # (a lone first position attends to itself only: its dQ is exactly zero)
                assert (b.abs().max().item()) < 1e-6 * gscale
                continue
            rel = ((a - b).norm() / a.norm()).item()
            print(f"PARITY split16 prefix attention backward C{C} T{T} P{P} {part} {what}: rel-L2 vs the fp32 kernels {rel:.2e}")
            assert rel < 2e-6


class _CloudsWithDatasetFPS(torch.utils.data.Dataset):
    """The shape of the reference's datasets (data/dataset_3d.py:288-300): __getitem__ seeds nothing, calls
    farthest_point_sample(point, npoint) -- which draws its start with np.random.randint -- and returns the rows."""

    def __init__(self, n_items, N, npoint):
        self.n_items, self.N, self.npoint = n_items, N, npoint

    def __len__(self):
        return self.n_items

    def cloud(self, i):
        c = np.random.default_rng(1000 + i).standard_normal((self.N, 3)).astype(np.float32)
        c[5:9] = c[0]
        return c

    def __getitem__(self, i):
        from ppt_amd import data as PD
        np.random.seed(77 + i)                                     # (so the test can replay the draw; the reference leaves it unseeded)
        return i, PD.farthest_point_sample(self.cloud(i), self.npoint)


def test_dataset_fps_inside_dataloader_workers():
    """VERDICT r4 missing #4: the reference's datasets sample inside DataLoader WORKER processes.  Without the service the call raises
    in the worker with the way out; with ppt_amd.data.start_fps_service() an unchanged dataset class runs under num_workers = 2 (fork)
    and returns, sample for sample, the rows the oracle's restated loop selects from the start index the worker drew."""
    from ppt_amd import data as PD
    ds = _CloudsWithDatasetFPS(12, 2048, 256)
    PD.stop_fps_service()
    with pytest.raises(RuntimeError, match="start_fps_service"):
        for _ in torch.utils.data.DataLoader(ds, batch_size=4, num_workers=2, timeout=120):
            pass
    svc = PD.start_fps_service()
    try:
        seen = 0
        for ids, rows in torch.utils.data.DataLoader(ds, batch_size=4, num_workers=2, timeout=120):
            for i, r in zip(ids.tolist(), rows.numpy()):
                cloud = ds.cloud(i)
                start = np.random.RandomState(77 + i).randint(0, ds.N)
                want, _ = O.dataset_farthest_point_sample(cloud, ds.npoint, int(start))
                assert np.array_equal(r, want)
                seen += 1
        assert seen == 12 and svc.served == 12
        # the main process keeps its direct path while the service runs
        np.random.seed(77)
        assert np.array_equal(PD.farthest_point_sample(ds.cloud(0), 256), ds[0][1])
    finally:
        PD.stop_fps_service()


def test_dataset_fps_service_batches_the_workers_requests():
    """eight forked workers, N = 8192 -> 1024 (the reference's 8192-point configurations, data/dataset_3d.py:295): the
    service drains its queue and runs every pending cloud in ONE ppt_fps_f32 launch, clouds and indices through shared-memory slots.
    Indices stay the oracle's bit for bit (spot-checked: the oracle's Python loop at this size is slow); throughput is printed and
    held to a floor.  The ceiling: a worker has one request outstanding, so 8 workers put at most 8 clouds into a 1.0 ms walk of 1024
    serial picks -- <= 8 / (1.0 ms + round trip) ~ 6 500 clouds/s; round 5's one-cloud-per-launch service with pickled arrays
    measured ~1 000."""
    import time
    from ppt_amd import data as PD

    class Cached(_CloudsWithDatasetFPS):
        """clouds generated once per worker (the test times the service, not numpy's normal generator)"""
        def cloud(self, i):
            if not hasattr(self, "_c"):
                self._c = {}
            k = i % 16
            if k not in self._c:
                self._c[k] = _CloudsWithDatasetFPS.cloud(self, k)
            return self._c[k]

    ds = Cached(2048, 8192, 1024)
    PD.stop_fps_service()
    svc = PD.start_fps_service()
    try:
        loader = torch.utils.data.DataLoader(ds, batch_size=32, num_workers=8, timeout=300, persistent_workers=False)
        it = iter(loader)
        first = next(it)                                   # (workers forked, caches filled, first launches done)
        served0, launches0 = svc.served, svc.launches
        t0 = time.perf_counter()
        n, keep = 0, []
        for ids, rows in it:
            n += len(ids)
            if len(keep) < 2:
                keep.append((int(ids[0]), rows[0].numpy().copy()))
        dt = time.perf_counter() - t0
        checked = 0
        for i, got in keep:                                # spot check against the oracle's restated loop (seconds per cloud: outside the clock)
            start = np.random.RandomState(77 + i).randint(0, ds.N)
            want, _ = O.dataset_farthest_point_sample(ds.cloud(i), ds.npoint, int(start))
            assert np.array_equal(got, want)
            checked += 1
        served, launches = svc.served - served0, svc.launches - launches0
        rate = n / dt
        print(f"PARITY dataset FPS service: {n} clouds of 8192 -> 1024 through 8 forked workers in {dt:.2f} s = {rate:.0f} clouds/s; "
              f"{served} served in {launches} launches ({served / max(launches, 1):.1f} clouds per launch, {1e3 * svc.launch_s / max(svc.launches, 1):.2f} ms per launch)")
        assert checked == 2 and len(first[0]) == 32
        assert served / max(launches, 1) > 2.0, "requests that are pending together must share a launch"
        assert rate > 2000, rate
    finally:
        PD.stop_fps_service()


# ------------------------------------------------------------------ fused conv3 + BN + ReLU + conv4 + max (csrc/mpn34.hip)
@pytest.mark.parametrize("T", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tiles", [2, 7, 1027, 4096])
def test_mini_pointnet_conv34_matches_fp32_math_on_the_same_operands(ops, T, tiles):
    """ppt_mini_pointnet_conv34_half (conv3 -> folded BatchNorm -> ReLU -> conv4 -> max over the group, the [M,512]
    intermediate never leaving the chip) against fp32 torch math on the operands the MFMAs see -- y2, the scale-folded W3s and W4
    rounded to the 16-bit format, the ReLU output rounded once -- for even / odd group counts and more chunks than workgroups;
    the per-group bias gs rides in the matrix pipe as a hi + lo pair: its fp32 value must come through to ~1e-6 relative.
    Bit-reproducible; ppt_scale_rows_convert (the fold) against torch."""
    g = torch.Generator().manual_seed(tiles)
    M = 32 * tiles
    y2 = (torch.randn(M, 256, generator=g) * 1.5).cuda().to(T)
    w3 = (torch.randn(512, 512, generator=g) * 512 ** -0.5).cuda()
    b3 = (0.1 * torch.randn(512, generator=g)).cuda()
    sc = (0.5 + torch.rand(512, generator=g)).cuda() * torch.where(torch.rand(512, generator=g) < 0.1, -1.0, 1.0).cuda()
    sh = (0.3 * torch.randn(512, generator=g)).cuda()
    w4 = (torch.randn(256, 512, generator=g) * 512 ** -0.5).cuda()
    b4 = (0.1 * torch.randn(256, generator=g)).cuda()
    gmax = torch.randn(tiles, 256, generator=g).cuda()
    # the fold (engine.mini_pointnet): W3s = scale o W3, b3s = scale * b3 + shift
    w3s_a, b3s = ops.scale_rows_convert(w3, sc, T, cols=(0, 256), bias=b3, shift=sh)
    w3s_b = ops.scale_rows_convert(w3, sc, T, cols=(256, 512))
    assert torch.equal(w3s_a, (w3[:, :256] * sc[:, None]).to(T)) and torch.equal(w3s_b, (w3[:, 256:] * sc[:, None]).to(T))
    assert torch.allclose(b3s, sc * b3 + sh, rtol=1e-6, atol=1e-7)
    gs = (gmax @ w3s_a.float().T + b3s) * 3.0                      # (a large group term: the hi + lo split has to carry it)
    w4t = ops.mpn34_retile(w4.to(T))
    tok = ops.mini_pointnet_conv34(y2, w3s_b, gs.contiguous(), w4t, b4)
    tok2 = ops.mini_pointnet_conv34(y2, w3s_b, gs.contiguous(), w4t, b4)
    assert torch.equal(tok, tok2)
    y3 = y2.float() @ w3s_b.float().T + gs.repeat_interleave(32, dim=0)
    a = torch.relu(y3).to(T).float()
    ref = (a @ w4.to(T).float().T + b4).view(tiles, 32, 256).max(dim=1).values
    err = (tok.float() - ref).abs().max().item()
    ulp = 2.0 ** (-10 if T == torch.float16 else -7)
    # one rounding of the output + the rare 1-ulp flip of a ReLU output whose fp32 sum came out in another order
    assert err < 1.5 * ulp * max(1.0, ref.abs().max().item()), err
    # ... and against the UNFUSED kernels (conv3 + affine-prologue conv4) on un-folded weights: same function, other roundings
    if tiles <= 1027:
        gterm = (gs - sh) / sc
        y3u = ops.mini_pointnet_conv3(y2, (w3[:, 256:]).contiguous().to(T), gterm.contiguous())
        toku = ops.mini_pointnet_conv4(y3u, sc, sh, w4.to(T), b4)
        assert (tok.float() - toku.float()).abs().max().item() < 0.05 * max(1.0, ref.abs().max().item())


def test_mini_pointnet_conv3_statistics_pass_without_store(ops):
    """ppt_mini_pointnet_conv3_half with y == NULL: the same BatchNorm partials, bit for bit, as the storing pass."""
    g = torch.Generator().manual_seed(3)
    M = 32 * 333
    y2 = torch.randn(M, 256, generator=g).cuda().to(torch.float16)
    w = (torch.randn(512, 256, generator=g) * 0.06).cuda().to(torch.float16)
    gterm = torch.randn(M // 32, 512, generator=g).cuda()
    st_a = (torch.empty(M // 32, 512, device="cuda"), torch.empty(M // 32, 512, device="cuda"))
    st_b = (torch.empty(M // 32, 512, device="cuda"), torch.empty(M // 32, 512, device="cuda"))
    y = ops.mini_pointnet_conv3(y2, w, gterm, st_a)
    assert ops.mini_pointnet_conv3(y2, w, gterm, st_b, store=False) is None and y is not None
    assert torch.equal(st_a[0], st_b[0]) and torch.equal(st_a[1], st_b[1])


def test_weights_prep_matches_the_per_weight_conversions(ops):
    """ppt_weights_prep (every decoder weight's padded 16-bit operand copy + its transpose in one launch) against the per-weight
    path it replaces (ppt_amd.autograd._pad_k / ops.transpose): bit-identical, including the DGCNN layer's Wb - Wa difference,
    K not a multiple of 8 (zero padding), N not a multiple of 32, and more items than one launch takes."""
    from ppt_amd.autograd import _pad_k
    g = torch.Generator().manual_seed(5)
    for T in (torch.float16, torch.bfloat16):
        ws = [(torch.randn(1536, 387, generator=g) * 0.05).cuda(), (torch.randn(384, 1536, generator=g) * 0.03).cuda(),
              (torch.randn(512, 768, generator=g) * 0.04).cuda(), (torch.randn(128, 384, generator=g) * 0.05).cuda(),
              (torch.randn(50, 403, generator=g) * 0.05).cuda()]
        items = [(ws[0], 0, 387, None, 392), (ws[1], 0, 1536, None, 1536), (ws[2], 0, 384, None, 384), (ws[2], 384, 384, 0, 384),
                 (ws[3], 0, 384, None, 384), (ws[4], 0, 403, None, 408)]
        items = items * 7                                              # 42 items: two launches
        outs = ops.weights_prep(items, T)
        for (w, c0, K, sub, Kp), (o, ot) in zip(items, outs):
            a = w[:, c0:c0 + K] - (w[:, sub:sub + K] if sub is not None else 0)
            ref = _pad_k(a.contiguous(), 8, T)
            assert tuple(o.shape) == (w.shape[0], Kp) and torch.equal(o, ref)
            assert torch.equal(ot, ops.transpose(ref)) and torch.equal(ot, ref.t().contiguous())
