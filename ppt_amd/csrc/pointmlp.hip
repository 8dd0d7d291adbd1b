// pointmlp.hip -- the two PointMLP pieces (models/pointmlp/pointMLP.py) that are not a GEMM, a BatchNorm fold or a gather
// the PointNet2 path already has:
//   * LocalGrouper's "anchor" normalisation (:170-175) divides every cloud's grouped, anchor-centred features by ONE
//     standard deviation over all groups x neighbours x channels: group_anchor_stats gathers the rows once and leaves
//     (sum, sum of squares) per group; the 512..64 group partials per cloud are folded in fp64 by the caller.
//   * ConvBNReLURes1D (:188-221) ends in relu(BN(conv2) + x): bn_res_act_rows applies the folded BatchNorm affine, adds
//     the block input and, for the last block of a PreExtraction stage (:251) / of the network (:332), takes the max over
//     the `pool` consecutive rows of a group so that the un-pooled activation is never written.
// Both are HBM-bound streaming kernels: every row is read exactly once with lanes along the channels.
#include "ppt_common.h"

namespace {

__device__ __forceinline__ float load_any(const void *p, int dtype, int64_t i)
{
    return dtype == PPT_F32 ? ((const float *)p)[i] : bf16_to_f32(((const bf16_t *)p)[i]);
}
__device__ __forceinline__ void store_any(void *p, int dtype, int64_t i, float v)
{
    if (dtype == PPT_F32) ((float *)p)[i] = v;
    else ((bf16_t *)p)[i] = f32_to_bf16(v);
}

// one wave per group (b, s): sum and sum of squares of x[b, idx[b,s,j], :] - x[b, anchor[b,s], :] over j < K, c < D
__global__ __launch_bounds__(256) void group_anchor_stats_kernel(const void *__restrict__ x, int x_dtype, const int64_t *__restrict__ idx,
                                                                  const int64_t *__restrict__ anchor, int Nsrc, int S, int K, int D,
                                                                  int64_t groups, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int64_t b = g / S;
    const int64_t arow = (b * Nsrc + anchor[g]) * D;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float a = load_any(x, x_dtype, arow + c);
        for (int j = 0; j < K; ++j) {
            const float d = load_any(x, x_dtype, (b * Nsrc + idx[g * K + j]) * D + c) - a;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
    }
    s1 = wave_reduce_sum(s1);
    s2 = wave_reduce_sum(s2);
    if (lane == 63) {
        out[2 * g] = s1;
        out[2 * g + 1] = s2;
    }
}

// y[r, c] = relu(scale[c] * x[r, c] + shift[c] + res'[r, c]);  pool > 1: y[g, c] = max over the pool rows of group g.
// res' = res, or relu(res_scale[c] * res[r, c] + res_shift[c]) when the block input is itself a conv output whose
// BatchNorm + ReLU was never materialised (the transfer conv in front of the first block).
// VEC channels per thread (8 when C % 8 == 0: 16-byte bf16 / 2 x 16-byte fp32 accesses).
template <int VEC>
__device__ __forceinline__ void load_vec(const void *p, int dtype, int64_t i, float (&v)[VEC])
{
    if constexpr (VEC == 8) {
        if (dtype == PPT_F32) {
            const float4 a = *reinterpret_cast<const float4 *>((const float *)p + i), b = *reinterpret_cast<const float4 *>((const float *)p + i + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
            const uint4 q = *reinterpret_cast<const uint4 *>((const bf16_t *)p + i);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = __uint_as_float(w[e] << 16);
                v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
            }
        }
    } else {
        v[0] = load_any(p, dtype, i);
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(void *p, int dtype, int64_t i, const float (&v)[VEC])
{
    if constexpr (VEC == 8) {
        if (dtype == PPT_F32) {
            *reinterpret_cast<float4 *>((float *)p + i) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>((float *)p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            uint4 q;
            q.x = pack_bf16x2(v[0], v[1]);
            q.y = pack_bf16x2(v[2], v[3]);
            q.z = pack_bf16x2(v[4], v[5]);
            q.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4 *>((bf16_t *)p + i) = q;
        }
    } else {
        store_any(p, dtype, i, v[0]);
    }
}

template <int VEC, bool RES_AFF, bool BF16IO>
__global__ __launch_bounds__(256) void bn_res_act_rows_kernel(const void *__restrict__ x, int x_dtype, const void *__restrict__ res,
                                                               int res_dtype, int64_t out_rows, int C, int pool,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               const float *__restrict__ res_scale, const float *__restrict__ res_shift,
                                                               void *__restrict__ y, int y_dtype)
{
    if constexpr (BF16IO) x_dtype = res_dtype = y_dtype = PPT_BF16;      // the bf16 pipeline: no dtype branches between the loads
    const int cv = C / VEC;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= out_rows * cv) return;
    const int64_t r = t / cv;
    const int c = (int)(t - r * cv) * VEC;
    float sc[VEC], sh[VEC], rs[VEC], rh[VEC], v[VEC];
    load_vec<VEC>(scale, PPT_F32, c, sc);                                // straight-line vector loads: no per-element branches
    load_vec<VEC>(shift, PPT_F32, c, sh);
    if constexpr (RES_AFF) {
        load_vec<VEC>(res_scale, PPT_F32, c, rs);
        load_vec<VEC>(res_shift, PPT_F32, c, rh);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;                            // relu output is >= 0
    for (int j = 0; j < pool; ++j) {
        const int64_t e0 = (r * pool + j) * C + c;
        float xv[VEC], rv[VEC];
        load_vec<VEC>(x, x_dtype, e0, xv);
        load_vec<VEC>(res, res_dtype, e0, rv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float rr = RES_AFF ? fmaxf(fmaf(rv[e], rs[e], rh[e]), 0.f) : rv[e];
            v[e] = fmaxf(v[e], fmaf(xv[e], sc[e], sh[e]) + rr);
        }
    }
    store_vec<VEC>(y, y_dtype, r * C + c, v);
}

// r[b] = 1 / (std_b + 1e-5), std_b the UNBIASED standard deviation over the n = S * k * d anchor-centred neighbour features of
// cloud b (pointMLP.py:170-175), from the per-group (sum, sum of squares) of group_anchor_stats_kernel, in fp64 as the ATen
// expression it replaces ([B,S,2].double().sum(1) -> var -> sqrt -> + eps -> reciprocal -> float: ~12 launches per stage).
__global__ __launch_bounds__(256) void cloud_rstd_kernel(const float *__restrict__ st, int S, double n, float *__restrict__ r)
{
    __shared__ double sh[2][256];
    const int b = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < S; i += 256) {
        const float2 v = *reinterpret_cast<const float2 *>(st + ((int64_t)b * S + i) * 2);
        s += (double)v.x;
        q += (double)v.y;
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sum = sh[0][0], sq = sh[1][0];
        double var = (sq - sum * sum / n) / (n - 1.0);
        var = var < 0.0 ? 0.0 : var;
        r[b] = (float)(1.0 / (sqrt(var) + 1e-5));
    }
}

// The operands of the transfer conv by linearity (engine.pointmlp_forward): PQ [B*N, 2C] = [(Wa * alpha) . x | Wb . x] ->
//   P[b, n, :] = PQ[b, n, :C] * r[b]                                        (the neighbour term, scaled by the cloud's 1 / std)
//   Q[b, s, :] = (c0 + PQ[b, a, C:]) - P[b, a, :],  a = cidx[b, s]          (the anchor term)
// -- the ATen expression's operations in its order, each rounded once (no contraction), in ONE launch instead of ~12.
__global__ __launch_bounds__(256) void pointmlp_pq_kernel(const float *__restrict__ PQ, const float *__restrict__ r,
                                                          const int64_t *__restrict__ cidx, const float *__restrict__ c0,
                                                          float *__restrict__ P, float *__restrict__ Q, int B, int N, int S, int C)
{
    const int C4 = C >> 2;
    const int64_t nP = (int64_t)B * N * C4, nQ = (int64_t)B * S * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nP + nQ; i += (int64_t)gridDim.x * 256) {
        if (i < nP) {
            const int64_t row = i / C4;
            const int c = (int)(i % C4) * 4;
            const float rb = r[row / N];
            const float4 v = *reinterpret_cast<const float4 *>(PQ + row * 2 * C + c);
            *reinterpret_cast<float4 *>(P + row * C + c) = make_float4(v.x * rb, v.y * rb, v.z * rb, v.w * rb);
        } else {
            const int64_t j = i - nP, qrow = j / C4;
            const int c = (int)(j % C4) * 4;
            const int b = (int)(qrow / S);
            const int64_t a = (int64_t)b * N + cidx[qrow];
            const float rb = r[b];
            const float4 pa = *reinterpret_cast<const float4 *>(PQ + a * 2 * C + c);
            const float4 qa = *reinterpret_cast<const float4 *>(PQ + a * 2 * C + C + c);
            const float4 cc = *reinterpret_cast<const float4 *>(c0 + c);
            // (the products are pinned in registers before the subtraction: -ffp-contract=fast would otherwise fold them into an
            // fma and round once where the ATen expression rounds twice)
            float p0 = pa.x * rb, p1 = pa.y * rb, p2 = pa.z * rb, p3 = pa.w * rb;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            *reinterpret_cast<float4 *>(Q + qrow * C + c) =
                make_float4((cc.x + qa.x) - p0, (cc.y + qa.y) - p1, (cc.z + qa.z) - p2, (cc.w + qa.w) - p3);
        }
    }
}

}  // namespace

extern "C" int ppt_pointmlp_cloud_rstd(const float *stats, int B, int S, double n, float *r, void *stream)
{
    if (!stats || !r || B <= 0 || S <= 0 || !(n > 1.0)) return PPT_EINVAL;
    hipLaunchKernelGGL(cloud_rstd_kernel, dim3(B), dim3(256), 0, ppt_stream(stream), stats, S, n, r);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_pointmlp_pq(const float *PQ, const float *r, const int64_t *cidx, const float *c0, float *P, float *Q, int B, int N,
                               int S, int C, void *stream)
{
    if (!PQ || !r || !cidx || !c0 || !P || !Q || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PPT_EINVAL;
    if ((C % 4) || (((uintptr_t)PQ | (uintptr_t)c0 | (uintptr_t)P | (uintptr_t)Q) & 15)) return PPT_EUNSUPPORTED;
    const int64_t n = ((int64_t)B * N + (int64_t)B * S) * (C / 4);
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(pointmlp_pq_kernel, dim3(blocks), dim3(256), 0, ppt_stream(stream), PQ, r, cidx, c0, P, Q, B, N, S, C);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_group_anchor_stats(const void *x, int x_dtype, const int64_t *idx, const int64_t *anchor, int B, int Nsrc, int S,
                                      int K, int D, float *out, void *stream)
{
    if (!x || !idx || !anchor || !out || B <= 0 || Nsrc <= 0 || S <= 0 || K <= 0 || D <= 0) return PPT_EINVAL;
    if (x_dtype != PPT_F32 && x_dtype != PPT_BF16) return PPT_EINVAL;
    const int64_t groups = (int64_t)B * S;
    hipLaunchKernelGGL(group_anchor_stats_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, ppt_stream(stream), x, x_dtype,
                       idx, anchor, Nsrc, S, K, D, groups, out);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_bn_res_act_rows(const void *x, int x_dtype, const void *res, int res_dtype, int64_t M, int C, int pool,
                                   const float *scale, const float *shift, const float *res_scale, const float *res_shift, void *y,
                                   int y_dtype, void *stream)
{
    if (!x || !res || !scale || !shift || !y || M <= 0 || C <= 0 || pool <= 0 || M % pool) return PPT_EINVAL;
    if ((res_scale == nullptr) != (res_shift == nullptr)) return PPT_EINVAL;
    for (int d : {x_dtype, res_dtype, y_dtype})
        if (d != PPT_F32 && d != PPT_BF16) return PPT_EINVAL;
    const bool vec = C % 8 == 0 && !(((uintptr_t)x | (uintptr_t)res | (uintptr_t)y | (uintptr_t)scale | (uintptr_t)shift |
                                      (uintptr_t)res_scale | (uintptr_t)res_shift) & 15);
    const int64_t n = M / pool * (vec ? C / 8 : C);
    const dim3 grid((unsigned)((n + 255) / 256));
#define PPT_LAUNCH_BRA(V, R, H)                                                                                         \
    hipLaunchKernelGGL((bn_res_act_rows_kernel<V, R, H>), grid, dim3(256), 0, ppt_stream(stream), x, x_dtype, res, res_dtype, \
                       M / pool, C, pool, scale, shift, res_scale, res_shift, y, y_dtype)
    const bool h = x_dtype == PPT_BF16 && res_dtype == PPT_BF16 && y_dtype == PPT_BF16;
    if (vec && h && res_scale) PPT_LAUNCH_BRA(8, true, true);
    else if (vec && h) PPT_LAUNCH_BRA(8, false, true);
    else if (vec && res_scale) PPT_LAUNCH_BRA(8, true, false);
    else if (vec) PPT_LAUNCH_BRA(8, false, false);
    else if (res_scale) PPT_LAUNCH_BRA(1, true, false);
    else PPT_LAUNCH_BRA(1, false, false);
#undef PPT_LAUNCH_BRA
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
