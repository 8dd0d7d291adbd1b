// interp.hip -- the memory-bound glue of the part-segmentation decoder (models/pointbert/pointnet2_utils.py:297-467):
//   * three_nn_interp_fwd: PointNetFeaturePropagation's inverse-distance interpolation over the 3 nearest sources
//     (:333-351: recip = 1 / (d + 1e-8), weight = recip / sum, sum of the three weighted rows) fused with the
//     concatenation in front of the first conv (:353-358): ONE pass writes the GEMM's A operand
//     [points1 | interpolated | zero padding] in the operand dtype -- the gathered [B,N,3,C] tensor, the weighted
//     product, the sum, torch.cat, the K padding and the dtype conversion (six ATen kernels) never exist;
//   * scatter_rows_bwd: the backward of a row gather, dSrc[b, s, :] = sum over the entries e with idx[b, e] == s of
//     w[b, e] * dRows[b, e / rows_div, col_off + :].  OWNER-COMPUTES: one wave per source row scans the cloud's index list
//     (staged in LDS) in ascending order and adds the matching rows -- no atomics, no sort, deterministic by
//     construction.  Serves the interpolation (3 entries per target, weights) and DGCNN_Propagation's neighbour
//     gather (k entries per query, no weights: autograd's index_put_(accumulate) runs a radix sort for it);
//   * sum_groups: out[g, :] = sum_j x[g, j, :] (the gradient of the query term broadcast over the k neighbours).
#include "ppt_common.h"

namespace {

template <typename TO> struct pair_store;
template <> struct pair_store<float> {
    static __device__ __forceinline__ void put(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
};
template <> struct pair_store<bf16_t> {
    static __device__ __forceinline__ void put(bf16_t *p, float a, float b) { *reinterpret_cast<uint32_t *>(p) = pack_bf16x2(a, b); }
};
template <> struct pair_store<f16_t> {
    static __device__ __forceinline__ void put(f16_t *p, float a, float b) { *reinterpret_cast<uint32_t *>(p) = pack_f16x2(a, b); }
};

// one thread per PAIR of output columns (c, c + 1) of one row; ld is even
template <typename TO>
__global__ __launch_bounds__(256) void three_nn_interp_fwd_kernel(const float *__restrict__ p1, int D1, const float *__restrict__ p2, int D2,
                                                                   const int64_t *__restrict__ idx, const float *__restrict__ dist,
                                                                   int N, int S, int64_t rows, int ld, TO *__restrict__ out,
                                                                   float *__restrict__ wout)
{
    const int half = ld >> 1;
    const int64_t total = rows * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / half;
        const int c = 2 * (int)(i - row * half);
        const int64_t b = row / N;
        // weights (pointnet2_utils.py:343-347), recomputed per thread from the row's three distances (12 bytes, cached)
        const float d0 = dist[row * 3], d1 = dist[row * 3 + 1], d2 = dist[row * 3 + 2];
        const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
        const float norm = (r0 + r1) + r2;
        const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
        if (c == 0 && wout) { wout[row * 3] = w0; wout[row * 3 + 1] = w1; wout[row * 3 + 2] = w2; }
        const float *s0 = p2 + (b * S + idx[row * 3]) * D2, *s1 = p2 + (b * S + idx[row * 3 + 1]) * D2,
                    *s2 = p2 + (b * S + idx[row * 3 + 2]) * D2;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cc = c + e;
            if (cc < D1) v[e] = p1[row * D1 + cc];
            else if (cc < D1 + D2) {
                const int ch = cc - D1;
                v[e] = (s0[ch] * w0 + s1[ch] * w1) + s2[ch] * w2;
            } else v[e] = 0.f;
        }
        pair_store<TO>::put(out + row * ld + c, v[0], v[1]);
    }
}

constexpr int SC_CHUNK = 8192;          // index entries staged per pass: 16 KB of uint16 + 32 KB of weights

// grid (ceil(S / 4 / SPW), B); 4 waves; wave -> sources s = (blockIdx.x * 4 + wave) * SPW ... + SPW
template <int CPL, bool HAS_W>
__global__ __launch_bounds__(256) void scatter_rows_bwd_kernel(const int64_t *__restrict__ idx, const float *__restrict__ w,
                                                                const float *__restrict__ d_rows, int64_t ld, int col_off, int E,
                                                                int rows_div, int S, int C, int spw, float *__restrict__ d_src)
{
    __shared__ uint16_t s_idx[SC_CHUNK];
    __shared__ float s_w[HAS_W ? SC_CHUNK : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.y;
    const int s_first = (blockIdx.x * 4 + wave) * spw;
    const int64_t *ib = idx + b * E;
    const float *wb = HAS_W ? w + b * E : nullptr;
    const float *db = d_rows + b * (int64_t)(E / rows_div) * ld + col_off;
    for (int s0 = 0; s0 < spw; ++s0) {                                    // (chunks re-staged per source pass only when E > SC_CHUNK)
        const int s = s_first + s0;
        float acc[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
        for (int e_base = 0; e_base < E; e_base += SC_CHUNK) {
            const int n_e = min(SC_CHUNK, E - e_base);
            if (s0 == 0 || E > SC_CHUNK) {
                __syncthreads();
                for (int e = threadIdx.x; e < n_e; e += 256) {
                    s_idx[e] = (uint16_t)ib[e_base + e];
                    if constexpr (HAS_W) s_w[e] = wb[e_base + e];
                }
                __syncthreads();
            }
            if (s < S) {
                for (int e0 = 0; e0 < n_e; e0 += 64) {
                    const int e = e0 + lane;
                    uint64_t hit = __ballot(e < n_e && s_idx[e] == (uint16_t)s);
                    // ascending entry order: a fixed summation order.  FOUR matching rows are requested before any is added --
                    // one row at a time, every hit was a dependent global round trip (169 us for the 2048 x 3 -> 512 gather of
                    // the part-seg decoder); the additions keep their order, so the sums are the same bits
                    while (hit) {
                        int ee[4];
                        float wv[4], v[4][CPL];
                        int nh = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ee[u] = -1;
                            if (hit) {
                                ee[u] = e0 + __builtin_ctzll(hit);
                                hit &= hit - 1;
                                nh = u + 1;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (u < nh) {
                                wv[u] = HAS_W ? s_w[ee[u]] : 1.0f;
                                const float *src = db + (int64_t)((e_base + ee[u]) / rows_div) * ld;
#pragma unroll
                                for (int k = 0; k < CPL; ++k) {
                                    const int c = lane + 64 * k;
                                    v[u][k] = c < C ? src[c] : 0.f;
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (u < nh) {
#pragma unroll
                                for (int k = 0; k < CPL; ++k) {
                                    const int c = lane + 64 * k;
                                    if (c < C) acc[k] = HAS_W ? fmaf(wv[u], v[u][k], acc[k]) : acc[k] + v[u][k];
                                }
                            }
                        }
                    }
                }
            }
        }
        if (s < S) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = lane + 64 * k;
                if (c < C) d_src[(b * S + s) * C + c] = acc[k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void sum_groups_kernel(const float *__restrict__ x, int64_t G, int k, int C, float *__restrict__ out)
{
    const int c4 = C >> 2;
    const int64_t total = G * c4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / c4;
        const int c = 4 * (int)(i - g * c4);
        const float *src = x + (g * k) * C + c;
        float4 a = *reinterpret_cast<const float4 *>(src);
        for (int j = 1; j < k; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)j * C);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + g * C + c) = a;
    }
}

}  // namespace

extern "C" int ppt_three_nn_interp_fwd(const float *points1, int D1, const float *points2, int D2, const int64_t *idx,
                                       const float *dist, int B, int N, int S, void *out, int out_dtype, int ld_out,
                                       float *weight_out, void *stream)
{
    if (!points2 || !idx || !dist || !out || B <= 0 || N <= 0 || S <= 0 || D2 <= 0 || D1 < 0 || (D1 > 0 && !points1)) return PPT_EINVAL;
    if (ld_out < D1 + D2 || (ld_out & 1)) return PPT_EINVAL;
    if (out_dtype != PPT_F32 && out_dtype != PPT_BF16 && out_dtype != PPT_F16) return PPT_EINVAL;
    const int64_t rows = (int64_t)B * N;
    const int64_t total = rows * (ld_out / 2);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (out_dtype == PPT_BF16)
        hipLaunchKernelGGL(three_nn_interp_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, ppt_stream(stream), points1, D1, points2, D2, idx,
                           dist, N, S, rows, ld_out, (bf16_t *)out, weight_out);
    else if (out_dtype == PPT_F16)
        hipLaunchKernelGGL(three_nn_interp_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, ppt_stream(stream), points1, D1, points2, D2, idx,
                           dist, N, S, rows, ld_out, (f16_t *)out, weight_out);
    else
        hipLaunchKernelGGL(three_nn_interp_fwd_kernel<float>, dim3(grid), dim3(256), 0, ppt_stream(stream), points1, D1, points2, D2, idx,
                           dist, N, S, rows, ld_out, (float *)out, weight_out);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_scatter_rows_bwd(const int64_t *idx, const float *weight, const float *d_rows, int64_t ld, int col_off, int B, int E,
                                    int rows_div, int S, int C, float *d_src, void *stream)
{
    if (!idx || !d_rows || !d_src || B <= 0 || E <= 0 || S <= 0 || C <= 0 || rows_div <= 0 || E % rows_div || col_off < 0 || ld < col_off + C)
        return PPT_EINVAL;
    if (S > 65535 || C > 512) return PPT_EUNSUPPORTED;
    // sources per wave: every workgroup stages the cloud's whole index list, so a workgroup should own many sources -- but
    // the launch still wants ~2 workgroups per CU
    int spw = 8;
    while (spw > 1 && (int64_t)B * ((S + 4 * spw - 1) / (4 * spw)) < 512) spw >>= 1;
    const dim3 grid((S + 4 * spw - 1) / (4 * spw), B);
    hipStream_t s = ppt_stream(stream);
#define PPT_SC(CPL)                                                                                                            \
    do {                                                                                                                       \
        if (weight) hipLaunchKernelGGL((scatter_rows_bwd_kernel<CPL, true>), grid, dim3(256), 0, s, idx, weight, d_rows, ld, col_off, E, \
                                       rows_div, S, C, spw, d_src);                                                            \
        else hipLaunchKernelGGL((scatter_rows_bwd_kernel<CPL, false>), grid, dim3(256), 0, s, idx, weight, d_rows, ld, col_off, E, \
                                rows_div, S, C, spw, d_src);                                                                   \
    } while (0)
    if (C <= 128) PPT_SC(2);
    else if (C <= 256) PPT_SC(4);
    else if (C <= 384) PPT_SC(6);
    else PPT_SC(8);
#undef PPT_SC
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_sum_groups(const float *x, int64_t G, int k, int C, float *out, void *stream)
{
    if (!x || !out || G <= 0 || k <= 0 || C <= 0) return PPT_EINVAL;
    if (C % 4 || (((uintptr_t)x | (uintptr_t)out) & 15)) return PPT_EUNSUPPORTED;
    const int64_t total = G * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(sum_groups_kernel, dim3(grid), dim3(256), 0, ppt_stream(stream), x, G, k, C, out);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
