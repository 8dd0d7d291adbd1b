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

// y[r, c] = relu(scale[c] * x[r, c] + shift[c] + res[r, c]);  pool > 1: y[g, c] = max over the pool rows of group g
__global__ void bn_res_act_rows_kernel(const void *__restrict__ x, int x_dtype, const void *__restrict__ res, int res_dtype,
                                       int64_t out_rows, int C, int pool, const float *__restrict__ scale,
                                       const float *__restrict__ shift, void *__restrict__ y, int y_dtype)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_rows * C) return;
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const float sc = scale[c], sh = shift[c];
    float v = 0.f;                                                       // relu output is >= 0
    for (int j = 0; j < pool; ++j) {
        const int64_t e = (r * pool + j) * C + c;
        v = fmaxf(v, fmaf(load_any(x, x_dtype, e), sc, sh) + load_any(res, res_dtype, e));
    }
    store_any(y, y_dtype, i, v);
}

}  // namespace

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
                                   const float *scale, const float *shift, void *y, int y_dtype, void *stream)
{
    if (!x || !res || !scale || !shift || !y || M <= 0 || C <= 0 || pool <= 0 || M % pool) return PPT_EINVAL;
    for (int d : {x_dtype, res_dtype, y_dtype})
        if (d != PPT_F32 && d != PPT_BF16) return PPT_EINVAL;
    const int64_t n = M / pool * C;
    hipLaunchKernelGGL(bn_res_act_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ppt_stream(stream), x, x_dtype, res,
                       res_dtype, M / pool, C, pool, scale, shift, y, y_dtype);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
