// groupnorm.hip -- GroupNorm + LeakyReLU + max over the k neighbours of DGCNN_Propagation (pointbert/pointnet2_utils.py
// :371-467: Conv2d(1x1, no bias) -> GroupNorm(4, C) -> LeakyReLU(0.2) -> max over k), forward and backward, on the
// channels-last rows the 1x1 conv GEMM writes: y [B][Q][K][C] fp32.  torch runs this as permute-copy + RowwiseMoments +
// elementwise + reduce (+ their backward kernels); here the normalised tensor is never written:
//   forward : (sum, sumsq) per (cloud, row chunk, group) -> [host fold: mean, rstd per (cloud, group)] -> one pass that
//             normalises, applies LeakyReLU and keeps the maximum over k with its index;
//   backward: the gradient enters only at the arg-max elements; one pass forms the two group sums of the GroupNorm
//             backward (and dgamma / dbeta partials), one pass writes dy for every element.
// All reductions are chunked with a fixed fold order (no atomics).  HBM-bound: y is read once per pass.
#include "ppt_common.h"

namespace {

constexpr int GN_ROWS = 64;       // rows (q, k) per statistics workgroup
constexpr int GN_Q = 32;          // query points per backward-sum workgroup
constexpr int GN_MAXC = 1024;

__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ y, int R, int C, int G, double *__restrict__ part)
{
    __shared__ float ssum[GN_MAXC], ssq[GN_MAXC];
    const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
    const int r0 = chunk * GN_ROWS, r1 = min(R, r0 + GN_ROWS);
    if ((C & 3) == 0 && (((uintptr_t)y) & 15) == 0) {
        // 16 row groups x 16 column quads per pass of 64 columns: float4 loads (256-byte row pieces), the 16 group sums folded
        // through LDS in group order (the one-thread-per-column walk ran at 1.8 TB/s)
        __shared__ float4 rs[16][17], rq[16][17];
        const int q4 = threadIdx.x & 15, g = threadIdx.x >> 4;
        for (int c0 = 0; c0 < C; c0 += 64) {
            const int c = c0 + q4 * 4;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
            if (c < C) {
#pragma unroll 4
                for (int r = r0 + g; r < r1; r += 16) {
                    const float4 v = *reinterpret_cast<const float4 *>(y + ((size_t)b * R + r) * C + c);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                    q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
                }
            }
            rs[g][q4] = s; rq[g][q4] = q;
            __syncthreads();
            if (g == 0 && c < C) {
                float4 ts = rs[0][q4], tq = rq[0][q4];
#pragma unroll
                for (int k = 1; k < 16; ++k) {
                    const float4 a = rs[k][q4], d = rq[k][q4];
                    ts.x += a.x; ts.y += a.y; ts.z += a.z; ts.w += a.w;
                    tq.x += d.x; tq.y += d.y; tq.z += d.z; tq.w += d.w;
                }
                ssum[c] = ts.x; ssum[c + 1] = ts.y; ssum[c + 2] = ts.z; ssum[c + 3] = ts.w;
                ssq[c] = tq.x; ssq[c + 1] = tq.y; ssq[c + 2] = tq.z; ssq[c + 3] = tq.w;
            }
            __syncthreads();
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f, q = 0.f;
            for (int r = r0; r < r1; ++r) {
                const float v = y[((size_t)b * R + r) * C + c];
                s += v;
                q = fmaf(v, v, q);
            }
            ssum[c] = s;
            ssq[c] = q;
        }
    }
    __syncthreads();
    const int Cg = C / G;
    if ((int)threadIdx.x < G) {
        double s = 0.0, q = 0.0;
        for (int c = threadIdx.x * Cg; c < ((int)threadIdx.x + 1) * Cg; ++c) { s += (double)ssum[c]; q += (double)ssq[c]; }
        double *o = part + (((size_t)b * nch + chunk) * G + threadIdx.x) * 2;
        o[0] = s;
        o[1] = q;
    }
}

// out[b,q,c] = max_j lrelu(gamma[c] * (y[b,q,j,c] - mean[b,g]) * rstd[b,g] + beta[c]), arg = first maximising j
__global__ __launch_bounds__(256) void gn_lrelu_max_kernel(const float *__restrict__ y, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int64_t total, int Q, int K, int C, int G,
                                                            float slope, float *__restrict__ out, int32_t *__restrict__ arg)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t bq = idx / C;
    const int b = (int)(bq / Q), g = c / (C / G);
    const float a = rstd[b * G + g] * gamma[c];
    const float sh = fmaf(-mean[b * G + g], a, beta[c]);
    float best = -INFINITY;
    int bi = 0;
    for (int j = 0; j < K; ++j) {
        float v = fmaf(y[(bq * K + j) * C + c], a, sh);
        v = v > 0.f ? v : slope * v;
        if (v > best) { best = v; bi = j; }
    }
    out[idx] = best;
    arg[idx] = bi;
}

// per (cloud b, chunk of GN_Q query points): group sums S1 = sum gamma*t, S2 = sum gamma*t*xhat over the arg-max elements,
// and per-channel partials of dgamma = sum t*xhat, dbeta = sum t;  t = dout * (out > 0 ? 1 : slope)
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const float *__restrict__ y, const float *__restrict__ dout,
                                                           const float *__restrict__ out, const int32_t *__restrict__ arg,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           const float *__restrict__ gamma, int Q, int K, int C, int G, float slope,
                                                           double *__restrict__ psum, float *__restrict__ pgb)
{
    __shared__ float s1[GN_MAXC], s2[GN_MAXC];
    const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
    const int q0 = chunk * GN_Q, q1 = min(Q, q0 + GN_Q);
    const int Cg = C / G;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / Cg;
        const float m = mean[b * G + g], rs = rstd[b * G + g], ga = gamma[c];
        float a1 = 0.f, a2 = 0.f, dg = 0.f, db = 0.f;
#pragma unroll 8                                          // (each query is two dependent loads -- arg, then y at arg: keep eight in flight)
        for (int q = q0; q < q1; ++q) {
            const int64_t o = ((int64_t)b * Q + q) * C + c;
            const float t = dout[o] * (out[o] > 0.f ? 1.f : slope);
            const float xh = (y[(((int64_t)b * Q + q) * K + arg[o]) * C + c] - m) * rs;
            a1 = fmaf(ga, t, a1);
            a2 = fmaf(ga * t, xh, a2);
            dg = fmaf(t, xh, dg);
            db += t;
        }
        s1[c] = a1;
        s2[c] = a2;
        float *pg = pgb + (((size_t)b * nch + chunk) * C + c) * 2;
        pg[0] = dg;
        pg[1] = db;
    }
    __syncthreads();
    if ((int)threadIdx.x < G) {
        double a = 0.0, d = 0.0;
        for (int c = threadIdx.x * Cg; c < ((int)threadIdx.x + 1) * Cg; ++c) { a += (double)s1[c]; d += (double)s2[c]; }
        double *o = psum + (((size_t)b * nch + chunk) * G + threadIdx.x) * 2;
        o[0] = a;
        o[1] = d;
    }
}

// dy[b,q,j,c] = rstd * ((j == arg ? gamma*t : 0) - (S1 + xhat * S2) / n),  S1/S2 already divided by n on the host
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *__restrict__ y, const float *__restrict__ dout,
                                                            const float *__restrict__ out, const int32_t *__restrict__ arg,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ s12n,
                                                            int64_t total, int Q, int K, int C, int G, float slope,
                                                            float *__restrict__ dy)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t bq = idx / C;
    const int b = (int)(bq / Q), g = c / (C / G);
    const float m = mean[b * G + g], rs = rstd[b * G + g];
    const float S1 = s12n[(b * G + g) * 2], S2 = s12n[(b * G + g) * 2 + 1];
    const float gt = gamma[c] * dout[idx] * (out[idx] > 0.f ? 1.f : slope);
    const int am = arg[idx];
    for (int j = 0; j < K; ++j) {
        const int64_t e = (bq * K + j) * C + c;
        const float xh = (y[e] - m) * rs;
        dy[e] = rs * ((j == am ? gt : 0.f) - fmaf(xh, S2, S1));
    }
}

// the (cloud, group) statistics out of the chunk partials, one thread each: mode 0 -> mean = S0 / n, rstd = 1 / sqrt(max(S1 / n -
// mean^2, 0) + eps) (biased variance, as nn.GroupNorm); mode 1 -> (S0 / n, S1 / n) (the two group sums of the backward).  Sums in
// chunk order, fp64.  Replaces ~9 (forward) / 3 (backward) tiny ATen kernels per GroupNorm layer.
__global__ __launch_bounds__(256) void gn_finish_kernel(const double *__restrict__ part, int BG, int nch, int G, double n, double eps,
                                                        int mode, float *__restrict__ o0, float *__restrict__ o1)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BG) return;
    const int b = i / G, g = i - b * G;
    double s0 = 0.0, s1 = 0.0;
    for (int c = 0; c < nch; ++c) {
        const double *q = part + (((size_t)b * nch + c) * G + g) * 2;
        s0 += q[0]; s1 += q[1];
    }
    if (mode == 0) {
        const double mean = s0 / n;
        double var = s1 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        o0[i] = (float)mean;
        o1[i] = (float)(1.0 / sqrt(var + eps));
    } else {
        o0[2 * i] = (float)(s0 / n);
        o0[2 * i + 1] = (float)(s1 / n);
    }
}

}  // namespace

extern "C" int ppt_gn_finish(const double *part, int B, int nch, int G, double n, double eps, int mode, float *out0, float *out1,
                             void *stream)
{
    if (!part || !out0 || B <= 0 || nch <= 0 || G <= 0 || n <= 0 || (mode == 0 && !out1) || mode < 0 || mode > 1) return PPT_EINVAL;
    hipLaunchKernelGGL(gn_finish_kernel, dim3((B * G + 255) / 256), dim3(256), 0, ppt_stream(stream), part, B * G, nch, G, n, eps, mode,
                       out0, out1);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_gn_stats_chunks(int R) { return (R + GN_ROWS - 1) / GN_ROWS; }
extern "C" int ppt_gn_bwd_chunks(int Q) { return (Q + GN_Q - 1) / GN_Q; }

extern "C" int ppt_gn_stats(const float *y, int B, int R, int C, int G, double *part, void *stream)
{
    if (!y || !part || B <= 0 || R <= 0 || C <= 0 || G <= 0 || C % G || C > GN_MAXC || G > 256) return PPT_EINVAL;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(ppt_gn_stats_chunks(R), B), dim3(256), 0, ppt_stream(stream), y, R, C, G, part);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_gn_lrelu_max(const float *y, const float *mean, const float *rstd, const float *gamma, const float *beta, int B,
                                int Q, int K, int C, int G, float slope, float *out, int32_t *arg, void *stream)
{
    if (!y || !mean || !rstd || !gamma || !beta || !out || !arg || B <= 0 || Q <= 0 || K <= 0 || C <= 0 || G <= 0 || C % G)
        return PPT_EINVAL;
    const int64_t total = (int64_t)B * Q * C;
    hipLaunchKernelGGL(gn_lrelu_max_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ppt_stream(stream), y, mean, rstd,
                       gamma, beta, total, Q, K, C, G, slope, out, arg);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_gn_bwd_sums(const float *y, const float *dout, const float *out, const int32_t *arg, const float *mean,
                               const float *rstd, const float *gamma, int B, int Q, int K, int C, int G, float slope, double *psum,
                               float *pgb, void *stream)
{
    if (!y || !dout || !out || !arg || !mean || !rstd || !gamma || !psum || !pgb || B <= 0 || Q <= 0 || K <= 0 || C <= 0 ||
        G <= 0 || C % G || C > GN_MAXC || G > 256)
        return PPT_EINVAL;
    hipLaunchKernelGGL(gn_bwd_sums_kernel, dim3(ppt_gn_bwd_chunks(Q), B), dim3(256), 0, ppt_stream(stream), y, dout, out, arg, mean,
                       rstd, gamma, Q, K, C, G, slope, psum, pgb);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_gn_bwd_apply(const float *y, const float *dout, const float *out, const int32_t *arg, const float *mean,
                                const float *rstd, const float *gamma, const float *s12n, int B, int Q, int K, int C, int G,
                                float slope, float *dy, void *stream)
{
    if (!y || !dout || !out || !arg || !mean || !rstd || !gamma || !s12n || !dy || B <= 0 || Q <= 0 || K <= 0 || C <= 0 || G <= 0 ||
        C % G)
        return PPT_EINVAL;
    const int64_t total = (int64_t)B * Q * C;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ppt_stream(stream), y, dout, out, arg,
                       mean, rstd, gamma, s12n, total, Q, K, C, G, slope, dy);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
