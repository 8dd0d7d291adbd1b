// pointnet.hip -- BatchNorm bookkeeping of the mini-PointNet (dvae.py:188-199) and the K=3 layers.
//
// model.train() keeps the "frozen" tokenizer's BatchNorm layers in batch-statistics mode
// (SURVEY.md App. A Q3), which puts two global reductions inside the GEMM chain:
//   BN1: statistics of y1 = w1.p + b1 (K=3) -- ppt_conv1_stats recomputes y1 on the VALU from the
//        12-byte points (y1 itself is never stored: the conv2 GEMM rebuilds it in its A prologue);
//   BN2: statistics of the 512-wide conv3 output -- produced by the GEMM epilogue (col_sum/col_sqsum).
// Both land in [partials, C] fp32 buffers; ppt_bn_finalize folds them in fp64 (deterministic, no
// atomics), emits the per-channel affine (scale, shift) the next GEMM applies on its A operand and
// updates running_mean / running_var / num_batches_tracked exactly as nn.BatchNorm1d does.
#include "ppt_common.h"

namespace {

constexpr int STAT_ROWS = 2048;   // points per partial

// y_c = w_c . p + b_c is linear in the point, so the chunk statistics of all C channels follow from the chunk's nine
// point moments (sum p, sum p p^T): sum_c = w_c . S1 + n b_c,  M2_c = w_c^T (S2 - S1 S1^T / n) w_c.  The moments are
// accumulated in fp64 (exact products of fp32 coordinates), so the result is the exactly-rounded statistic of the
// un-rounded y -- and the kernel reads each point once instead of evaluating 2 x C x rows FMAs (70 -> ~8 us).
__global__ __launch_bounds__(256) void conv1_stats_kernel(const float *__restrict__ pts, int64_t M,
                                                          const float *__restrict__ w1, const float *__restrict__ b1,
                                                          int C, float *__restrict__ psum, float *__restrict__ psq)
{
    __shared__ double red[9][256];
    const int64_t r0 = (int64_t)blockIdx.x * STAT_ROWS;
    const int nrow = (int)min((int64_t)STAT_ROWS, M - r0);
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = threadIdx.x; r < nrow; r += blockDim.x) {
        const double x = pts[(r0 + r) * 3 + 0], y = pts[(r0 + r) * 3 + 1], z = pts[(r0 + r) * 3 + 2];
        m[0] += x; m[1] += y; m[2] += z;
        m[3] += x * x; m[4] += x * y; m[5] += x * z; m[6] += y * y; m[7] += y * z; m[8] += z * z;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[k][threadIdx.x] = m[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int k = 0; k < 9; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        }
        __syncthreads();
    }
    const double n = (double)nrow;
    const double sx = red[0][0], sy = red[1][0], sz = red[2][0];
    const double cxx = red[3][0] - sx * sx / n, cxy = red[4][0] - sx * sy / n, cxz = red[5][0] - sx * sz / n;
    const double cyy = red[6][0] - sy * sy / n, cyz = red[7][0] - sy * sz / n, czz = red[8][0] - sz * sz / n;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double wx = w1[c * 3 + 0], wy = w1[c * 3 + 1], wz = w1[c * 3 + 2], wb = b1[c];
        const double q = wx * wx * cxx + wy * wy * cyy + wz * wz * czz + 2.0 * (wx * wy * cxy + wx * wz * cxz + wy * wz * cyz);
        psum[(size_t)blockIdx.x * C + c] = (float)(wx * sx + wy * sy + wz * sz + n * wb);
        psq[(size_t)blockIdx.x * C + c] = (float)fmax(q, 0.0);
    }
}

// (count, mean, M2) triples merge associatively (Chan et al.): every thread folds a stripe of the chunk
// partials in fp64, then the 128 stripes of a channel are folded through LDS.  8 channels per block
// (32-byte pieces of the [P, C] partial rows; C/8 = 64 workgroups for the 512-channel layer, whose
// P = M/32 = 16 384 partial rows would otherwise be walked by 16 workgroups only), 128 stripes.
struct stat3 { double n, mean, m2; };
__device__ __forceinline__ void stat_merge(stat3 &a, const stat3 &b)
{
    if (b.n <= 0.0) return;
    const double tot = a.n + b.n;
    const double d = b.mean - a.mean;
    a.mean += d * (b.n / tot);
    a.m2 += b.m2 + d * d * (a.n * b.n / tot);
    a.n = tot;
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float *__restrict__ psum, const float *__restrict__ psq,
                                                          int P, double count, int rpp, int C,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float eps, int train, float momentum, float *__restrict__ rmean,
                                                          float *__restrict__ rvar, int64_t *__restrict__ nbt,
                                                          float *__restrict__ scale, float *__restrict__ shift,
        float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    constexpr int CH = 8, ST = 128;
    __shared__ stat3 red[ST][CH + 1];
    const int tx = threadIdx.x & (CH - 1), ty = threadIdx.x / CH;
    const int c = blockIdx.x * CH + tx;
    float mean_f = 0.f, var_f = 1.f;
    if (train) {
        // Each stripe folds its partials as three fp64 sums -- S1 = sum of sums, S2 = sum of M2, S3 = sum of sum^2 / n --
        // from which (n, mean, M2) of the stripe follow exactly as from a chain of Chan merges (M2 = S2 + S3 - S1^2 / N;
        // fp64 leaves ~1e-16 x N x mean^2 of cancellation error, far below the fp32 inputs), without a division per
        // partial: the walk over the 16 384 x 512 partials of conv3 is then bandwidth-bound (55 -> ~20 us).
        double s1 = 0.0, s2 = 0.0, s3 = 0.0, ntot = 0.0;
        const double inv_rpp = 1.0 / (double)rpp;
        if (c < C)
            for (int p = ty; p < P; p += ST) {
                const double n = fmin((double)rpp, count - (double)p * rpp);
                if (n > 0.0) {
                    const double sv = (double)psum[(size_t)p * C + c];
                    s1 += sv; s2 += (double)psq[(size_t)p * C + c];
                    s3 += sv * sv * (n == (double)rpp ? inv_rpp : 1.0 / n);
                    ntot += n;
                }
            }
        stat3 acc{0.0, 0.0, 0.0};
        if (ntot > 0.0) acc = stat3{ntot, s1 / ntot, fmax(s2 + s3 - s1 * s1 / ntot, 0.0)};
        red[ty][tx] = acc;
        __syncthreads();
        for (int off = ST / 2; off > 0; off >>= 1) {
            if (ty < off) { stat3 a = red[ty][tx]; stat_merge(a, red[ty + off][tx]); red[ty][tx] = a; }
            __syncthreads();
        }
        if (ty != 0 || c >= C) return;
        const stat3 t = red[0][tx];
        const double var = t.m2 / count;                // biased variance normalises (nn.BatchNorm1d)
        mean_f = (float)t.mean; var_f = (float)var;
        if (rmean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean_f;
            rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unbiased;
            if (c == 0 && nbt) *nbt += 1;
        }
    } else {
        if (ty != 0 || c >= C) return;
        mean_f = rmean[c]; var_f = rvar[c];
    }
    const float sc = gamma[c] / sqrtf(var_f + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mean_f * sc;
    if (mean_out) { mean_out[c] = mean_f; rstd_out[c] = 1.0f / sqrtf(var_f + eps); }
}

// ---- two-stage fold for many partials (conv3: 16 384 x 512) -----------------------------------------------------
// stage 1: workgroup (channel group of 32, split) walks its slice of the partial rows with full 128-byte rows per half
// wave and adds up, per channel, the four fp64 sums {S1 = sum of sums, S2 = sum of M2, S3 = sum of sum^2 / n, N};
// they are plainly additive, so the splits are reduced by summation in stage 2 (M2 = S2 + S3 - S1^2 / N).
__global__ __launch_bounds__(1024) void bn_fold_kernel(const float *__restrict__ psum, const float *__restrict__ psq, int P,
                                                       double count, int rpp, int C, int per_split, double *__restrict__ ws)
{
    __shared__ double red[4][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int p0 = blockIdx.y * per_split, p1 = min(P, p0 + per_split);
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, nt = 0.0;
    const double inv_rpp = 1.0 / (double)rpp;
    if (c < C)
        for (int p = p0 + ty; p < p1; p += 32) {
            const double n = fmin((double)rpp, count - (double)p * rpp);
            if (n > 0.0) {
                const double sv = (double)psum[(size_t)p * C + c];
                s1 += sv; s2 += (double)psq[(size_t)p * C + c];
                s3 += sv * sv * (n == (double)rpp ? inv_rpp : 1.0 / n);
                nt += n;
            }
        }
    red[0][ty][tx] = s1; red[1][ty][tx] = s2; red[2][ty][tx] = s3; red[3][ty][tx] = nt;
    __syncthreads();
    for (int off = 16; off > 0; off >>= 1) {
        if (ty < off) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][ty][tx] += red[k][ty + off][tx];
        }
        __syncthreads();
    }
    if (ty == 0 && c < C) {
        double *o = ws + ((size_t)blockIdx.y * C + c) * 4;
        o[0] = red[0][0][tx]; o[1] = red[1][0][tx]; o[2] = red[2][0][tx]; o[3] = red[3][0][tx];
    }
}

__global__ __launch_bounds__(256) void bn_finish_kernel(const double *__restrict__ ws, int nsplit, double count, int C,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float eps, float momentum, float *__restrict__ rmean,
                                                        float *__restrict__ rvar, int64_t *__restrict__ nbt,
                                                        float *__restrict__ scale, float *__restrict__ shift,
        float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int k0 = 0; k0 < nsplit; k0 += 8) {             // eight independent loads in flight per step (a serial walk of
        double a[8][3];                                  // 32 dependent round trips made this tiny kernel take 11 us)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double *o = ws + ((size_t)min(k0 + u, nsplit - 1) * C + c) * 4;
            a[u][0] = o[0]; a[u][1] = o[1]; a[u][2] = o[2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nsplit) { s1 += a[u][0]; s2 += a[u][1]; s3 += a[u][2]; }
    }
    const double mean = s1 / count;
    const double var = fmax(s2 + s3 - s1 * s1 / count, 0.0) / count;      // biased variance normalises (nn.BatchNorm1d)
    const float mean_f = (float)mean, var_f = (float)var;
    if (rmean) {
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean_f;
        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unbiased;
        if (c == 0 && nbt) *nbt += 1;
    }
    const float sc = gamma[c] / sqrtf(var_f + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mean_f * sc;
    if (mean_out) { mean_out[c] = mean_f; rstd_out[c] = 1.0f / sqrtf(var_f + eps); }
}

// ---- BatchNorm over the rows of a row-major [M, C] fp32 tensor (the part-segmentation decoder: BatchNorm1d behind every
// 1x1 convolution, pointnet2_utils.py:297-368).  Threads run along the columns (coalesced rows), a workgroup owns a
// chunk of RS_ROWS rows: forward statistics by Welford's update (single pass, no cancellation) into the same
// (sum, M2) partial format bn_finalize folds; backward in two kernels: chunk partials of {sum g, sum g xhat} with
// g = dy * [y > 0] (the fused ReLU), then dx = scale * (g - mean(g) - xhat * mean(g xhat)).
constexpr int RS_ROWS = 64;

__global__ __launch_bounds__(256) void rows_stats_kernel(const float *__restrict__ x, int64_t M, int C,
                                                         float *__restrict__ psum, float *__restrict__ pm2)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * RS_ROWS;
    const int nrow = (int)min((int64_t)RS_ROWS, M - r0);
    float mean = 0.f, m2 = 0.f;
    for (int r = 0; r < nrow; ++r) {
        const float v = x[(r0 + r) * C + c];
        const float d = v - mean;
        mean += d * __builtin_amdgcn_rcpf((float)(r + 1));
        m2 = fmaf(d, v - mean, m2);
    }
    psum[(size_t)blockIdx.y * C + c] = mean * (float)nrow;
    pm2[(size_t)blockIdx.y * C + c] = m2;
}

__global__ __launch_bounds__(256) void bn_rows_bwd_reduce_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                 int relu, int64_t M, int C, float *__restrict__ pg,
                                                                 float *__restrict__ pgx)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * RS_ROWS;
    const int nrow = (int)min((int64_t)RS_ROWS, M - r0);
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    float sg = 0.f, sgx = 0.f;
    for (int r = 0; r < nrow; ++r) {
        const float v = x[(r0 + r) * C + c];
        float g = dy[(r0 + r) * C + c];
        if (relu && !(fmaf(v, sc, sh) > 0.f)) g = 0.f;
        sg += g;
        sgx = fmaf(g, (v - mu) * rs, sgx);
    }
    pg[(size_t)blockIdx.y * C + c] = sg;
    pgx[(size_t)blockIdx.y * C + c] = sgx;
}

// C % 4 == 0: the same partial sums with float4 loads -- a workgroup owns RS_ROWS rows x 64 columns as 16 row groups x 16 column
// quads (256-byte row pieces), rows of a group taken in ascending order, the 16 group sums folded through LDS in group order.
// (The one-thread-per-column walk above: 42 us for 32768 x 256, 1.6 TB/s; 7 per part-seg step.)
__global__ __launch_bounds__(256) void bn_rows_bwd_reduce_vec_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                     int relu, int64_t M, int C, float *__restrict__ pg,
                                                                     float *__restrict__ pgx)
{
    __shared__ float4 red[2][16][17];
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + q * 4;
    const int64_t r0 = (int64_t)blockIdx.y * RS_ROWS;
    const int nrow = (int)min((int64_t)RS_ROWS, M - r0);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgx = sg;
    if (c < C) {
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c), rs = *reinterpret_cast<const float4 *>(rstd + c);
#pragma unroll 4
        for (int r = g; r < nrow; r += 16) {
            const float4 v = *reinterpret_cast<const float4 *>(x + (r0 + r) * C + c);
            float4 gd = *reinterpret_cast<const float4 *>(dy + (r0 + r) * C + c);
            if (relu) {
                if (!(fmaf(v.x, sc.x, sh.x) > 0.f)) gd.x = 0.f;
                if (!(fmaf(v.y, sc.y, sh.y) > 0.f)) gd.y = 0.f;
                if (!(fmaf(v.z, sc.z, sh.z) > 0.f)) gd.z = 0.f;
                if (!(fmaf(v.w, sc.w, sh.w) > 0.f)) gd.w = 0.f;
            }
            sg.x += gd.x; sg.y += gd.y; sg.z += gd.z; sg.w += gd.w;
            sgx.x = fmaf(gd.x, (v.x - mu.x) * rs.x, sgx.x); sgx.y = fmaf(gd.y, (v.y - mu.y) * rs.y, sgx.y);
            sgx.z = fmaf(gd.z, (v.z - mu.z) * rs.z, sgx.z); sgx.w = fmaf(gd.w, (v.w - mu.w) * rs.w, sgx.w);
        }
    }
    red[0][g][q] = sg; red[1][g][q] = sgx;
    __syncthreads();
    if (g < 2 && c < C) {                                  // group 0 finishes sum g, group 1 finishes sum g x
        float4 t = red[g][0][q];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[g][k][q]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4 *>((g == 0 ? pg : pgx) + (size_t)blockIdx.y * C + c) = t;
    }
}

__global__ __launch_bounds__(256) void bn_rows_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ sum_g, const float *__restrict__ sum_gx,
                                                                int relu, int batch_stats, int64_t M, int C, float *__restrict__ dx,
                                                                uint16_t *__restrict__ dx_half, int half_dtype)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * RS_ROWS;
    const int nrow = (int)min((int64_t)RS_ROWS, M - r0);
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    const float mg = batch_stats ? sum_g[c] / (float)M : 0.f, mgx = batch_stats ? sum_gx[c] / (float)M : 0.f;
    for (int r = 0; r < nrow; ++r) {
        const float v = x[(r0 + r) * C + c];
        float g = dy[(r0 + r) * C + c];
        if (relu && !(fmaf(v, sc, sh) > 0.f)) g = 0.f;
        const float o = sc * (g - mg - (v - mu) * rs * mgx);
        if (dx) dx[(r0 + r) * C + c] = o;
        if (dx_half) dx_half[(r0 + r) * C + c] = from_f32_dt(half_dtype, o);
    }
}

template <typename TY>
__global__ __launch_bounds__(256) void linear3_gelu_kernel(const float *__restrict__ pts, int64_t M,
                                                           const float *__restrict__ w, const float *__restrict__ b,
                                                           int C, TY *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C) return;
    const int64_t m = i / C;
    const int c = (int)(i % C);
    const float v = fmaf(w[c * 3 + 2], pts[m * 3 + 2], fmaf(w[c * 3 + 1], pts[m * 3 + 1], fmaf(w[c * 3 + 0], pts[m * 3 + 0], b[c])));
    dt<TY>::store(y + i, 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)));
}

// ---- PointNet2 set abstraction -----------------------------------------------------------------
// y[b,s,k,:] = P[b, idx[b,s,k], :] + Q[b,s,:]; one workgroup = one 32-row chunk (threads == channels), which is
// also the chunk of the (sum, M2) BatchNorm partials.
template <typename TP, typename TY>
__global__ __launch_bounds__(256) void gather_add_kernel(const TP *__restrict__ P, const float *__restrict__ Q,
                                                         const int64_t *__restrict__ idx, int Nsrc, int S, int K, int C,
                                                         int64_t M, TY *__restrict__ y, float *__restrict__ psum,
                                                         float *__restrict__ pm2)
{
    __shared__ int src[32];
    __shared__ int ctr[32];
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    if (threadIdx.x < 32) {
        const int64_t m = r0 + threadIdx.x;
        if (m < M) {
            const int64_t bs = m / K;                      // (b, s)
            const int b = (int)(bs / S);
            src[threadIdx.x] = b * Nsrc + (int)min<int64_t>(idx[m], Nsrc - 1);
            ctr[threadIdx.x] = (int)bs;
        }
    }
    __syncthreads();
    const int nrow = (int)min<int64_t>(32, M - r0);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v[32];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            v[r] = 0.f;
            if (r < nrow) {
                v[r] = dt<TP>::load(P + (size_t)src[r] * C + c) + Q[(size_t)ctr[r] * C + c];
                s += v[r];
                dt<TY>::store(y + (size_t)(r0 + r) * C + c, v[r]);
            }
        }
        if (psum) {
            const float mean = s / (float)nrow;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) if (r < nrow) { const float d = v[r] - mean; q = fmaf(d, d, q); }
            psum[(size_t)blockIdx.x * C + c] = s;
            pm2[(size_t)blockIdx.x * C + c] = q;
        }
    }
}

template <typename TP, typename TO>
__global__ void pool_finish_kernel(const TP *__restrict__ pmax, const TP *__restrict__ pmin, int G, int fold, int C,
                                   const float *__restrict__ scale, const float *__restrict__ shift, TO *__restrict__ out,
                                   int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)G * C) return;
    const int64_t g = i / C;
    const int c = (int)(i % C);
    float mx = -INFINITY, mn = INFINITY;
    for (int f = 0; f < fold; ++f) {
        mx = fmaxf(mx, dt<TP>::load(pmax + (size_t)(g * fold + f) * C + c));
        mn = fminf(mn, dt<TP>::load(pmin + (size_t)(g * fold + f) * C + c));
    }
    const float sc = scale[c];
    dt<TO>::store(out + (size_t)g * ld + c, fmaxf(fmaf(sc, sc >= 0.f ? mx : mn, shift[c]), 0.f));   // max_k relu(bn(y_k))
}

template <typename TY>
__global__ void bn_act_rows_kernel(const float *__restrict__ x, int64_t n, int C, const float *__restrict__ scale,
                                   const float *__restrict__ shift, const float *__restrict__ mask, TY *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    float v = fmaxf(fmaf(x[i], scale[c], shift[c]), 0.f);
    if (mask) v *= mask[i];
    dt<TY>::store(y + i, v);
}

}  // namespace

extern "C" int ppt_gather_add(const void *P, int p_dtype, const float *Q, const int64_t *idx, int B, int Nsrc, int S, int K,
                              int C, void *y, int y_dtype, float *part_sum, float *part_m2, void *stream)
{
    if (!P || !Q || !idx || !y || B <= 0 || Nsrc <= 0 || S <= 0 || K <= 0 || C <= 0) return PPT_EINVAL;
    if ((part_sum == nullptr) != (part_m2 == nullptr)) return PPT_EINVAL;
    const int64_t M = (int64_t)B * S * K;
    dim3 grid((unsigned)((M + 31) / 32));
    hipStream_t s = ppt_stream(stream);
    if (p_dtype == PPT_F32 && y_dtype == PPT_F32)
        hipLaunchKernelGGL((gather_add_kernel<float, float>), grid, dim3(128), 0, s, (const float *)P, Q, idx, Nsrc, S, K, C, M, (float *)y, part_sum, part_m2);
    else if (p_dtype == PPT_F32 && y_dtype == PPT_BF16)
        hipLaunchKernelGGL((gather_add_kernel<float, bf16_t>), grid, dim3(128), 0, s, (const float *)P, Q, idx, Nsrc, S, K, C, M, (bf16_t *)y, part_sum, part_m2);
    else
        return PPT_EUNSUPPORTED;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_pool_finish(const void *pmax, const void *pmin, int p_dtype, int G, int fold, int C, const float *scale,
                               const float *shift, void *out, int out_dtype, int64_t ld_out, void *stream)
{
    if (!pmax || !pmin || !scale || !shift || !out || G <= 0 || fold <= 0 || C <= 0 || ld_out < C) return PPT_EINVAL;
    const int64_t n = (int64_t)G * C;
    dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t s = ppt_stream(stream);
#define PF(TP, TO) hipLaunchKernelGGL((pool_finish_kernel<TP, TO>), grid, dim3(256), 0, s, (const TP *)pmax, (const TP *)pmin, G, fold, C, scale, shift, (TO *)out, ld_out)
    if (p_dtype == PPT_F32 && out_dtype == PPT_F32) PF(float, float);
    else if (p_dtype == PPT_F32 && out_dtype == PPT_BF16) PF(float, bf16_t);
    else if (p_dtype == PPT_BF16 && out_dtype == PPT_BF16) PF(bf16_t, bf16_t);
    else if (p_dtype == PPT_BF16 && out_dtype == PPT_F32) PF(bf16_t, float);
    else return PPT_EINVAL;
#undef PF
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_bn_act_rows(const float *x, int M, int C, const float *scale, const float *shift, const float *mask,
                               void *y, int y_dtype, void *stream)
{
    if (!x || !scale || !shift || !y || M <= 0 || C <= 0) return PPT_EINVAL;
    const int64_t n = (int64_t)M * C;
    dim3 grid((unsigned)((n + 255) / 256));
    if (y_dtype == PPT_F32)
        hipLaunchKernelGGL(bn_act_rows_kernel<float>, grid, dim3(256), 0, ppt_stream(stream), x, n, C, scale, shift, mask, (float *)y);
    else if (y_dtype == PPT_BF16)
        hipLaunchKernelGGL(bn_act_rows_kernel<bf16_t>, grid, dim3(256), 0, ppt_stream(stream), x, n, C, scale, shift, mask, (bf16_t *)y);
    else if (y_dtype == PPT_F16)
        hipLaunchKernelGGL(bn_act_rows_kernel<f16_t>, grid, dim3(256), 0, ppt_stream(stream), x, n, C, scale, shift, mask, (f16_t *)y);
    else
        return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_conv1_stats_max_partials(int64_t M) { return (int)((M + STAT_ROWS - 1) / STAT_ROWS); }
extern "C" int ppt_conv1_stats_rows_per_partial(void) { return STAT_ROWS; }

extern "C" int ppt_conv1_stats(const float *pts, int64_t M, const float *w1, const float *b1, int C, float *part_sum,
                               float *part_sqsum, int *n_partials, void *stream)
{
    if (!pts || !w1 || !b1 || !part_sum || !part_sqsum || M <= 0 || C <= 0) return PPT_EINVAL;
    const int P = ppt_conv1_stats_max_partials(M);
    if (n_partials) *n_partials = P;
    hipLaunchKernelGGL(conv1_stats_kernel, dim3(P), dim3(256), 0, ppt_stream(stream), pts, M, w1, b1, C, part_sum,
                       part_sqsum);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

static int bn_finalize_single(const float *part_sum, const float *part_sqsum, int n_partials, int rows_per_partial,
                              int64_t count, int C,
                              const float *gamma, const float *beta, float eps, int train, float momentum,
                              float *running_mean, float *running_var, int64_t *num_batches_tracked, float *scale,
                              float *shift, float *mean_out, float *rstd_out, void *stream)
{
    if (!gamma || !beta || !scale || !shift || C <= 0) return PPT_EINVAL;
    if ((mean_out == nullptr) != (rstd_out == nullptr)) return PPT_EINVAL;
    if (train && (!part_sum || !part_sqsum || n_partials <= 0 || count <= 0 || rows_per_partial <= 0 ||
                  (int64_t)n_partials * rows_per_partial < count))
        return PPT_EINVAL;
    if (!train && (!running_mean || !running_var)) return PPT_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, ppt_stream(stream), part_sum, part_sqsum,
                       n_partials, (double)count, rows_per_partial, C, gamma, beta, eps, train, momentum, running_mean, running_var,
                       num_batches_tracked, scale, shift, mean_out, rstd_out);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_bn_finalize(const float *part_sum, const float *part_sqsum, int n_partials, int rows_per_partial,
                               int64_t count, int C,
                               const float *gamma, const float *beta, float eps, int train, float momentum,
                               float *running_mean, float *running_var, int64_t *num_batches_tracked, float *scale,
                               float *shift, void *stream)
{
    return bn_finalize_single(part_sum, part_sqsum, n_partials, rows_per_partial, count, C, gamma, beta, eps, train, momentum,
                              running_mean, running_var, num_batches_tracked, scale, shift, nullptr, nullptr, stream);
}

extern "C" size_t ppt_bn_finalize_workspace_bytes(int n_partials, int C)
{
    const int nsplit = n_partials >= 2048 ? 32 : 0;          // the single-kernel fold is as fast below that
    return (size_t)nsplit * (size_t)C * 4 * sizeof(double);
}

extern "C" int ppt_bn_finalize_ws(const float *part_sum, const float *part_sqsum, int n_partials, int rows_per_partial,
                                  int64_t count, int C,
                                  const float *gamma, const float *beta, float eps, int train, float momentum,
                                  float *running_mean, float *running_var, int64_t *num_batches_tracked, float *scale,
                                  float *shift, float *mean_out, float *rstd_out, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    const size_t need = train ? ppt_bn_finalize_workspace_bytes(n_partials, C) : 0;
    if (need == 0 || !workspace || workspace_bytes < need || ((uintptr_t)workspace & 7))
        return bn_finalize_single(part_sum, part_sqsum, n_partials, rows_per_partial, count, C, gamma, beta, eps, train, momentum,
                                  running_mean, running_var, num_batches_tracked, scale, shift, mean_out, rstd_out, stream);
    if ((mean_out == nullptr) != (rstd_out == nullptr)) return PPT_EINVAL;
    if (!gamma || !beta || !scale || !shift || C <= 0 || !part_sum || !part_sqsum || count <= 0 || rows_per_partial <= 0 ||
        (int64_t)n_partials * rows_per_partial < count)
        return PPT_EINVAL;
    const int nsplit = 32, per_split = (n_partials + nsplit - 1) / nsplit;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 31) / 32, nsplit), dim3(1024), 0, ppt_stream(stream), part_sum, part_sqsum,
                       n_partials, (double)count, rows_per_partial, C, per_split, (double *)workspace);
    PPT_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ppt_stream(stream), (const double *)workspace,
                       nsplit, (double)count, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                       scale, shift, mean_out, rstd_out);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_rows_stats_rows_per_partial(void) { return RS_ROWS; }

extern "C" int ppt_rows_stats_f32(const float *x, int64_t M, int C, float *part_sum, float *part_m2, void *stream)
{
    if (!x || !part_sum || !part_m2 || M <= 0 || C <= 0 || (M + RS_ROWS - 1) / RS_ROWS > 65535) return PPT_EINVAL;
    hipLaunchKernelGGL(rows_stats_kernel, dim3((C + 255) / 256, (unsigned)((M + RS_ROWS - 1) / RS_ROWS)), dim3(256), 0,
                       ppt_stream(stream), x, M, C, part_sum, part_m2);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_bn_rows_bwd_reduce(const float *dy, const float *x, const float *scale, const float *shift,
                                      const float *mean, const float *rstd, int relu, int64_t M, int C, float *part_g,
                                      float *part_gx, void *stream)
{
    if (!dy || !x || !scale || !shift || !mean || !rstd || !part_g || !part_gx || M <= 0 || C <= 0 ||
        (M + RS_ROWS - 1) / RS_ROWS > 65535)
        return PPT_EINVAL;
    if (C % 4 == 0 && !(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)rstd |
                          (uintptr_t)part_g | (uintptr_t)part_gx) & 15))
        hipLaunchKernelGGL(bn_rows_bwd_reduce_vec_kernel, dim3((C + 63) / 64, (unsigned)((M + RS_ROWS - 1) / RS_ROWS)), dim3(256), 0,
                           ppt_stream(stream), dy, x, scale, shift, mean, rstd, relu, M, C, part_g, part_gx);
    else
        hipLaunchKernelGGL(bn_rows_bwd_reduce_kernel, dim3((C + 255) / 256, (unsigned)((M + RS_ROWS - 1) / RS_ROWS)), dim3(256), 0,
                           ppt_stream(stream), dy, x, scale, shift, mean, rstd, relu, M, C, part_g, part_gx);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_bn_rows_bwd_apply(const float *dy, const float *x, const float *scale, const float *shift, const float *mean,
                                     const float *rstd, const float *sum_g, const float *sum_gx, int relu, int batch_stats,
                                     int64_t M, int C, float *dx, void *dx_half, int half_dtype, void *stream)
{
    if (!dy || !x || !scale || !shift || !mean || !rstd || (!dx && !dx_half) || M <= 0 || C <= 0 || (M + RS_ROWS - 1) / RS_ROWS > 65535)
        return PPT_EINVAL;
    if (dx_half && half_dtype != PPT_BF16 && half_dtype != PPT_F16) return PPT_EINVAL;
    if (batch_stats && (!sum_g || !sum_gx)) return PPT_EINVAL;
    hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3((C + 255) / 256, (unsigned)((M + RS_ROWS - 1) / RS_ROWS)), dim3(256), 0,
                       ppt_stream(stream), dy, x, scale, shift, mean, rstd, sum_g, sum_gx, relu, batch_stats, M, C, dx,
                       (uint16_t *)dx_half, half_dtype);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_linear3_gelu(const float *pts, int64_t M, const float *w, const float *b, int C, void *y, int y_dtype,
                                void *stream)
{
    if (!pts || !w || !b || !y || M <= 0 || C <= 0) return PPT_EINVAL;
    const int64_t n = M * C;
    dim3 grid((unsigned)((n + 255) / 256));
    if (y_dtype == PPT_BF16)
        hipLaunchKernelGGL(linear3_gelu_kernel<bf16_t>, grid, dim3(256), 0, ppt_stream(stream), pts, M, w, b, C, (bf16_t *)y);
    else if (y_dtype == PPT_F16)
        hipLaunchKernelGGL(linear3_gelu_kernel<f16_t>, grid, dim3(256), 0, ppt_stream(stream), pts, M, w, b, C, (f16_t *)y);
    else if (y_dtype == PPT_F32)
        hipLaunchKernelGGL(linear3_gelu_kernel<float>, grid, dim3(256), 0, ppt_stream(stream), pts, M, w, b, C, (float *)y);
    else
        return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
