// head.hip -- the step between the towers when only the prompt trains (head_type 0), as two small kernels:
//   ppt_head_logits : (two kernels) pc_embed = feat @ pc_projection (ULIP_models.py:257), text features L2-normalised (:279),
//                     logits = exp(logit_scale) * pc_embed @ text^T (:281)
//   ppt_head_ce_bwd : label-smoothed cross entropy, mean reduction (main_cls.py:52,196), and d loss / d text features
//                     (through the normalisation) -- all the backward needs.
// Replaces ~30 launches of tiny ATen / GEMM kernels on the critical path between two point towers (~120 us) by two
// three (~15 us).  fp32 throughout, fixed summation order (bit-reproducible).
#include "ppt_common.h"

namespace {

constexpr int HT = 256, CT = 1024;

template <int NT_>
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_reduce_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT_ / 64; ++i) t += red[i];
    return t;
}

// projection: workgroup = (sample b, 64 output columns), 4 waves each walk a quarter of F (coalesced 256-byte pieces of
// the rows of W[F,E] = pc_projection as stored, four independent chains), the four partial sums fold through LDS in a
// fixed order; feat row in LDS.  grid (E / 64, B).  (One thread per column over all of F was a 768-deep serial walk on 64
// workgroups: 84 us between the towers.)
__global__ __launch_bounds__(HT) void head_project_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                          const float *__restrict__ logit_scale, int F, int E,
                                                          float *__restrict__ spc)
{
    extern __shared__ float sm[];                         // [F] feat row, then [4][64] partials
    float *part = sm + F;
    const int b = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    for (int i = threadIdx.x; i < F; i += HT) sm[i] = feat[(size_t)b * F + i];
    __syncthreads();
    const int fq = (F + 3) / 4, f0 = q * fq, f1 = min(F, f0 + fq);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < E) {
        int i = f0;
        for (; i + 4 <= f1; i += 4) {                     // four independent chains: the loads stream, the FMAs do not wait
            a0 = fmaf(sm[i + 0], w[(size_t)(i + 0) * E + e], a0);
            a1 = fmaf(sm[i + 1], w[(size_t)(i + 1) * E + e], a1);
            a2 = fmaf(sm[i + 2], w[(size_t)(i + 2) * E + e], a2);
            a3 = fmaf(sm[i + 3], w[(size_t)(i + 3) * E + e], a3);
        }
        for (; i < f1; ++i) a0 = fmaf(sm[i], w[(size_t)i * E + e], a0);
    }
    part[q * 64 + lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && e < E)
        spc[(size_t)b * E + e] = __expf(logit_scale[0]) * ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]));
}

// The same projection for F <= 1536 with EIGHT samples per workgroup: grid (E / 64, ceil(B / 8)), 8 waves each walk an eighth of F
// with the lane on the output column; every W element loaded feeds eight FMAs (one per sample, feat rows in LDS), so W is read
// B / 8 times instead of B times and a wave's walk is F / 8 = 96 steps instead of 192: 24.7 -> ~4 us for 32 x 768 x 512 -- this
// kernel sits between two point towers AND at the head of the prompt chain.  Partials fold in wave order (fixed).
constexpr int HS = 8;
__global__ __launch_bounds__(512) void head_project8_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                            const float *__restrict__ logit_scale, int B, int F, int E,
                                                            float *__restrict__ spc, int prio, float alpha)
{
    PPT_PRIO(prio);
    extern __shared__ float sm[];                         // [HS][F] feat rows, then [8 waves][HS][64] partials
    float *part = sm + HS * F;
    const int b0 = blockIdx.y * HS, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    for (int i = threadIdx.x; i < HS * F; i += 512) {
        const int s_ = i / F;
        sm[i] = b0 + s_ < B ? feat[(size_t)(b0 + s_) * F + (i - s_ * F)] : 0.f;
    }
    __syncthreads();
    const int fq = (F + 7) / 8, f0 = q * fq, f1 = min(F, f0 + fq);
    float acc[HS];
#pragma unroll
    for (int s_ = 0; s_ < HS; ++s_) acc[s_] = 0.f;
    if (e < E) {
        int i = f0;
        // (round 5: SIXTEEN weight loads in flight per lane instead of four -- a wave's walk is 64-96 dependent-latency steps of a
        // 128-196 KB weight slice on 40 workgroups: 14-20 us in the step's traces, not the ~4 us of the stand-alone measurement;
        // same products in the same order per sample: the sums are the same bits)
        for (; i + 16 <= f1; i += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w[(size_t)(i + u) * E + e];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
#pragma unroll
                for (int s_ = 0; s_ < HS; ++s_) {
                    const float4 f = *reinterpret_cast<const float4 *>(sm + s_ * F + i + u);
                    acc[s_] = fmaf(f.w, wv[u + 3], fmaf(f.z, wv[u + 2], fmaf(f.y, wv[u + 1], fmaf(f.x, wv[u], acc[s_]))));
                }
            }
        }
        for (; i + 4 <= f1; i += 4) {
            const float w0 = w[(size_t)(i + 0) * E + e], w1 = w[(size_t)(i + 1) * E + e], w2 = w[(size_t)(i + 2) * E + e],
                        w3 = w[(size_t)(i + 3) * E + e];
#pragma unroll
            for (int s_ = 0; s_ < HS; ++s_) {
                const float4 f = *reinterpret_cast<const float4 *>(sm + s_ * F + i);      // (F % 4 == 0 and fq % 4 == 0 checked on the host)
                acc[s_] = fmaf(f.w, w3, fmaf(f.z, w2, fmaf(f.y, w1, fmaf(f.x, w0, acc[s_]))));
            }
        }
        for (; i < f1; ++i) {
            const float wv = w[(size_t)i * E + e];
#pragma unroll
            for (int s_ = 0; s_ < HS; ++s_) acc[s_] = fmaf(sm[s_ * F + i], wv, acc[s_]);
        }
    }
#pragma unroll
    for (int s_ = 0; s_ < HS; ++s_) part[(q * HS + s_) * 64 + lane] = acc[s_];
    __syncthreads();
    const int s_ = q;                                     // wave q finishes sample b0 + q
    if (e < E && b0 + s_ < B) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[(k * HS + s_) * 64 + lane];
        spc[(size_t)(b0 + s_) * E + e] = logit_scale ? __expf(logit_scale[0]) * t : alpha * t;      // (alpha: a power of two or 1 -- exact)
    }
}

// logits[b,c] = spc[b,:] . text[c,:] / |text[c,:]|: one wave per (b, c) pair, 16 waves per workgroup
__global__ __launch_bounds__(1024) void head_logits_kernel(const float *__restrict__ spc, const float *__restrict__ text, int B,
                                                           int E, int C, float *__restrict__ logits, int prio)
{
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (pair >= B * C) return;
    const int b = pair / C, c = pair % C;
    const float *pr = spc + (size_t)b * E, *tr = text + (size_t)c * E;
    float dot = 0.f, nn = 0.f;
    for (int i = lane; i < E; i += 64) { const float tv = tr[i]; dot = fmaf(pr[i], tv, dot); nn = fmaf(tv, tv, nn); }
    dot = wave_reduce_sum(dot); nn = wave_reduce_sum(nn);
    if (lane == 0) logits[pair] = dot / sqrtf(nn);
}

// one workgroup per class c: d_raw[c,:] = (d_tn - tn * (d_tn . tn)) / |text[c]|, d_tn = sum_b dlogits[b,c] * spc[b,:];
// dlogits = (softmax(logits) - ((1-eps) onehot + eps/C)) / B is recomputed per workgroup (B x C exps); workgroup 0 also
// writes the loss.
__global__ __launch_bounds__(CT) void head_ce_bwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels,
                                                         const float *__restrict__ spc, const float *__restrict__ text,
                                                         float smoothing, int B, int E, int C, float *__restrict__ loss,
                                                         float *__restrict__ d_raw, int prio)
{
    PPT_PRIO(prio);
    extern __shared__ float sm[];
    float *dl = sm;                                      // dlogits[:, c] for this class, [B]
    float *rowloss = sm + B;                             // [B] (workgroup 0)
    __shared__ float red[CT / 64];
    const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int b = w; b < B; b += CT / 64) {               // one wave per row: log-softmax of logits[b, :]
        const float *lr = logits + (size_t)b * C;
        float mx = -INFINITY;
        for (int i = lane; i < C; i += 64) mx = fmaxf(mx, lr[i]);
        mx = wave_reduce_max(mx);
        float se = 0.f, sl = 0.f;
        for (int i = lane; i < C; i += 64) { se += __expf(lr[i] - mx); sl += lr[i]; }
        se = wave_reduce_sum(se); sl = wave_reduce_sum(sl);
        const float lse = mx + __logf(se);
        const int y = (int)labels[b];
        if (lane == 0) {
            const float p = __expf(lr[c] - lse);
            const float tgt = smoothing / (float)C + (c == y ? 1.0f - smoothing : 0.0f);
            dl[b] = (p - tgt) / (float)B;
            // (1-eps) * nll + eps * mean_c(-log p_c)
            rowloss[b] = (1.0f - smoothing) * (lse - lr[y]) + smoothing * (lse - sl / (float)C);
        }
    }
    __syncthreads();
    if (c == 0) {
        float v = 0.f;
        for (int b = t; b < B; b += CT) v += rowloss[b];
        v = block_sum<CT>(v, red);
        if (t == 0) loss[0] = v / (float)B;
    }
    const float *tr = text + (size_t)c * E;
    float nn = 0.f;
    for (int i = t; i < E; i += CT) nn = fmaf(tr[i], tr[i], nn);
    nn = block_sum<CT>(nn, red);
    const float inv = 1.0f / sqrtf(nn);
    // d_tn[e] for this thread's columns, and its projection on tn
    float proj = 0.f;
    for (int e = t; e < E; e += CT) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(dl[b], spc[(size_t)b * E + e], acc);
        d_raw[(size_t)c * E + e] = acc;                  // (d_tn, finished below)
        proj = fmaf(acc, tr[e] * inv, proj);
    }
    proj = block_sum<CT>(proj, red);
    for (int e = t; e < E; e += CT) {
        const float dtn = d_raw[(size_t)c * E + e];
        d_raw[(size_t)c * E + e] = (dtn - tr[e] * inv * proj) * inv;
    }
}

}  // namespace

extern "C" int ppt_head_logits(const float *feat, const float *w, const float *text, const float *logit_scale, int B, int F,
                               int E, int C, float *spc, float *logits, void *stream)
{
    if (!feat || !w || !text || !logit_scale || !spc || !logits || B <= 0 || F <= 0 || E <= 0 || C <= 0) return PPT_EINVAL;
    if (F > 8192 || B > 65535) return PPT_EUNSUPPORTED;
    if (F <= 1536 && F % 32 == 0 && (((uintptr_t)feat) & 15) == 0)
        hipLaunchKernelGGL(head_project8_kernel, dim3((E + 63) / 64, (B + HS - 1) / HS), dim3(512), sizeof(float) * (size_t)(HS * F + 8 * HS * 64),
                           ppt_stream(stream), feat, w, logit_scale, B, F, E, spc, ppt_get_wave_priority(), 1.0f);
    else
        hipLaunchKernelGGL(head_project_kernel, dim3((E + 63) / 64, B), dim3(HT), sizeof(float) * (size_t)(F + 256), ppt_stream(stream),
                           feat, w, logit_scale, F, E, spc);
    PPT_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_logits_kernel, dim3((B * C + 15) / 16), dim3(1024), 0, ppt_stream(stream), spc, text, B, E, C, logits, ppt_get_wave_priority());
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// CrossEntropyLoss(label_smoothing = eps, mean reduction) over R rows of C classes (main_partseg.py:213: R = B x 2048 points,
// C = 50 parts; main_cls.py:52,196 when the point side trains) AND its gradient in one pass over the logits:
//   loss_r = (1 - eps) (lse_r - x_r[y_r]) + eps (lse_r - mean_c x_rc);  dlogits_rc = (softmax_rc - ((1 - eps) [c == y_r] + eps / C)) / R.
// 128 rows per workgroup go through LDS (coalesced both ways), a thread owns a row; per-workgroup loss partials are folded by
// ce_rows_finish in workgroup order (fixed).  Replaces log_softmax + nll_loss forward / backward + the smoothing ops.
constexpr int CE_ROWS = 128;
__global__ __launch_bounds__(CE_ROWS) void ce_rows_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels, float eps,
                                                          int64_t R, int C, int64_t ignore_index, float *__restrict__ dlogits,
                                                          float *__restrict__ partial)
{
    extern __shared__ float sm[];                         // [CE_ROWS][C + 1] (odd pitch for C even: no bank conflicts on the row walk)
    __shared__ float red[CE_ROWS], redc[CE_ROWS];
    const int pitch = C | 1;
    const int64_t r0 = (int64_t)blockIdx.x * CE_ROWS;
    const int nrow = (int)min((int64_t)CE_ROWS, R - r0);
    for (int i = threadIdx.x; i < nrow * C; i += CE_ROWS) sm[(i / C) * pitch + i % C] = logits[r0 * C + i];
    __syncthreads();
    float lr = 0.f, cnt = 0.f;
    if ((int)threadIdx.x < nrow) {
        float *x = sm + threadIdx.x * pitch;
        const int64_t y64 = labels[r0 + threadIdx.x];
        // label == ignore_index (nn.CrossEntropyLoss: -100 by default) is an IGNORED row, as in ATen: no loss, zero gradient, not
        // counted in the mean (ce_rows_finish divides by the number of counted rows).  Any OTHER label outside [0, C) is a
        // corrupt label: ATen raises a device assert; here the row's loss is NaN, so the mean is NaN and the caller's
        // non-finite-loss check (main_cls.py:205-207) fires instead of the row being dropped silently.
        const bool valid = y64 >= 0 && y64 < (int64_t)C;
        const bool corrupt = !valid && y64 != ignore_index;
        const int y = valid ? (int)y64 : 0;
        float m = x[0], sx = 0.f;
        for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
        float z = 0.f;
        for (int c = 0; c < C; ++c) { z += expf(x[c] - m); sx += x[c]; }
        const float lse = m + logf(z);
        lr = valid ? (1.0f - eps) * (lse - x[y]) + eps * (lse - sx / (float)C) : (corrupt ? __uint_as_float(0x7fc00000u) : 0.f);
        const float inv_r = valid ? 1.0f / (float)R : 0.f, base = eps / (float)C;
        for (int c = 0; c < C; ++c) x[c] = (expf(x[c] - lse) - (base + (c == y ? 1.0f - eps : 0.f))) * inv_r;
        cnt = (valid || corrupt) ? 1.f : 0.f;
    }
    red[threadIdx.x] = lr;
    redc[threadIdx.x] = cnt;
    __syncthreads();
    for (int i = threadIdx.x; i < nrow * C; i += CE_ROWS) dlogits[r0 * C + i] = sm[(i / C) * pitch + i % C];
    if (threadIdx.x == 0) {
        float t = 0.f, n = 0.f;
        for (int i = 0; i < CE_ROWS; ++i) { t += red[i]; n += redc[i]; }
        partial[blockIdx.x] = t;
        partial[gridDim.x + blockIdx.x] = n;
    }
}

// loss[0] = sum / counted rows; loss[1] = R / counted rows: dlogits was scaled by 1 / R, the caller multiplies the incoming
// gradient by loss[1] (exactly 1.0f when no row is ignored, so that case is bit for bit what it was)
__global__ __launch_bounds__(64) void ce_rows_finish(const float *__restrict__ partial, int n, float r, float *__restrict__ loss)
{
    if (threadIdx.x) return;
    double t = 0.0, cnt = 0.0;
    for (int i = 0; i < n; ++i) { t += (double)partial[i]; cnt += (double)partial[n + i]; }
    loss[0] = cnt > 0.0 ? (float)(t / cnt) : __uint_as_float(0x7fc00000u);      // (no counted row: NaN, as ATen)
    loss[1] = cnt > 0.0 ? (float)((double)r / cnt) : 0.f;
}

extern "C" int ppt_cross_entropy_rows(const float *logits, const int64_t *labels, float smoothing, int64_t R, int C, int64_t ignore_index,
                                      float *loss, float *dlogits, float *partial, void *stream)
{
    if (!logits || !labels || !loss || !dlogits || !partial || R <= 0 || C <= 0) return PPT_EINVAL;
    if (C > 96) return PPT_EUNSUPPORTED;                  // (rows stay in LDS: CE_ROWS x (C | 1) floats)
    const int nwg = (int)((R + CE_ROWS - 1) / CE_ROWS);
    hipLaunchKernelGGL(ce_rows_kernel, dim3(nwg), dim3(CE_ROWS), sizeof(float) * (size_t)CE_ROWS * (C | 1), ppt_stream(stream), logits, labels,
                       smoothing, R, C, ignore_index, dlogits, partial);
    PPT_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_rows_finish, dim3(1), dim3(64), 0, ppt_stream(stream), partial, nwg, (float)R, loss);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// out[M,N] = A[M,K] . W[K,N] in fp32 for a FEW rows (the text tower's EOT projection x @ text_projection, ULIP_models.py:222, and
// its backward: 40 rows): head_project8_kernel without the scale.  The fp32 MFMA tile loop ran these 10 MFLOP in 18.9 us (eight
// workgroups walking K serially); here ~4 us.
extern "C" int ppt_rows_matmul_f32(const float *A, const float *W, int M, int K, int N, float alpha, float *out, void *stream)
{
    if (!A || !W || !out || M <= 0 || K <= 0 || N <= 0 || !(alpha > 0.f)) return PPT_EINVAL;
    if (K > 1536 || (K % 32) != 0 || M > 65535 * HS || (((uintptr_t)A) & 15)) return PPT_EUNSUPPORTED;
    hipLaunchKernelGGL(head_project8_kernel, dim3((N + 63) / 64, (M + HS - 1) / HS), dim3(512), sizeof(float) * (size_t)(HS * K + 8 * HS * 64),
                       ppt_stream(stream), A, W, (const float *)nullptr, M, K, N, out, ppt_get_wave_priority(), alpha);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_head_ce_bwd(const float *logits, const int64_t *labels, const float *spc, const float *text, float smoothing,
                               int B, int E, int C, float *loss, float *d_text, void *stream)
{
    if (!logits || !labels || !spc || !text || !loss || !d_text || B <= 0 || E <= 0 || C <= 0) return PPT_EINVAL;
    if (B > 4096) return PPT_EUNSUPPORTED;
    hipLaunchKernelGGL(head_ce_bwd_kernel, dim3(C), dim3(CT), sizeof(float) * (size_t)(2 * B), ppt_stream(stream), logits, labels,
                       spc, text, smoothing, B, E, C, loss, d_text, ppt_get_wave_priority());
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
