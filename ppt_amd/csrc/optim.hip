// optim.hip -- the prompt side's small serial steps as single kernels (every node of the prompt chain costs a dispatch
// round trip: text forward -> head -> text backward -> optimizer -> next text forward is ~250 dependent tiny kernels, and
// that chain, not the point tower, is half of what bounds the C2 step -- tools/critical_path.py):
//   * adamw_step: torch.optim.AdamW's update of one tensor (main_cls.py:58-60, 198) in ONE launch instead of the ~10
//     multi-tensor kernels of the foreach implementation; the same arithmetic in the same order:
//       p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = v b2 + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps);
//   * prompt_rows: PromptLearner.forward (ULIP_models.py:104-151) + the positional add of encode_text (:210) for the row
//     layout the text tower runs on: row i = base[i] (the frozen embedding + positional embedding, a constant of the
//     model) or, where slot[i] >= 0, learnable_tokens[slot[i]] + pos[i];
//   * prompt_rows_bwd: d learnable_tokens[t] = scale * sum over the rows i with slot[i] == t of g[i], rows taken in ascending
//     order (owner-computes: deterministic, no atomics) -- the backward of the splice; scale = 1 / the text tower's gradient
//     scale (ppt_amd/gradscale.py), a power of two;
//   * adamw_multi: the same AdamW update for up to PPT_ADAMW_MAX_TENSORS tensors in ONE launch (the tensor table travels as a
//     kernel argument) -- head_type >= 1 trains 4 ... 13 tensors, part segmentation 47.
// Given a skip counter (the mixed 16-bit mode), both AdamW kernels leave an element whose gradient is not finite alone (parameter and
// moments unchanged, gradient zeroed) and COUNT it in that device word, which the caller reads when it wants to (train.Trainer.nonfinite_grad_elements): the fp32 reference
// cannot overflow where a 16-bit backward stage can, and main_cls.py:205-207 stops on a non-finite loss.
#include "ppt_common.h"

namespace {

// one element of torch.optim.AdamW's single-tensor update; returns false (nothing written but g = 0) for a non-finite gradient
__device__ __forceinline__ bool adamw_element(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                              int64_t i, float decay, float omb1, float b2, float omb2, float inv_sqrt_bc2, float eps,
                                              float step_size, float grad_scale, bool guard)
{
    // grad_scale != 1: g holds the gradient of (loss / grad_scale) -- a caller that scaled its loss; the true gradient is
    // written back so that g reads like the reference's .grad afterwards (a power of two: exact)
    const float gi = g[i] * grad_scale;
    // A backward through fp16 operand stages can overflow where the fp32 reference would not: an element whose gradient is not
    // finite is left alone this step (parameter and moments keep their values, the gradient reads 0) instead of poisoning the
    // state for good, and is counted.
    // guard == false (no skip counter passed: the fp32 parity mode, where no stage can overflow that the reference's would not):
    // torch.optim.AdamW's behaviour -- a NaN gradient propagates into the parameter and main_cls.py:205-207 stops the run.
    if (guard && !(fabsf(gi) <= 3.0e38f)) { g[i] = 0.f; return false; }
    if (grad_scale != 1.f) g[i] = gi;
    float pi = p[i] * decay;
    const float mi = m[i] + (gi - m[i]) * omb1;
    const float vi = v[i] * b2 + omb2 * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    return true;
}

__global__ __launch_bounds__(256) void adamw_step_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v, int64_t n, float decay, float omb1, float b2,
                                                         float omb2, float inv_sqrt_bc2, float eps, float step_size, float grad_scale,
                                                         unsigned long long *__restrict__ skipped, int prio)
{
    PPT_PRIO(prio);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (!adamw_element(p, g, m, v, i, decay, omb1, b2, omb2, inv_sqrt_bc2, eps, step_size, grad_scale, skipped != nullptr)) atomicAdd(skipped, 1ull);
}

// the tensor table of ppt_adamw_multi as a kernel argument (48 B per tensor: 64 tensors = 3 KB of the 4 KB argument segment)
struct adamw_table {
    float *p[PPT_ADAMW_MAX_TENSORS], *g[PPT_ADAMW_MAX_TENSORS], *m[PPT_ADAMW_MAX_TENSORS], *v[PPT_ADAMW_MAX_TENSORS];
    int64_t n[PPT_ADAMW_MAX_TENSORS];
    float step_size[PPT_ADAMW_MAX_TENSORS], inv_sqrt_bc2[PPT_ADAMW_MAX_TENSORS];
    int first_block[PPT_ADAMW_MAX_TENSORS + 1];          // workgroups [first_block[t], first_block[t + 1]) walk tensor t, 1024 elements each
    int count;
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(const adamw_table tb, float decay, float omb1, float b2, float omb2, float eps,
                                                          float grad_scale, unsigned long long *__restrict__ skipped, int prio)
{
    PPT_PRIO(prio);
    int t = 0;                                            // (wave-uniform: blockIdx only -- scalar loads of the argument segment)
    while (t + 1 < tb.count && (int)blockIdx.x >= tb.first_block[t + 1]) ++t;
    const int64_t base = (int64_t)((int)blockIdx.x - tb.first_block[t]) * 1024;
    unsigned bad = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < tb.n[t] && !adamw_element(tb.p[t], tb.g[t], tb.m[t], tb.v[t], i, decay, omb1, b2, omb2, tb.inv_sqrt_bc2[t], eps,
                                          tb.step_size[t], grad_scale, skipped != nullptr))
            ++bad;
    }
    if (bad && skipped) atomicAdd(skipped, (unsigned long long)bad);
}

__global__ __launch_bounds__(256) void prompt_rows_kernel(const float *__restrict__ base, const int *__restrict__ slot,
                                                          const float *__restrict__ tokens, const float *__restrict__ pos_rows,
                                                          int rows, int W, float *__restrict__ out, int prio)
{
    PPT_PRIO(prio);
    const int w4 = W >> 2;
    const int64_t total = (int64_t)rows * w4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / w4), c = 4 * (int)(i - (int64_t)row * w4);
        const int s = slot[row];
        float4 v;
        if (s >= 0) {
            const float4 t = *reinterpret_cast<const float4 *>(tokens + (int64_t)s * W + c);
            const float4 q = *reinterpret_cast<const float4 *>(pos_rows + (int64_t)row * W + c);
            v = make_float4(t.x + q.x, t.y + q.y, t.z + q.z, t.w + q.w);
        } else {
            v = *reinterpret_cast<const float4 *>(base + (int64_t)row * W + c);
        }
        *reinterpret_cast<float4 *>(out + (int64_t)row * W + c) = v;
    }
}

// one thread per (token, 4 columns); the token's rows are listed (ascending) in rows_of[token * max_rows ...], -1 terminated
__global__ __launch_bounds__(256) void prompt_rows_bwd_kernel(const float *__restrict__ g, const int *__restrict__ rows_of, int max_rows,
                                                              int n_tok, int W, float *__restrict__ d_tokens, int prio, float scale)
{
    PPT_PRIO(prio);
    const int w4 = W >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_tok * w4) return;
    const int t = i / w4, c = 4 * (i - t * w4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight rows in flight, added in list order (one dependent index -> row walk per element was 40 round trips: 21.6 us)
    for (int k = 0; k < max_rows; k += 8) {
        int row[8];
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) row[u] = k + u < max_rows ? rows_of[t * max_rows + k + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = row[u] >= 0 ? *reinterpret_cast<const float4 *>(g + (int64_t)row[u] * W + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (row[u] >= 0) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        if (row[7] < 0) break;
    }
    *reinterpret_cast<float4 *>(d_tokens + (int64_t)t * W + c) = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
}

}  // namespace

extern "C" int ppt_adamw_step(float *p, float *g, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, uint64_t *skipped, void *stream)
{
    if (!p || !g || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0 || !(grad_scale > 0.f)) return PPT_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ppt_stream(stream), p, g, exp_avg, exp_avg_sq, n,
                       (float)(1.0 - (double)lr * weight_decay), 1.0f - beta1, beta2, 1.0f - beta2, (float)(1.0 / sqrt(bc2)), eps,
                       (float)((double)lr / bc1), grad_scale, (unsigned long long *)skipped, ppt_get_wave_priority());
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_adamw_multi(const ppt_adamw_tensor *tensors, int count, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float grad_scale, uint64_t *skipped, void *stream)
{
    if (!tensors || count <= 0 || !(grad_scale > 0.f)) return PPT_EINVAL;
    for (int i = 0; i < count; ++i)
        if (!tensors[i].p || !tensors[i].g || !tensors[i].exp_avg || !tensors[i].exp_avg_sq || tensors[i].n <= 0 || tensors[i].step <= 0)
            return PPT_EINVAL;
    for (int c0 = 0; c0 < count; c0 += PPT_ADAMW_MAX_TENSORS) {
        adamw_table tb;
        tb.count = count - c0 < PPT_ADAMW_MAX_TENSORS ? count - c0 : PPT_ADAMW_MAX_TENSORS;
        int64_t blocks = 0;
        for (int i = 0; i < tb.count; ++i) {
            const ppt_adamw_tensor &t = tensors[c0 + i];
            const double bc1 = 1.0 - pow((double)beta1, t.step), bc2 = 1.0 - pow((double)beta2, t.step);
            tb.p[i] = t.p; tb.g[i] = t.g; tb.m[i] = t.exp_avg; tb.v[i] = t.exp_avg_sq; tb.n[i] = t.n;
            tb.step_size[i] = (float)((double)lr / bc1);
            tb.inv_sqrt_bc2[i] = (float)(1.0 / sqrt(bc2));
            tb.first_block[i] = (int)blocks;
            blocks += (t.n + 1023) / 1024;
            if (blocks > 0x7fffffff) return PPT_EUNSUPPORTED;
        }
        tb.first_block[tb.count] = (int)blocks;
        hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, ppt_stream(stream), tb,
                           (float)(1.0 - (double)lr * weight_decay), 1.0f - beta1, beta2, 1.0f - beta2, eps, grad_scale,
                           (unsigned long long *)skipped, ppt_get_wave_priority());
        PPT_CHECK_LAUNCH();
    }
    return PPT_OK;
}

extern "C" int ppt_prompt_rows(const float *base, const int *slot, const float *tokens, const float *pos_rows, int rows, int W,
                               float *out, void *stream)
{
    if (!base || !slot || !tokens || !pos_rows || !out || rows <= 0 || W <= 0 || (W & 3)) return PPT_EINVAL;
    const int64_t total = (int64_t)rows * (W / 4);
    hipLaunchKernelGGL(prompt_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ppt_stream(stream), base, slot, tokens,
                       pos_rows, rows, W, out, ppt_get_wave_priority());
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_prompt_rows_bwd(const float *g, const int *rows_of, int max_rows, int n_tok, int W, float scale, float *d_tokens,
                                   void *stream)
{
    if (!g || !rows_of || !d_tokens || max_rows <= 0 || n_tok <= 0 || W <= 0 || (W & 3) || !(scale > 0.f)) return PPT_EINVAL;
    hipLaunchKernelGGL(prompt_rows_bwd_kernel, dim3((n_tok * (W / 4) + 255) / 256), dim3(256), 0, ppt_stream(stream), g, rows_of, max_rows,
                       n_tok, W, d_tokens, ppt_get_wave_priority(), scale);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
