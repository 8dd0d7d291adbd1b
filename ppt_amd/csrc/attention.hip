// attention.hip -- softmax(scale * q k^T [+causal]) v per (batch, head): forward and backward.
//
// Replaces point_encoder.py:46-55 (ViT attention, T=513, non-causal) and the nn.MultiheadAttention
// of ULIP_models.py:38,49-51 with the causal mask of :224-230 (L=77).  The [Bt,H,T,T] score tensor
// (202 MiB per layer at B=32 in the reference) is never materialised: scores live in registers and
// the running max / sum of the online softmax are per-row scalars.
//
// "quad" kernels (this file, both dtypes, fp32 math on the VALU): 4 lanes own one row, 16 of the
// 64 head dims each; dot products are 16 FMAs + a 2-step DPP quad reduction; the other operand is
// staged through LDS in 32-row tiles.  They are the parity-mode (PPT_F32) implementation and the
// general backward.  The bf16 MFMA flash forward lives in attention_mfma.hip and is dispatched from
// here for PPT_BF16.
#include "ppt_common.h"
#include "attn_rowmap.h"

// (fmt: PPT_BF16 or PPT_F16, the 16-bit operand format)
extern "C" int ppt_attention_fwd_mfma_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H,
                                           float scale, int causal, int P, int fmt, hipStream_t s);
extern "C" int ppt_attention_bwd_mfma_bf16(const void *qkv, const void *dout, const float *lse, const float *delta,
                                           void *dqkv, int Bt, int T, int H, float scale, int causal, int P, float *part,
                                           int fmt, hipStream_t s);

namespace {

constexpr int HD = 64, SEG = 16, KT = 32, ROWS = 64;   // rows per 256-thread block

__device__ __forceinline__ float quad_sum(float v)
{
    v += __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(v)));   // quad_perm [1,0,3,2]
    v += __uint_as_float(dpp_mov<0x4E, 0xf>(__float_as_uint(v)));   // quad_perm [2,3,0,1]
    return v;
}

// stage rows [r0, r0+KT) of one (b, h, which) slice into LDS as fp32 [KT][HD]; rows >= T -> 0
// (rows are positions of virtual sequence v: attn_rowmap.h)
template <typename T>
__device__ __forceinline__ void stage_rows(const T *__restrict__ base, int64_t row_stride, int r0, int Tlen,
                                           float *dst, int Tfull, int P, int v)
{
    for (int i = threadIdx.x; i < KT * HD; i += blockDim.x) {
        const int r = i >> 6, d = i & 63;
        dst[i] = (r0 + r) < Tlen ? dt<T>::load(base + am_row(Tfull, P, v, r0 + r) * row_stride + d) : 0.0f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_quad(const T *__restrict__ qkv, T *__restrict__ out,
                                                     float *__restrict__ lse, int Tfull, int H, float scale,
                                                     int causal, int P, int C)
{
    __shared__ __align__(16) float Ks[KT * HD];
    __shared__ __align__(16) float Vs[KT * HD];
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int seg = threadIdx.x & 3;
    const int qi = blockIdx.x * ROWS + (threadIdx.x >> 2);
    const int64_t rs = 3 * H * HD;
    const int Tlen = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);
    const T *qb = qkv + h * HD;
    const T *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const bool active = qi < Tlen;

    float q[SEG], o[SEG];
#pragma unroll
    for (int d = 0; d < SEG; ++d) {
        q[d] = active ? dt<T>::load(qb + am_row(Tfull, P, b, qi) * rs + seg * SEG + d) * scale : 0.f;
        o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    const int q_hi = min(Tlen, (int)(blockIdx.x + 1) * ROWS) - 1;
    const int nkt = causal ? q_hi / KT + 1 : (Tlen + KT - 1) / KT;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        stage_rows<T>(kb, rs, kt * KT, Tlen, Ks, Tfull, P, b);
        stage_rows<T>(vb, rs, kt * KT, Tlen, Vs, Tfull, P, b);
        __syncthreads();
        float s[KT];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < SEG; ++d) acc = fmaf(q[d], Ks[j * HD + seg * SEG + d], acc);
            acc = quad_sum(acc);
            const int kj = kt * KT + j;
            const bool ok = kj < Tlen && (!causal || kj <= qi);
            s[j] = ok ? acc : -INFINITY;
            tmax = fmaxf(tmax, s[j]);
        }
        const float mn = fmaxf(m, tmax);
        if (mn == -INFINITY) continue;            // inactive row (qi >= Tlen handled by stores) / fully masked
        const float alpha = __expf(m - mn);
        l *= alpha;
#pragma unroll
        for (int d = 0; d < SEG; ++d) o[d] *= alpha;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const float p = __expf(s[j] - mn);
            l += p;
#pragma unroll
            for (int d = 0; d < SEG; ++d) o[d] = fmaf(p, Vs[j * HD + seg * SEG + d], o[d]);
        }
        m = mn;
    }
    if (active && qi >= q_lo) {
        const float inv = 1.0f / l;
        T *ob = out + am_row(Tfull, P, b, qi) * (H * HD) + h * HD + seg * SEG;
#pragma unroll
        for (int d = 0; d < SEG; ++d) dt<T>::store(ob + d, o[d] * inv);
        if (lse && seg == 0) lse[am_stat(Tfull, P, H, b, h, qi)] = m + __logf(l);
    }
}

// delta[b,h,i] = sum_d out[i,d] * dout[i,d]
template <typename T>
__global__ __launch_bounds__(256) void attn_delta(const T *__restrict__ out, const T *__restrict__ dout,
                                                  float *__restrict__ delta, int Tlen, int H, int64_t total_rows, int P)
{
    const int64_t row = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);   // row over (b, t, h)
    const int seg = threadIdx.x & 3;
    float acc = 0.f;
    if (row < total_rows) {
        const T *o = out + row * HD + seg * SEG, *g = dout + row * HD + seg * SEG;
#pragma unroll
        for (int d = 0; d < SEG; ++d) acc = fmaf(dt<T>::load(o + d), dt<T>::load(g + d), acc);
    }
    acc = quad_sum(acc);
    if (row < total_rows && seg == 0 && P > 0) {
        delta[row] = acc;                                   // prefix-shared layout: statistics by physical row, [row * H + head]
    } else if (row < total_rows && seg == 0) {
        const int64_t bt = row / H;
        const int h = (int)(row % H);
        const int64_t b = bt / Tlen, t = bt % Tlen;
        delta[(b * H + h) * Tlen + t] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq(const T *__restrict__ qkv, const T *__restrict__ dout,
                                                   const float *__restrict__ lse, const float *__restrict__ delta,
                                                   T *__restrict__ dqkv, int Tfull, int H, float scale, int causal, int P, int C)
{
    __shared__ __align__(16) float Ks[KT * HD];
    __shared__ __align__(16) float Vs[KT * HD];
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int seg = threadIdx.x & 3;
    const int qi = blockIdx.x * ROWS + (threadIdx.x >> 2);
    const int64_t rs = 3 * H * HD;
    const int Tlen = am_len(Tfull, P, C, b);
    const T *qb = qkv + h * HD;
    const T *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const bool active = qi < Tlen && qi >= am_qlo(P, C, b);          // (rows this sequence does not own are the prefix sequence's)
    float q[SEG], g[SEG], dq[SEG];
#pragma unroll
    for (int d = 0; d < SEG; ++d) {
        q[d] = active ? dt<T>::load(qb + am_row(Tfull, P, b, qi) * rs + seg * SEG + d) : 0.f;
        g[d] = active ? dt<T>::load(dout + am_row(Tfull, P, b, qi) * (H * HD) + h * HD + seg * SEG + d) : 0.f;
        dq[d] = 0.f;
    }
    const float L = active ? lse[am_stat(Tfull, P, H, b, h, qi)] : 0.f;
    const float dl = active ? delta[am_stat(Tfull, P, H, b, h, qi)] : 0.f;
    const int q_hi = min(Tlen, (int)(blockIdx.x + 1) * ROWS) - 1;
    const int nkt = causal ? q_hi / KT + 1 : (Tlen + KT - 1) / KT;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        stage_rows<T>(kb, rs, kt * KT, Tlen, Ks, Tfull, P, b);
        stage_rows<T>(vb, rs, kt * KT, Tlen, Vs, Tfull, P, b);
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < KT; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < SEG; ++d) {
                s = fmaf(q[d], Ks[j * HD + seg * SEG + d], s);
                dp = fmaf(g[d], Vs[j * HD + seg * SEG + d], dp);
            }
            s = quad_sum(s) * scale;
            dp = quad_sum(dp);
            const int kj = kt * KT + j;
            const bool ok = active && kj < Tlen && (!causal || kj <= qi);
            const float p = ok ? __expf(s - L) : 0.f;
            const float ds = p * (dp - dl) * scale;
#pragma unroll
            for (int d = 0; d < SEG; ++d) dq[d] = fmaf(ds, Ks[j * HD + seg * SEG + d], dq[d]);
        }
    }
    if (active) {
        T *o = dqkv + am_row(Tfull, P, b, qi) * rs + h * HD + seg * SEG;
#pragma unroll
        for (int d = 0; d < SEG; ++d) dt<T>::store(o + d, dq[d]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv(const T *__restrict__ qkv, const T *__restrict__ dout,
                                                    const float *__restrict__ lse, const float *__restrict__ delta,
                                                    T *__restrict__ dqkv, int Tfull, int H, float scale, int causal, int P, int C,
                                                    float *__restrict__ part)
{
    __shared__ __align__(16) float Qs[KT * HD];
    __shared__ __align__(16) float Gs[KT * HD];
    __shared__ float Ls[KT], Ds[KT];
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int seg = threadIdx.x & 3;
    const int kj = blockIdx.x * ROWS + (threadIdx.x >> 2);
    const int64_t rs = 3 * H * HD;
    const int Tlen = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);
    const T *qb = qkv + h * HD;
    const T *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const T *gb = dout + h * HD;
    const bool active = kj < Tlen;
    float k[SEG], v[SEG], dk[SEG], dv[SEG];
#pragma unroll
    for (int d = 0; d < SEG; ++d) {
        k[d] = active ? dt<T>::load(kb + am_row(Tfull, P, b, kj) * rs + seg * SEG + d) : 0.f;
        v[d] = active ? dt<T>::load(vb + am_row(Tfull, P, b, kj) * rs + seg * SEG + d) : 0.f;
        dk[d] = 0.f; dv[d] = 0.f;
    }
    const int nqt = (Tlen + KT - 1) / KT;
    const int qt0 = causal ? (blockIdx.x * ROWS) / KT : 0;    // queries below the block's first key see none of it
    for (int qt = qt0; qt < nqt; ++qt) {
        __syncthreads();
        stage_rows<T>(qb, rs, qt * KT, Tlen, Qs, Tfull, P, b);
        stage_rows<T>(gb, (int64_t)H * HD, qt * KT, Tlen, Gs, Tfull, P, b);
        if (threadIdx.x < KT) {
            const int qi = qt * KT + threadIdx.x;
            Ls[threadIdx.x] = qi < Tlen ? lse[am_stat(Tfull, P, H, b, h, qi)] : 0.f;
            Ds[threadIdx.x] = qi < Tlen ? delta[am_stat(Tfull, P, H, b, h, qi)] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < KT; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < SEG; ++d) {
                s = fmaf(Qs[i * HD + seg * SEG + d], k[d], s);
                dp = fmaf(Gs[i * HD + seg * SEG + d], v[d], dp);
            }
            s = quad_sum(s) * scale;
            dp = quad_sum(dp);
            const int qi = qt * KT + i;
            const bool ok = active && qi < Tlen && qi >= q_lo && (!causal || kj <= qi);
            const float p = ok ? __expf(s - Ls[i]) : 0.f;
            const float ds = p * (dp - Ds[i]) * scale;
#pragma unroll
            for (int d = 0; d < SEG; ++d) {
                dv[d] = fmaf(p, Gs[i * HD + seg * SEG + d], dv[d]);
                dk[d] = fmaf(ds, Qs[i * HD + seg * SEG + d], dk[d]);
            }
        }
    }
    if (active && P > 0 && kj < P) {                       // a shared key: fp32 partial slot [b][kj][K | V][H * HD] (attn_prefix_reduce)
        float *pk = part + (((int64_t)b * P + kj) * 2) * (H * HD) + h * HD + seg * SEG, *pv = pk + H * HD;
#pragma unroll
        for (int d = 0; d < SEG; ++d) { pk[d] = dk[d]; pv[d] = dv[d]; }
    } else if (active) {
        T *o = dqkv + am_row(Tfull, P, b, kj) * rs + h * HD + seg * SEG;
#pragma unroll
        for (int d = 0; d < SEG; ++d) {
            dt<T>::store(o + H * HD + d, dk[d]);
            dt<T>::store(o + 2 * H * HD + d, dv[d]);
        }
    }
}

// dK / dV of the shared prefix rows = sum over the C + 1 virtual sequences of their partials, in sequence order (fixed:
// deterministic, no atomics); one thread per (position, K | V, column)
template <typename T>
__global__ __launch_bounds__(256) void attn_prefix_reduce(const float *__restrict__ part, int nseq, int P, int HHD, T *__restrict__ dqkv, int prio)
{
    PPT_PRIO(prio);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P * 2 * HHD) return;
    float acc = 0.f;
    const float *src = part + i;
    const int64_t stride = (int64_t)P * 2 * HHD;
    int v = 0;
    for (; v + 8 <= nseq; v += 8) {                      // eight loads in flight, added in sequence order
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(v + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; v < nseq; ++v) acc += src[v * stride];
    const int pos = i / (2 * HHD), rem = i - pos * 2 * HHD;
    dt<T>::store(dqkv + (int64_t)pos * 3 * HHD + HHD + rem, acc);
}

template <typename T>
int attn_fwd_t(const void *qkv, void *out, float *lse, int Bt, int Tl, int H, float scale, int causal, int P, hipStream_t s)
{
    dim3 grid((Tl + ROWS - 1) / ROWS, (Bt + (P > 0)) * H);
    hipLaunchKernelGGL(attn_fwd_quad<T>, grid, dim3(256), 0, s, (const T *)qkv, (T *)out, lse, Tl, H, scale, causal, P, Bt);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

__host__ int64_t attn_rows(int Bt, int Tl, int P) { return P > 0 ? (int64_t)P + (int64_t)Bt * (Tl - P) : (int64_t)Bt * Tl; }

template <typename T>
int attn_reduce_t(const float *part, void *dqkv, int Bt, int P, int H, hipStream_t s)
{
    if (P <= 0) return PPT_OK;
    const int n = P * 2 * H * HD;
    hipLaunchKernelGGL(attn_prefix_reduce<T>, dim3((n + 255) / 256), dim3(256), 0, s, part, Bt + 1, P, H * HD, (T *)dqkv, ppt_get_wave_priority());
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

template <typename T>
int attn_bwd_t(const void *qkv, const void *out, const void *dout, const float *lse, float *delta, void *dqkv,
               int Bt, int Tl, int H, float scale, int causal, int P, float *part, bool with_delta, hipStream_t s)
{
    if (with_delta) {
        const int64_t rows = attn_rows(Bt, Tl, P) * H;
        hipLaunchKernelGGL(attn_delta<T>, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, s, (const T *)out,
                           (const T *)dout, delta, Tl, H, rows, P);
        PPT_CHECK_LAUNCH();
    }
    dim3 grid((Tl + ROWS - 1) / ROWS, (Bt + (P > 0)) * H);
    hipLaunchKernelGGL(attn_bwd_dq<T>, grid, dim3(256), 0, s, (const T *)qkv, (const T *)dout, lse, delta, (T *)dqkv,
                       Tl, H, scale, causal, P, Bt);
    PPT_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv<T>, grid, dim3(256), 0, s, (const T *)qkv, (const T *)dout, lse, delta, (T *)dqkv,
                       Tl, H, scale, causal, P, Bt, part);
    PPT_CHECK_LAUNCH();
    return attn_reduce_t<T>(part, dqkv, Bt, P, H, s);
}

}  // namespace

static int attn_fwd_any(const void *qkv, void *out, float *lse, int Bt, int T, int P, int H, int hd, float scale, int causal,
                        int dtype, void *stream)
{
    if (!qkv || !out || Bt <= 0 || T <= 0 || H <= 0 || hd != HD) return PPT_EINVAL;
    if (dtype == PPT_F32) return attn_fwd_t<float>(qkv, out, lse, Bt, T, H, scale, causal, P, ppt_stream(stream));
    if (dtype == PPT_BF16 || dtype == PPT_F16) return ppt_attention_fwd_mfma_bf16(qkv, out, lse, Bt, T, H, scale, causal, P, dtype, ppt_stream(stream));
    return PPT_EINVAL;
}

extern "C" int ppt_attention_fwd(const void *qkv, void *out, float *lse, int Bt, int T, int H, int hd, float scale,
                                 int causal, int dtype, void *stream)
{
    return attn_fwd_any(qkv, out, lse, Bt, T, 0, H, hd, scale, causal, dtype, stream);
}

// used by attention_mfma.hip until every shape has an MFMA kernel
extern "C" int ppt_attention_bwd_short_mfma_bf16(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int Bt, int T,
                                                 int H, float scale, int causal, int P, float *part, int fmt, hipStream_t s);
extern "C" int ppt_attention_fwd_quad_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H, float scale,
                                           int causal, int P, int fmt, hipStream_t s)
{
    return fmt == PPT_F16 ? attn_fwd_t<f16_t>(qkv, out, lse, Bt, T, H, scale, causal, P, s) : attn_fwd_t<bf16_t>(qkv, out, lse, Bt, T, H, scale, causal, P, s);
}

static int attn_bwd_any(const void *qkv, const void *out, const void *dout, const float *lse, float *delta, void *dqkv, float *part,
                        int Bt, int T, int P, int H, int hd, float scale, int causal, int dtype, void *stream)
{
    if (!qkv || !out || !dout || !lse || !delta || !dqkv || Bt <= 0 || T <= 0 || H <= 0 || hd != HD) return PPT_EINVAL;
    if (P > 0 && !part) return PPT_EINVAL;
    if (dtype == PPT_F32)
        return attn_bwd_t<float>(qkv, out, dout, lse, delta, dqkv, Bt, T, H, scale, causal, P, part, true, ppt_stream(stream));
    if (dtype == PPT_BF16 || dtype == PPT_F16) {
        hipStream_t s = ppt_stream(stream);
        const bool f16 = dtype == PPT_F16;
        auto reduce = [&]() { return f16 ? attn_reduce_t<f16_t>(part, dqkv, Bt, P, H, s) : attn_reduce_t<bf16_t>(part, dqkv, Bt, P, H, s); };
        if (T <= 128) {                                    // short sequences (the text tower): delta + dK/dV + dQ in ONE launch
            const int rc = ppt_attention_bwd_short_mfma_bf16(qkv, out, dout, lse, dqkv, Bt, T, H, scale, causal, P, part, dtype, s);
            if (rc == PPT_OK) return reduce();
            if (rc != PPT_EUNSUPPORTED) return rc;
        }
        const int64_t rows = attn_rows(Bt, T, P) * H;
        if (f16)
            hipLaunchKernelGGL(attn_delta<f16_t>, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, s, (const f16_t *)out,
                               (const f16_t *)dout, delta, T, H, rows, P);
        else
            hipLaunchKernelGGL(attn_delta<bf16_t>, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, s, (const bf16_t *)out,
                               (const bf16_t *)dout, delta, T, H, rows, P);
        PPT_CHECK_LAUNCH();
        const int rc = ppt_attention_bwd_mfma_bf16(qkv, dout, lse, delta, dqkv, Bt, T, H, scale, causal, P, part, dtype, s);
        if (rc == PPT_OK) return reduce();
        if (rc != PPT_EUNSUPPORTED) return rc;
        return f16 ? attn_bwd_t<f16_t>(qkv, out, dout, lse, delta, dqkv, Bt, T, H, scale, causal, P, part, false, s)
                   : attn_bwd_t<bf16_t>(qkv, out, dout, lse, delta, dqkv, Bt, T, H, scale, causal, P, part, false, s);
    }
    return PPT_EINVAL;
}

extern "C" int ppt_attention_bwd(const void *qkv, const void *out, const void *dout, const float *lse, float *delta,
                                 void *dqkv, int Bt, int T, int H, int hd, float scale, int causal, int dtype,
                                 void *stream)
{
    return attn_bwd_any(qkv, out, dout, lse, delta, dqkv, nullptr, Bt, T, 0, H, hd, scale, causal, dtype, stream);
}

// ---- prefix-shared causal attention (attn_rowmap.h): C prompts of T positions whose first P positions are identical and
// stored once; qkv / out / dout / dqkv have P + C (T - P) rows; lse / delta [rows, H] f32; part [C + 1, P, 2, H * 64] f32.
extern "C" size_t ppt_attention_prefix_workspace_bytes(int C, int P, int H, int hd)
{
    return (size_t)(C + 1) * (size_t)P * 2 * (size_t)H * (size_t)hd * sizeof(float);
}

extern "C" int ppt_attention_prefix_fwd(const void *qkv, void *out, float *lse, int C, int T, int P, int H, int hd, float scale,
                                        int dtype, void *stream)
{
    if (P <= 0 || P >= T) return PPT_EINVAL;
    return attn_fwd_any(qkv, out, lse, C, T, P, H, hd, scale, 1, dtype, stream);
}

extern "C" int ppt_attention_prefix_bwd(const void *qkv, const void *out, const void *dout, const float *lse, float *delta,
                                        void *dqkv, float *part, int C, int T, int P, int H, int hd, float scale, int dtype,
                                        void *stream)
{
    if (P <= 0 || P >= T) return PPT_EINVAL;
    return attn_bwd_any(qkv, out, dout, lse, delta, dqkv, part, C, T, P, H, hd, scale, 1, dtype, stream);
}
