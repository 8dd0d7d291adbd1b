// gemm_common.h -- what the tile GEMM kernels of gemm.hip and gemm256.hip share: the XOR-swizzled LDS image, the activation
// helpers, the A-prologue loaders of the register-staged kernel and the three epilogues (scalar, 16-byte LDS walk, register
// layout).  Internal linkage on purpose: every translation unit gets its own copy.
#pragma once
#include <stdlib.h>
#include "ppt_common.h"
#include "ppt_act.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int ROWB = 128, NT = 256;


__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <bool FAST>
__device__ __forceinline__ float act_fwd(float v, int act)
{
    switch (act) {
    case PPT_ACT_RELU: return fmaxf(v, 0.0f);
    case PPT_ACT_GELU: return FAST ? gelu_poly(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    case PPT_ACT_QUICKGELU: return v / (1.0f + __expf(-1.702f * v));
    default: return v;
    }
}
template <bool FAST>
__device__ __forceinline__ float act_bwd(float x, int act)   // d act(x) / dx
{
    switch (act) {
    case PPT_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
    case PPT_ACT_GELU: {
        const float cdf = 0.5f * (1.0f + (FAST ? erf_fast(x * 0.70710678118654752f) : erff(x * 0.70710678118654752f)));
        return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    }
    case PPT_ACT_QUICKGELU: {
        const float s = 1.0f / (1.0f + __expf(-1.702f * x));
        return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    default: return 1.0f;
    }
}

// the activation of N register values with the kind selected ONCE (a switch inside the per-element loop stays a
// branch per element)
template <bool FAST, int N>
__device__ __forceinline__ void act_fwd_n(float (&v)[N], int act)
{
    if (act == PPT_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = act_fwd<FAST>(v[e], PPT_ACT_GELU);
    } else if (act == PPT_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = fmaxf(v[e], 0.0f);
    } else if (act == PPT_ACT_QUICKGELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = act_fwd<FAST>(v[e], PPT_ACT_QUICKGELU);
    }
}
template <bool FAST, int N>
__device__ __forceinline__ void act_bwd_n(float (&v)[N], const float (&x)[N], int act)      // v *= act'(x)
{
    if (act == PPT_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] *= act_bwd<FAST>(x[e], PPT_ACT_GELU);
    } else if (act == PPT_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = x[e] > 0.0f ? v[e] : 0.0f;
    } else if (act == PPT_ACT_QUICKGELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] *= act_bwd<FAST>(x[e], PPT_ACT_QUICKGELU);
    }
}

template <typename T> __device__ __forceinline__ float load_as_f32(const void *p, int64_t i);
template <> __device__ __forceinline__ float load_as_f32<float>(const void *p, int64_t i) { return ((const float *)p)[i]; }
template <> __device__ __forceinline__ float load_as_f32<bf16_t>(const void *p, int64_t i) { return bf16_to_f32(((const bf16_t *)p)[i]); }
template <> __device__ __forceinline__ float load_as_f32<f16_t>(const void *p, int64_t i) { return f16_to_f32(((const uint16_t *)p)[i]); }
// element i of a tensor whose dtype is a run-time code
__device__ __forceinline__ float load_dt(const void *p, int dtype, int64_t i)
{
    return dtype == PPT_F32 ? ((const float *)p)[i] : to_f32_dt(dtype, ((const uint16_t *)p)[i]);
}

__device__ __forceinline__ void store_dt(void *p, int dtype, int64_t i, float v)
{
    if (dtype != PPT_F32) ((uint16_t *)p)[i] = from_f32_dt(dtype, v);
    else ((float *)p)[i] = v;
}

// ---- A / B slab loaders ---------------------------------------------------------------------
// thread t owns chunk column ch = t&7 of rows (t>>3) + 32*i, i < NR, in every slab.
template <int NR> struct Stage { uint4 v[NR]; };

template <typename T, int NR>
__device__ __forceinline__ void load_plain(Stage<NR> &st, const T *base, int64_t ld, int rows, int K, int r0, int k0)
{
    constexpr int EPC = 16 / sizeof(T);
    const int t = threadIdx.x, ch = t & 7;
    const int k = k0 + ch * EPC;
    const int kc = min(k, K - EPC);                 // always a valid address; out-of-range chunks are zeroed by mask_plain
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = r0 + (t >> 3) + 32 * i;
        st.v[i] = *reinterpret_cast<const uint4 *>(base + (int64_t)min(r, rows - 1) * ld + kc);
    }
}

// zero the chunks that lie outside [rows) x [K): done at LDS-write time, NOT at load time -- a select
// right behind the load would make the compiler wait for the data immediately (vmcnt(0)) and
// serialise the whole pipeline.
template <typename T, int NR>
__device__ __forceinline__ void mask_plain(Stage<NR> &st, int rows, int K, int r0, int k0)
{
    constexpr int EPC = 16 / sizeof(T);
    const int t = threadIdx.x, ch = t & 7;
    const bool kok = k0 + ch * EPC < K;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const bool ok = kok && (r0 + (t >> 3) + 32 * i) < rows;
        if (!ok) st.v[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

template <typename T> __device__ __forceinline__ void affine_relu_chunk(uint4 &v, const float *sc, const float *sh);
template <typename T>
__device__ __forceinline__ void affine_relu_chunk16(uint4 &v, const float *sc, const float *sh)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = fmaxf(fmaf(h16<T>::lo(w[e]), sc[2 * e], sh[2 * e]), 0.0f);
        const float hi = fmaxf(fmaf(h16<T>::hi(w[e]), sc[2 * e + 1], sh[2 * e + 1]), 0.0f);
        w[e] = h16<T>::pack2(lo, hi);
    }
    v = make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ __forceinline__ void affine_relu_chunk<bf16_t>(uint4 &v, const float *sc, const float *sh) { affine_relu_chunk16<bf16_t>(v, sc, sh); }
template <> __device__ __forceinline__ void affine_relu_chunk<f16_t>(uint4 &v, const float *sc, const float *sh) { affine_relu_chunk16<f16_t>(v, sc, sh); }
template <>
__device__ __forceinline__ void affine_relu_chunk<float>(uint4 &v, const float *sc, const float *sh)
{
    float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.0f);
    v = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}

// The A prologues split into the part that touches memory (issue early) and the part that only touches
// registers (run late, just before the LDS write), so that the transform does not wait for the loads.
template <typename T, int A_MODE, int NR>
__device__ __forceinline__ void load_A(Stage<NR> &st, const ppt_gemm_params &p, const T *A, int m0, int k0)
{
    if constexpr (A_MODE == PPT_A_CONV1) {          // stage the raw points (12 B / row); the conv happens in finish_A
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = min(m0 + (t >> 3) + 32 * i, p.M - 1);
            st.v[i] = make_uint4(__float_as_uint(p.pts[(int64_t)r * 3 + 0]), __float_as_uint(p.pts[(int64_t)r * 3 + 1]),
                                 __float_as_uint(p.pts[(int64_t)r * 3 + 2]), 0u);
        }
    } else {
        load_plain<T, NR>(st, A, p.lda, p.M, p.K, m0, k0);
    }
}

// Per-channel constants of the A prologues live in a small LDS table filled once per workgroup
// (AFFINE: {scale, shift} per k; CONV1: the BN-folded {s*wx, s*wy, s*wz, s*b + shift} per channel).  Fetching them
// from global memory inside finish_A put one exposed memory latency in front of every LDS write of every slab.
constexpr int PRO_TAB_K = 1024;                       // max K of a prologue GEMM (8 KiB of float2 / 16 KiB of float4 at K=1024)

template <int A_MODE>
__device__ __forceinline__ void fill_prologue_table(const ppt_gemm_params &p, float *tab)
{
    if constexpr (A_MODE == PPT_A_AFFINE_RELU) {
        for (int k = threadIdx.x; k < p.K; k += NT) { tab[2 * k] = p.a_scale[k]; tab[2 * k + 1] = p.a_shift[k]; }
    } else if constexpr (A_MODE == PPT_A_CONV1) {
        for (int c = threadIdx.x; c < p.K; c += NT) {
            const float s = p.a_scale ? p.a_scale[c] : 1.0f;
            const float h = p.a_shift ? p.a_shift[c] : 0.0f;
            tab[4 * c + 0] = s * p.w1[c * 3 + 0]; tab[4 * c + 1] = s * p.w1[c * 3 + 1]; tab[4 * c + 2] = s * p.w1[c * 3 + 2];
            tab[4 * c + 3] = fmaf(s, p.b1[c], h);
        }
    }
}

template <typename T, int A_MODE, int NR>
__device__ __forceinline__ void finish_A(Stage<NR> &st, const ppt_gemm_params &p, int m0, int k0, const float *tab)
{
    constexpr int EPC = 16 / sizeof(T);
    const int t = threadIdx.x, ch = t & 7;
    const int k = k0 + ch * EPC;
    if constexpr (A_MODE != PPT_A_CONV1) mask_plain<T, NR>(st, p.M, p.K, m0, k0);
    if constexpr (A_MODE == PPT_A_AFFINE_RELU) {
        if (k < p.K) {
            float sc[EPC], sh[EPC];
#pragma unroll
            for (int e = 0; e < EPC; e += 2) {
                const float4 v = *reinterpret_cast<const float4 *>(tab + 2 * (k + e));
                sc[e] = v.x; sh[e] = v.y; sc[e + 1] = v.z; sh[e + 1] = v.w;
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = m0 + (t >> 3) + 32 * i;
                if (r < p.M) affine_relu_chunk<T>(st.v[i], sc, sh);   // rows >= M stay zero
            }
        }
    } else if constexpr (A_MODE == PPT_A_CONV1) {   // a'[m][c] = relu(scale[c]*(w1[c].p_m + b1[c]) + shift[c]), c = k index
        float wx[EPC], wy[EPC], wz[EPC], wb[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = k + e;
            if (c < p.K) {
                const float4 v = *reinterpret_cast<const float4 *>(tab + 4 * c);
                wx[e] = v.x; wy[e] = v.y; wz[e] = v.z; wb[e] = v.w;
            } else { wx[e] = wy[e] = wz[e] = 0.f; wb[e] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = m0 + (t >> 3) + 32 * i;
            const float x = __uint_as_float(st.v[i].x), y = __uint_as_float(st.v[i].y), z = __uint_as_float(st.v[i].z);
            float f[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                f[e] = r < p.M ? fmaxf(fmaf(wz[e], z, fmaf(wy[e], y, fmaf(wx[e], x, wb[e]))), 0.0f) : 0.0f;
            if constexpr (sizeof(T) == 2)
                st.v[i] = make_uint4(h16<T>::pack2(f[0], f[1]), h16<T>::pack2(f[2], f[3]), h16<T>::pack2(f[4], f[5]),
                                     h16<T>::pack2(f[6], f[7]));
            else
                st.v[i] = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                                     __float_as_uint(f[3]));
        }
    }
}

template <int NR>
__device__ __forceinline__ void write_stage(const Stage<NR> &st, unsigned char *tile)
{
    const int t = threadIdx.x, ch = t & 7;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int row = (t >> 3) + 32 * i;
        *reinterpret_cast<uint4 *>(tile + lds_off(row, ch)) = st.v[i];
    }
}

// ---- one 128-byte K slab of MFMAs for this wave's (32*TI) x (32*TJ) ------------------------------
template <typename T, int TI, int TJ>
__device__ __forceinline__ void mma_slab(const unsigned char *As, const unsigned char *Bs, int arow0, int brow0, int lane,
                                         f32x16_t (&acc)[TI][TJ])
{
    const int r = lane & 31, h = lane >> 5;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                a[i] = *reinterpret_cast<const uint4 *>(As + lds_off(arow0 + i * 32 + r, kk * 2 + h));
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                b[j] = *reinterpret_cast<const uint4 *>(Bs + lds_off(brow0 + j * 32 + r, kk * 2 + h));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = h16<T>::mfma32(a[i], b[j], acc[i][j]);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int kq = 2 * kk + h;      // k index (in floats) inside the 32-float slab
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                a[i] = *reinterpret_cast<const float *>(As + lds_off(arow0 + i * 32 + r, kq >> 2) + (kq & 3) * 4);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                b[j] = *reinterpret_cast<const float *>(Bs + lds_off(brow0 + j * 32 + r, kq >> 2) + (kq & 3) * 4);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

// ---- split16: fp32 operands multiplied as hi + lo IEEE-half pairs on the 16-bit matrix pipe (ppt_gemm_params.split16) --------
// x * s = hi + lo with hi = half(x * s), lo = half(x * s - hi): 22 significand bits while |x * s| >= 2^-2, an absolute floor of
// 2^-25 below (half's subnormal step), inf above 65 504 -- s is the caller's power of two (split_a_pow2 / split_b_pow2).
// A . B = A_lo . B_hi + A_hi . B_lo + A_hi . B_hi (A_lo . B_lo, < 2^-22 relative, is dropped): three v_mfma_f32_32x32x16_f16
// (96 cycles) for the sixteen k that cost eight v_mfma_f32_32x32x2_f32 (512 cycles); fp32 accumulation as before.
// The split happens where the register-staged kernel moves a slab from registers to LDS (after the A prologue, after the
// out-of-range mask): a thread's 16-byte fp32 chunk (4 k) becomes 8 bytes of the tile's hi image and 8 of its lo image, so
// every value is split ONCE per workgroup (~3 VALU instructions) and the K loop reads ready 16-bit fragments.  A tile's
// two images take the bytes its fp32 image took: [ROWS][64 B] hi, then [ROWS][64 B] lo, 16-byte chunk c of row r at slot
// c ^ ((r >> 2) & 3) (four rows per 256-byte bank row: the ds_read_b128 fragment reads touch 16 distinct slots).
// (First version, measured and replaced: splitting at fragment-read time out of the fp32 LDS image -- every wave of a 2 x 2
// workgroup splits both operands again, 96 VALU instructions per 6 MFMAs: 1.3-1.5x the fp32 MFMA, tools/split16_bench.py.)
typedef _Float16 ppt_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int lds_off_s(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }

// Range (round 6): half(x * s) is inf beyond 65 504 and lo = half(x * s - inf) is NaN, where the fp32 MFMA this mode
// stands in for would have produced a finite product.  A FINITE value beyond half's range is therefore SATURATED to +-65 504
// before the split (lo = 0: the product is finite, and wrong by what was cut off) and counted in `over`; the kernel adds the
// workgroup's count to *p.split_overflow, which ppt_amd/health.py polls (BIT_SPLIT: the model leaves the split16 mode).  inf and
// NaN inputs are left alone: they propagate as they do in the fp32 mode.
constexpr float HALF_MAX = 65504.0f;
__device__ __forceinline__ float split_saturate(float x, uint32_t &over)
{
    const float ax = fabsf(x);
    const bool cut = ax > HALF_MAX && ax < __builtin_inff();       // (false for NaN and for inf)
    over |= (uint32_t)cut;
    return cut ? copysignf(HALF_MAX, x) : x;
}

template <int NR, int ROWS, int STRIDE = 32>                 // STRIDE = threads / 8: rows one pass of the workgroup covers
__device__ __forceinline__ void write_stage_split(const Stage<NR> &st, unsigned char *tile, float s, uint32_t &over)
{
    const int t = threadIdx.x, ch = t & 7;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int row = (t >> 3) + STRIDE * i;
        const float x[4] = {split_saturate(__uint_as_float(st.v[i].x) * s, over), split_saturate(__uint_as_float(st.v[i].y) * s, over),
                            split_saturate(__uint_as_float(st.v[i].z) * s, over), split_saturate(__uint_as_float(st.v[i].w) * s, over)};
        uint32_t H[2], L[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const _Float16 h0 = (_Float16)x[2 * q], h1 = (_Float16)x[2 * q + 1];
            const ppt_h2 hh = {h0, h1};
            const ppt_h2 ll = {(_Float16)(x[2 * q] - (float)h0), (_Float16)(x[2 * q + 1] - (float)h1)};
            H[q] = __builtin_bit_cast(uint32_t, hh);
            L[q] = __builtin_bit_cast(uint32_t, ll);
        }
        const int off = lds_off_s(row, ch >> 1) + (ch & 1) * 8;
        *reinterpret_cast<uint2 *>(tile + off) = make_uint2(H[0], H[1]);
        *reinterpret_cast<uint2 *>(tile + ROWS * 64 + off) = make_uint2(L[0], L[1]);
    }
}

// one atomic per wave that saturated anything (ppt_gemm_params.split_overflow may be NULL: saturation without a report)
__device__ __forceinline__ void split_report(uint32_t over, unsigned int *counter)
{
    if (counter && __builtin_amdgcn_ballot_w64(over != 0) != 0 && (threadIdx.x & 63) == 0) atomicAdd(counter, 1u);
}

template <int TI, int TJ, int AROWS, int BROWS>
__device__ __forceinline__ void mma_slab_split(const unsigned char *As, const unsigned char *Bs, int arow0, int brow0, int lane,
                                               f32x16_t (&acc)[TI][TJ])
{
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {                       // two 16-deep steps per 32-k slab
        uint4 ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const unsigned char *q = As + lds_off_s(arow0 + i * 32 + r, kk * 2 + h);
            ah[i] = *reinterpret_cast<const uint4 *>(q);
            al[i] = *reinterpret_cast<const uint4 *>(q + AROWS * 64);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const unsigned char *q = Bs + lds_off_s(brow0 + j * 32 + r, kk * 2 + h);
            bh[j] = *reinterpret_cast<const uint4 *>(q);
            bl[j] = *reinterpret_cast<const uint4 *>(q + BROWS * 64);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                acc[i][j] = h16<f16_t>::mfma32(al[i], bh[j], acc[i][j]);
                acc[i][j] = h16<f16_t>::mfma32(ah[i], bl[j], acc[i][j]);
                acc[i][j] = h16<f16_t>::mfma32(ah[i], bh[j], acc[i][j]);
            }
    }
}
template <int TI, int TJ>
__device__ __forceinline__ void scale_acc(f32x16_t (&acc)[TI][TJ], float inv)
{
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
}

// ---- scalar epilogue (any N / alignment): WN lanes span a row, 64/WN rows per pass ----------------
template <int WM, int WN>
__device__ __forceinline__ void epilogue_scalar(const ppt_gemm_params &p, float *ct, int lane, int mw, int nw, int m0, int wm,
                                                int64_t zc)
{
    constexpr int RP = 64 / WN;                     // rows per pass
    const int cl = lane % WN, rsub = lane / WN;
    const int n = nw + cl;
    const bool nok = n < p.N;
    const float bias = (p.bias && nok) ? p.bias[n] : 0.0f;
#pragma unroll 1
    for (int it = 0; it < WM / RP; ++it) {
        const int rr = it * RP + rsub;
        const int m = mw + rr;
        if (nok && m < p.M) {
            float v = ct[rr * WN + cl] + bias;
            if (p.group_add) v += p.group_add[(int64_t)(m / p.group_rows) * p.N + n];
            if (p.C2 && p.c2_pre) store_dt(p.C2, p.c2_dtype, (int64_t)m * p.ldc2 + n, v);
            if (p.dact_pre) {
                const float x = load_dt(p.dact_pre, p.dtype, (int64_t)m * p.ld_dact + n);
                v *= act_bwd<false>(x, p.act);
            } else {
                v = act_fwd<false>(v, p.act);
            }
            if (p.row_scale) v *= p.row_scale[m / p.row_scale_rows];
            if (p.residual) v += p.residual[(int64_t)m * p.ld_res + n];
            if (p.residual2) v += p.residual2[(int64_t)m * p.ld_res2 + n];
            if (p.C) store_dt(p.C, p.c_dtype, zc + (int64_t)m * p.ldc + n, v);
            if (p.C2 && !p.c2_pre) store_dt(p.C2, p.c2_dtype, (int64_t)m * p.ldc2 + n, v);
        }
    }
}

// ---- vector epilogue: 8 consecutive columns per lane, 64/(WN/8) rows per pass -------------------
// global stores are issue-bound, not byte-bound, on this chip (a 2-byte-per-lane store costs the same
// issue slot as a 16-byte one), so the tile is written as 16-byte pieces: lane = (row-in-pass, column
// group).  Column statistics / pooled maxima fold the row-lanes with three xor-shuffles.
struct f8 { float v[8]; };

__device__ __forceinline__ f8 ld8_f32(const float *p)
{
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    return f8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ void st8_f32(float *p, const f8 &x)
{
    *reinterpret_cast<float4 *>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
__device__ __forceinline__ f8 ld8_dt(const void *p, int dtype, int64_t i)
{
    if (dtype != PPT_F32) {
        const uint4 u = *reinterpret_cast<const uint4 *>((const uint16_t *)p + i);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        f8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { r.v[2 * e] = lo_dt(dtype, w[e]); r.v[2 * e + 1] = hi_dt(dtype, w[e]); }
        return r;
    }
    return ld8_f32((const float *)p + i);
}
__device__ __forceinline__ void st8_dt(void *p, int dtype, int64_t i, const f8 &x)
{
    if (dtype != PPT_F32)
        *reinterpret_cast<uint4 *>((uint16_t *)p + i) = make_uint4(pack2_dt(dtype, x.v[0], x.v[1]), pack2_dt(dtype, x.v[2], x.v[3]),
                                                                   pack2_dt(dtype, x.v[4], x.v[5]), pack2_dt(dtype, x.v[6], x.v[7]));
    else
        st8_f32((float *)p + i, x);
}

// FEAT bit 0: a prefetched global operand (group_add / residual / dact_pre) may be present;
// FEAT bit 1: column statistics / 32-row pooling may be requested.  Compile-time so that kernels that
// never use them (the register-heavy row-panel kernel) do not pay their registers.
template <int WM, int WN, int FEAT = 3>
__device__ __forceinline__ void epilogue_vec8(const ppt_gemm_params &p, float *ct, int lane, int mw, int nw, int64_t zc)
{
    constexpr int CGS = WN / 8, RP = 64 / CGS, NPASS = WM / RP;
    const int cg = lane % CGS, rl = lane / CGS;
    const int n = nw + cg * 8;
    const bool nok = n < p.N;
    const int nc = nok ? n : 0;                       // clamped column for the unconditional prefetches
    f8 bias, csum, pm, pn;
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias.v[e] = 0.f; csum.v[e] = 0.f; pm.v[e] = -INFINITY; pn.v[e] = INFINITY; }
    if (p.bias && nok) bias = ld8_f32(p.bias + n);

    // Phase A -- the ONE global operand of the epilogue (per-group term, residual or saved pre-activation)
    // is fetched for ALL passes up front: a load issued inside the pass loop is consumed immediately,
    // which exposes one full memory latency per pass (8 per tile; measured 2x on conv3 / proj / fc2).
    // The accumulators are parked in LDS by now, so their 64 registers are free to hold the 8 x 32 bytes.
    const int kind = (FEAT & 1) ? (p.group_add ? 1 : (p.residual ? 2 : (p.dact_pre ? 3 : 0))) : 0;
    uint4 pre[NPASS][2];
    if (kind != 0) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int m = min(mw + pass * RP + rl, p.M - 1);
            if (kind == 1) {
                const float *q = p.group_add + (int64_t)(m / p.group_rows) * p.N + nc;
                pre[pass][0] = *reinterpret_cast<const uint4 *>(q); pre[pass][1] = *reinterpret_cast<const uint4 *>(q + 4);
            } else if (kind == 2) {
                const float *q = p.residual + (int64_t)m * p.ld_res + nc;
                pre[pass][0] = *reinterpret_cast<const uint4 *>(q); pre[pass][1] = *reinterpret_cast<const uint4 *>(q + 4);
            } else if (p.dtype != PPT_F32) {
                pre[pass][0] = *reinterpret_cast<const uint4 *>((const uint16_t *)p.dact_pre + (int64_t)m * p.ld_dact + nc);
                pre[pass][1] = make_uint4(0, 0, 0, 0);
            } else {
                const float *q = (const float *)p.dact_pre + (int64_t)m * p.ld_dact + nc;
                pre[pass][0] = *reinterpret_cast<const uint4 *>(q); pre[pass][1] = *reinterpret_cast<const uint4 *>(q + 4);
            }
        }
    }
    constexpr bool PRE2 = NPASS <= 2;                      // (64x64 tiles; eight passes of it would spill)
    uint4 pre2[PRE2 ? NPASS : 1][2];                       // second fp32 residual (the next block's "+ pos")
    if (PRE2 && (FEAT & 1) && p.residual2) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const float *q = p.residual2 + (int64_t)min(mw + pass * RP + rl, p.M - 1) * p.ld_res2 + nc;
            pre2[pass][0] = *reinterpret_cast<const uint4 *>(q); pre2[pass][1] = *reinterpret_cast<const uint4 *>(q + 4);
        }
    }
    auto pre_f8 = [&](int pass, bool packed16) {
        f8 r;
        const uint4 a = pre[pass][0], b = pre[pass][1];
        if (packed16) {
            const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { r.v[2 * e] = lo_dt(p.dtype, w[e]); r.v[2 * e + 1] = hi_dt(p.dtype, w[e]); }
        } else {
            r = f8{{__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                    __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)}};
        }
        return r;
    };

#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int rr = pass * RP + rl;
        const int m = mw + rr;
        if (nok && m < p.M) {
            f8 v = ld8_f32(ct + rr * WN + cg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] += bias.v[e];
            if (kind == 1) {
                const f8 g = pre_f8(pass, false);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += g.v[e];
            }
            if ((FEAT & 2) && p.col_sum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) csum.v[e] += v.v[e];
                st8_f32(ct + rr * WN + cg * 8, v);
            }
            if (p.C2 && p.c2_pre) st8_dt(p.C2, p.c2_dtype, (int64_t)m * p.ldc2 + n, v);
            if ((FEAT & 1) && p.dact_pre) {
                const f8 x = kind == 3 ? pre_f8(pass, p.dtype != PPT_F32) : ld8_dt(p.dact_pre, p.dtype, (int64_t)m * p.ld_dact + n);
                if (p.dtype != PPT_F32) act_bwd_n<true, 8>(v.v, x.v, p.act);
                else act_bwd_n<false, 8>(v.v, x.v, p.act);
            } else if (p.act != PPT_ACT_NONE) {
                if (p.dtype != PPT_F32) act_fwd_n<true, 8>(v.v, p.act);
                else act_fwd_n<false, 8>(v.v, p.act);
            }
            if (p.row_scale) {
                const float sc = p.row_scale[m / p.row_scale_rows];
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] *= sc;
            }
            if ((FEAT & 1) && p.residual) {
                const f8 r = kind == 2 ? pre_f8(pass, false) : ld8_f32(p.residual + (int64_t)m * p.ld_res + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += r.v[e];
            }
            if ((FEAT & 1) && p.residual2) {
                f8 r;
                if constexpr (PRE2) {
                    const uint4 a = pre2[pass][0], b = pre2[pass][1];
                    r = f8{{__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                            __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)}};
                } else {
                    r = ld8_f32(p.residual2 + (int64_t)m * p.ld_res2 + n);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += r.v[e];
            }
            if (p.C) st8_dt(p.C, p.c_dtype, zc + (int64_t)m * p.ldc + n, v);
            if (p.C2 && !p.c2_pre) st8_dt(p.C2, p.c2_dtype, (int64_t)m * p.ldc2 + n, v);
            if (FEAT & 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { pm.v[e] = fmaxf(pm.v[e], v.v[e]); pn.v[e] = fminf(pn.v[e], v.v[e]); }
            }
        }
        if constexpr (WN == 64 && (FEAT & 2) != 0) {
            if (p.pool_max) {                             // groups of pool_rows (16 / 32 / 64) consecutive rows
                const int ppg = (p.pool_rows > 0 ? p.pool_rows : 32) / 8;     // passes per group
                if (((pass + 1) % ppg) == 0) {
                    const int mg = mw + (pass + 1 - ppg) * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = pm.v[e], u = pn.v[e];
                        t = fmaxf(t, __shfl_xor(t, 8, 64)); t = fmaxf(t, __shfl_xor(t, 16, 64)); t = fmaxf(t, __shfl_xor(t, 32, 64));
                        u = fminf(u, __shfl_xor(u, 8, 64)); u = fminf(u, __shfl_xor(u, 16, 64)); u = fminf(u, __shfl_xor(u, 32, 64));
                        pm.v[e] = t; pn.v[e] = u;
                    }
                    if (rl == 0 && nok && mg < p.M) {
                        const int64_t o = (int64_t)(mg / (ppg * 8)) * p.N + n;
                        st8_dt(p.pool_max, p.pool_dtype, o, pm);
                        if (p.pool_min) st8_dt(p.pool_min, p.pool_dtype, o, pn);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pm.v[e] = -INFINITY; pn.v[e] = INFINITY; }
                }
            }
            if ((pass & 3) == 3) {                        // rows [32*(pass>>2), +32) of the wave tile are complete
                const int mg = mw + (pass >> 2) * 32;
                if (p.col_sum && mg < p.M) {
                    // BatchNorm statistics of this 32-row chunk: (sum, M2 about the chunk mean) -- ppt_bn_finalize
                    // merges the chunks with the parallel-variance formula in fp64 (no cancellation, no atomics)
                    const int nrow = min(32, p.M - mg);
                    f8 csq;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = csum.v[e];
                        t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
                        csum.v[e] = t / (float)nrow; csq.v[e] = 0.f;          // chunk mean
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r2 = (pass - 3 + q) * 8 + rl;
                        if (nok && (r2 & 31) < nrow) {
                            const f8 v = ld8_f32(ct + r2 * 64 + cg * 8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float d = v.v[e] - csum.v[e]; csq.v[e] = fmaf(d, d, csq.v[e]); }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = csq.v[e];
                        t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
                        csq.v[e] = t; csum.v[e] *= (float)nrow;               // back to the chunk sum
                    }
                    if (rl == 0 && nok) {
                        st8_f32(p.col_sum + (int64_t)(mg >> 5) * p.N + n, csum);
                        st8_f32(p.col_sqsum + (int64_t)(mg >> 5) * p.N + n, csq);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) csum.v[e] = 0.f;
                }
            }
        }
    }
}

__host__ __device__ __forceinline__ bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// wave-uniform: every pointer / leading dimension the vector epilogue touches is 16-byte friendly
__device__ __forceinline__ bool vec_epilogue_ok(const ppt_gemm_params &p, int64_t zc)
{
    bool ok = (p.N % 8) == 0;
    if (p.C) ok = ok && (p.ldc % 8) == 0 && (zc % 8) == 0 && al16(p.C);
    if (p.C2) ok = ok && (p.ldc2 % 8) == 0 && al16(p.C2);
    if (p.bias) ok = ok && al16(p.bias);
    if (p.group_add) ok = ok && al16(p.group_add);
    if (p.dact_pre) ok = ok && (p.ld_dact % 8) == 0 && al16(p.dact_pre);
    if (p.residual) ok = ok && (p.ld_res % 8) == 0 && al16(p.residual);
    if (p.residual2) ok = ok && (p.ld_res2 % 8) == 0 && al16(p.residual2);
    if (p.col_sum) ok = ok && al16(p.col_sum) && al16(p.col_sqsum);
    if (p.pool_max) ok = ok && al16(p.pool_max);
    return ok;
}

// =================================================================================================
// Register-layout epilogue.  In the 32x32 MFMA C layout a lane owns ONE column (lane & 31) and 16 rows
// ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) of each tile, so everything that is per column or per row group
// -- bias, the per-group term, BatchNorm chunk statistics (a 32-row chunk IS one MFMA tile), max / min
// pooling over 16 / 32 / 64 rows -- is a register loop plus one v_permlane32_swap, with no LDS round trip
// (the LDS walk of epilogue_vec8 cost more than the K loop on the mini-PointNet GEMMs: conv3 575 us with,
// 208 us without its epilogue).  Only the C store needs row-major data: neighbouring lanes trade one value
// over DPP so that each holds two adjacent columns of one row, v_cvt_pk_bf16_f32 packs them, and the tile is
// parked as bf16 (half the LDS bytes and half the ds_write count of the fp32 park) for 16-byte row stores.
// Anything with a row-major fp32 operand (residual, saved pre-activation, second output) or an fp32 C stays
// on epilogue_vec8.
// =================================================================================================
template <int TI>
__host__ __device__ __forceinline__ bool reg_epilogue_ok(const ppt_gemm_params &p, int64_t zc)
{
    if (p.residual || p.residual2 || p.dact_pre || p.row_scale || p.C2) return false;
    if ((p.N % 8) != 0) return false;
    if (p.C && (p.c_dtype == PPT_F32 || (p.ldc % 8) != 0 || (zc % 8) != 0 || !al16(p.C))) return false;
    if (p.group_add && !(p.group_rows == 16 || (p.group_rows > 0 && (p.group_rows % 32) == 0))) return false;
    if (p.pool_max) {
        const int pr = p.pool_rows > 0 ? p.pool_rows : 32;
        if (!(pr == 16 || pr == 32 || (pr == 64 && TI == 2))) return false;
    }
    return true;
}

__device__ __forceinline__ void st_pool(void *base, int dtype, int64_t i, float v)
{
    if (dtype != PPT_F32) reinterpret_cast<uint16_t *>(base)[i] = from_f32_dt(dtype, v);
    else reinterpret_cast<float *>(base)[i] = v;
}

// per-column operands of epilogue_regs, fetched BEFORE the K loop (their addresses depend on the tile only), so that
// the epilogue never waits on memory: bias[j], and the per-group term of each MFMA tile (two for 16-row groups)
template <int TI, int TJ>
struct EpiPre { float bias[TJ]; float g[TI][TJ][2]; };

template <int TI, int TJ, bool GROUP = true>
__device__ __forceinline__ void epilogue_prefetch(const ppt_gemm_params &p, EpiPre<TI, TJ> &e, int lane, int mw, int nw)
{
    const int cl = lane & 31;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = nw + j * 32 + cl;
        const int nc = n < p.N ? n : 0;
        e.bias[j] = p.bias ? p.bias[nc] : 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int mg = mw + i * 32;
            e.g[i][j][0] = e.g[i][j][1] = 0.f;
            if (GROUP && p.group_add) {
                const int gr = p.group_rows > 0 ? p.group_rows : 32;
                e.g[i][j][0] = p.group_add[(int64_t)(min(mg, p.M - 1) / gr) * p.N + nc];
                e.g[i][j][1] = p.group_add[(int64_t)(min(mg + 16, p.M - 1) / gr) * p.N + nc];
            }
        }
    }
}

// EPI < 0: every feature tested at run time (the general kernel).  EPI >= 0: fixed at compile time -- 2 per-group
// term, 4 BatchNorm statistics, 8 pooling, activation kind << 4 -- so that the hot launches (qkv: 0, fc1: GELU, conv3: 6)
// run an epilogue of a few hundred instructions: with every feature compiled in and the four MFMA tiles unrolled the
// 128x128 kernels carry ~70 KB of epilogue code, more than the instruction cache two CUs share, and each enabled flag
// cost microseconds of instruction fetch (bias + ReLU on fc1: +10 us over the plain store).
constexpr int EPI_GROUP = 2, EPI_STATS = 4, EPI_POOL = 8, EPI_ACT_SHIFT = 4;      // + (activation kind << 4)
constexpr int EPI_GELU = PPT_ACT_GELU << EPI_ACT_SHIFT;
__host__ __device__ inline int epi_mask(const ppt_gemm_params &p)
{
    return (p.act << EPI_ACT_SHIFT) | (p.group_add ? EPI_GROUP : 0) | (p.col_sum ? EPI_STATS : 0) | (p.pool_max ? EPI_POOL : 0);
}

// FAST: -1 decided at run time from p.dtype, 1 bf16 operands (A&S erf), 0 fp32 parity (libm erff)
template <int TI, int TJ, int EPI = -1, int FAST = -1>
__device__ __forceinline__ void epilogue_regs(const ppt_gemm_params &p, f32x16_t (&acc)[TI][TJ], const EpiPre<TI, TJ> &pre,
                                              unsigned char *park, int lane, int mw, int nw, int64_t zc)
{
    const int act = EPI < 0 ? p.act : (EPI >> EPI_ACT_SHIFT);
    const bool has_act = act != PPT_ACT_NONE;
    const bool has_group = EPI < 0 ? p.group_add != nullptr : (EPI & EPI_GROUP) != 0;
    const bool has_stats = EPI < 0 ? p.col_sum != nullptr : (EPI & EPI_STATS) != 0;
    const bool has_pool = EPI < 0 ? p.pool_max != nullptr : (EPI & EPI_POOL) != 0;
    constexpr int WM = TI * 32, WN = TJ * 32, ROWBYTES = WN * 2;
    const int cl = lane & 31, h = lane >> 5, odd = cl & 1;
    const bool fast = FAST < 0 ? p.dtype != PPT_F32 : FAST != 0;
    const int pool_rows = p.pool_rows > 0 ? p.pool_rows : 32;
    // parked dword of this lane inside a (row pair, column tile): row + odd, columns (cl & ~1, cl | 1); for 64-column
    // wave tiles odd rows keep their two 64-byte halves swapped, which puts the even-lane row and the odd-lane row of
    // one ds_write_b32 on disjoint banks (and leaves the ds_read_b128 walk below conflict-free)
    const int lane_byte = (4 * h + odd) * ROWBYTES + (cl >> 1) * 4;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = nw + j * 32 + cl;
        const bool nok = n < p.N;
        const float bias = pre.bias[j];
        float pmx = -INFINITY, pmn = INFINITY;                       // carried over i for 64-row pools
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int mg = mw + i * 32;                               // first row of this MFMA tile (wave-uniform)
            const bool full = mg + 32 <= p.M;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bias;
            if (has_group) {                                          // (rows 16-31 of a 32k-row group: g[1] == g[0])
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += pre.g[i][j][r < 8 ? 0 : 1];
            }
            bool rv[16];                                              // row of register r exists
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = full || (mg + (r & 3) + 8 * (r >> 2) + 4 * h) < p.M;
            if (has_stats && mg < p.M) {
                // BatchNorm statistics of this 32-row chunk: (sum, M2 about the chunk mean); ppt_bn_finalize merges
                // the chunks with the parallel-variance formula in fp64 (no cancellation, no atomics)
                float sacc = 0.f, q = 0.f, mean;
                if (full) {                                           // (wave-uniform) no row masks on whole tiles
#ifdef PPT_DBG_PK
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) { s0 += v[r]; s1 += v[r + 1]; }
                    sacc = xor32_sum(s0 + s1);
                    mean = sacc * (1.0f / 32.0f);
                    float q0 = 0.f, q1 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float d0 = v[r] - mean, d1 = v[r + 1] - mean;
                        q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1);
                    }
                    q = xor32_sum(q0 + q1);
#else
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += v[r];
                    sacc = xor32_sum(sacc);
                    mean = sacc * (1.0f / 32.0f);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float d = v[r] - mean; q = fmaf(d, d, q); }
                    q = xor32_sum(q);
#endif
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += rv[r] ? v[r] : 0.f;
                    sacc = xor32_sum(sacc);
                    mean = sacc / (float)min(32, p.M - mg);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float d = v[r] - mean; q = rv[r] ? fmaf(d, d, q) : q; }
                    q = xor32_sum(q);
                }
                if (h == 0 && nok) {
                    p.col_sum[(int64_t)(mg >> 5) * p.N + n] = sacc;
                    p.col_sqsum[(int64_t)(mg >> 5) * p.N + n] = q;
                }
            }
            if (has_act) {
                if (fast) act_fwd_n<true, 16>(v, act);
                else act_fwd_n<false, 16>(v, act);
            }
            if (has_pool) {
                float a0 = -INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = INFINITY;   // rows 0-15 / 16-31 of the tile
                if (full) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { a0 = fmaxf(a0, v[r]); a1 = fmaxf(a1, v[r + 8]); }
                    if (p.pool_min) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) { b0 = fminf(b0, v[r]); b1 = fminf(b1, v[r + 8]); }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        a0 = fmaxf(a0, rv[r] ? v[r] : -INFINITY); b0 = fminf(b0, rv[r] ? v[r] : INFINITY);
                        a1 = fmaxf(a1, rv[r + 8] ? v[r + 8] : -INFINITY); b1 = fminf(b1, rv[r + 8] ? v[r + 8] : INFINITY);
                    }
                }
                if (pool_rows == 16) {
                    a0 = xor32_max(a0); a1 = xor32_max(a1);
                    if (p.pool_min) { b0 = xor32_min(b0); b1 = xor32_min(b1); }
                    if (h == 0 && nok) {
                        if (mg < p.M) { st_pool(p.pool_max, p.pool_dtype, (int64_t)(mg / 16) * p.N + n, a0);
                                        if (p.pool_min) st_pool(p.pool_min, p.pool_dtype, (int64_t)(mg / 16) * p.N + n, b0); }
                        if (mg + 16 < p.M) { st_pool(p.pool_max, p.pool_dtype, (int64_t)(mg / 16 + 1) * p.N + n, a1);
                                             if (p.pool_min) st_pool(p.pool_min, p.pool_dtype, (int64_t)(mg / 16 + 1) * p.N + n, b1); }
                    }
                } else {
                    pmx = fmaxf(pmx, fmaxf(a0, a1)); pmn = fminf(pmn, fminf(b0, b1));
                    if (pool_rows == 32 || i == TI - 1) {
                        const float tmx = xor32_max(pmx);
                        const int mg0 = pool_rows == 32 ? mg : mw;
                        if (h == 0 && nok && mg0 < p.M) st_pool(p.pool_max, p.pool_dtype, (int64_t)(mg0 / pool_rows) * p.N + n, tmx);
                        if (p.pool_min) {
                            const float tmn = xor32_min(pmn);
                            if (h == 0 && nok && mg0 < p.M) st_pool(p.pool_min, p.pool_dtype, (int64_t)(mg0 / pool_rows) * p.N + n, tmn);
                        }
                        pmx = -INFINITY; pmn = INFINITY;
                    }
                }
            }
            if (p.C) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // even lane keeps row(r): (own, neighbour's); odd lane keeps row(r+1): (neighbour's, own)
                    const float give = odd ? v[r] : v[r + 1];
                    const float got = __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(give)));   // quad_perm [1,0,3,2]
                    const uint32_t w = pack2_dt(p.c_dtype, odd ? got : v[r], odd ? v[r + 1] : got);
                    const int jb = TJ == 2 ? ((j ^ odd) << 6) : 0;     // the written row (r + odd) is odd exactly on odd lanes
                    *reinterpret_cast<uint32_t *>(park + (i * 32 + (r & 3) + 8 * (r >> 2)) * ROWBYTES + jb + lane_byte) = w;
                }
            }
        }
    }
    if (!p.C) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CPR = WN / 8, RP = 64 / CPR;                        // 16-byte chunks per row, rows per pass
    const int ch = lane % CPR, rl = lane / CPR;
    const int n = nw + ch * 8;
    uint16_t *C = reinterpret_cast<uint16_t *>(p.C) + zc;
#pragma unroll
    for (int pass = 0; pass < WM / RP; ++pass) {
        const int row = pass * RP + rl;
        const int m = mw + row;
        const uint4 d = *reinterpret_cast<const uint4 *>(park + row * ROWBYTES + ((ch * 16) ^ (TJ == 2 ? (row & 1) << 6 : 0)));
        if (m < p.M && n < p.N) *reinterpret_cast<uint4 *>(C + (int64_t)m * p.ldc + n) = d;
    }
}

}  // namespace
