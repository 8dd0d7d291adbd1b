// gemm.hip -- C[M,N] = epilogue( prologue(A)[M,K] . B[N,K]^T ) on the gfx950 matrix cores.
//
// One kernel family serves every nn.Linear / Conv1d(k=1) of the path (see include/ppt_hip.h):
//   * 128x128 (or, for grids that would under-fill the 256 CUs, 64x64) output tile per 256-thread
//     workgroup, 4 waves as 2x2, each wave (BM/2)x(BN/2) in MFMA tiles of 32x32
//     (v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 in parity mode), fp32 accumulators in registers;
//   * K is walked in 128-BYTE slabs (64 bf16 / 32 f32) so both dtypes share one LDS image:
//     [128 rows][8 x 16-B chunks], chunk index XOR-swizzled with (row>>1)&7, which makes the
//     ds_read_b128 fragment reads of the 32x32x16 operand conflict-free;
//   * register-staged pipeline, THREE slabs deep: while slab s is multiplied out of LDS, slabs s+1 and
//     s+2 sit in (or fly into) two register sets and the loads of slab s+3 are issued into the third;
//     slab s+1 moves to the other LDS buffer after the MFMAs -> one barrier per slab and ~2 slab-times
//     of latency cover for every global load (the K loop of these shapes is latency-, not
//     bandwidth-bound: M is huge or the grid is small, K is 128..2048).  Loads are branch-free
//     (clamped address + select), so the compiler's counted vmcnt keeps two slabs in flight.  Staging
//     through registers (rather than LDS-DMA) is what lets the A operand be TRANSFORMED on the way
//     in: BatchNorm+ReLU of the previous layer (A_AFFINE_RELU) or the whole K=3 first conv of the
//     mini-PointNet (A_CONV1) never touch HBM;
//   * epilogue on the accumulator registers: bias, per-group additive term, activation or its
//     derivative, DropPath row scale, up to two residual adds, a second output copy, BatchNorm
//     column statistics and the 32-row max-pool of the mini-PointNet (one MFMA row-tile == one
//     kNN group, so the pool is 15 v_max + one cross-half exchange).
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

template <int TI, int TJ>
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
            if (p.group_add) {
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

template <typename T, int A_MODE, int BM, int BN>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const ppt_gemm_params p)
{
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 32, TJ = WN / 32, NRA = BM / 32, NRB = BN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES), PARK_BYTES = 4 * WM * WN * 4;
    constexpr int MAIN_BYTES = STAGE_BYTES > PARK_BYTES ? STAGE_BYTES : PARK_BYTES;
    constexpr int TAB_BYTES = A_MODE == PPT_A_PLAIN ? 0 : (A_MODE == PPT_A_CONV1 ? 16 : 8) * PRO_TAB_K;
    __shared__ __align__(16) unsigned char smem[MAIN_BYTES + TAB_BYTES];   // A0 A1 B0 B1 | prologue table
    const float *tab = reinterpret_cast<const float *>(smem + MAIN_BYTES);
    if constexpr (A_MODE != PPT_A_PLAIN) {
        fill_prologue_table<A_MODE>(p, reinterpret_cast<float *>(smem + MAIN_BYTES));
        __syncthreads();
    }
    constexpr int BK = ROWB / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
#ifndef PPT_DBG_NO_XCD_SWIZZLE
    // consecutive tiles (same m-tile, n fastest) onto one XCD: blocks are dealt round-robin over the 8
    // XCDs, each with a private L2, so without this remap the column tiles that share one A row
    // panel land on 8 different L2s (measured +5..8 % on the block GEMMs; speed only, never correctness)
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    const int n0 = (lin % gridDim.x) * BN, m0 = (lin / gridDim.x) * BM;
#else
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
#endif
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)blockIdx.z * p.strideA;
    const T *B = reinterpret_cast<const T *>(p.B) + (int64_t)blockIdx.z * p.strideB;

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    EpiPre<TI, TJ> epre;        // (the CONV1 prologue already fills the register file: it fetches these after the K loop)
    if constexpr (A_MODE != PPT_A_CONV1) epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);

    const int nslab = (p.K + BK - 1) / BK;
    Stage<NRA> a0, a1, a2;
    Stage<NRB> b0, b1, b2;
    unsigned char *const Abuf = smem, *const Bbuf = smem + 2 * A_BYTES;

#ifdef PPT_DBG_SKIP_LDS_WRITE
#define PPT_DBG_WRITE(X) do { if (p.M < 0) { X; } } while (0)
#else
#define PPT_DBG_WRITE(X) X
#endif
#define PPT_LOAD(SA, SB, S)                                                   \
    do {                                                                      \
        load_A<T, A_MODE, NRA>(SA, p, A, m0, (S) * BK);                       \
        load_plain<T, NRB>(SB, B, p.ldb, p.N, p.K, n0, (S) * BK);             \
    } while (0)
#define PPT_WRITE(SA, SB, S, BUF)                                             \
    do {                                                                      \
        finish_A<T, A_MODE, NRA>(SA, p, m0, (S) * BK, tab);                   \
        mask_plain<T, NRB>(SB, p.N, p.K, n0, (S) * BK);                       \
        PPT_DBG_WRITE(write_stage<NRA>(SA, Abuf + ((BUF) & 1) * A_BYTES));    \
        PPT_DBG_WRITE(write_stage<NRB>(SB, Bbuf + ((BUF) & 1) * B_BYTES));    \
    } while (0)
    // STEP(s): multiply slab s; FREE set (held slab s) receives slab s+3; NEXT set (slab s+1) moves to LDS.
    // Loads and LDS writes are UNCONDITIONAL (slab indices clamped to the last slab; the surplus write
    // lands in the buffer nobody reads again): every path then has the same number of loads in flight
    // at every wait, which is what lets the compiler emit counted vmcnt(16) instead of draining to 0.
#define PPT_STEP(S, FA, FB, NA, NB)                                                                       \
    {                                                                                                     \
        PPT_LOAD(FA, FB, min((S) + 3, last));                                                             \
        mma_slab<T, TI, TJ>(Abuf + ((S) & 1) * A_BYTES, Bbuf + ((S) & 1) * B_BYTES, wm * WM, wn * WN, lane, acc); \
        PPT_WRITE(NA, NB, min((S) + 1, last), (S) + 1);                                                   \
        __syncthreads();                                                                                  \
    }

    const int last = nslab - 1;
    PPT_LOAD(a0, b0, 0);
    PPT_LOAD(a1, b1, min(1, last));
    PPT_LOAD(a2, b2, min(2, last));
    PPT_WRITE(a0, b0, 0, 0);
    __syncthreads();
#ifdef PPT_DBG_SKIP_LOADS
#undef PPT_LOAD
#define PPT_LOAD(SA, SB, S) do { } while (0)
#endif
    for (int s = 0;; s += 3) {
        PPT_STEP(s, a0, b0, a1, b1)
        if (s + 1 >= nslab) break;
        PPT_STEP(s + 1, a1, b1, a2, b2)
        if (s + 2 >= nslab) break;
        PPT_STEP(s + 2, a2, b2, a0, b0)
        if (s + 3 >= nslab) break;
    }
#undef PPT_STEP
#undef PPT_WRITE
#undef PPT_LOAD

    // ---------------- epilogue ----------------
    // The operand tiles are dead (the loop ends on a barrier): every wave parks its fp32 accumulators in
    // its own slice of LDS, row-major, and walks them in 16-byte pieces.  That turns the MFMA C layout
    // (column on the lane, rows scattered over 16 registers) into full-row global stores, makes the
    // BatchNorm column sums and the 32-row max-pool per-lane running values, and keeps the flag-driven
    // epilogue body out of the unroller.
    const int64_t zc = (int64_t)blockIdx.z * p.strideC;
#ifdef PPT_DBG_SKIP_EPILOGUE
    if (acc[0][0][0] != 12345.678f) return;
#endif
#ifndef PPT_DBG_NO_REG_EPILOGUE
    if (reg_epilogue_ok<TI>(p, zc)) {
        if constexpr (A_MODE == PPT_A_CONV1) epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);
        epilogue_regs<TI, TJ>(p, acc, epre, smem + w * (WM * WN * 2), lane, m0 + wm * WM, n0 + wn * WN, zc);
        return;
    }
#endif
    float *ct = reinterpret_cast<float *>(smem) + w * (WM * WN);
    {
        const int h = lane >> 5, cl = lane & 31;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * WN + j * 32 + cl] = acc[i][j][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if (vec_epilogue_ok(p, zc)) epilogue_vec8<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, zc);
    else epilogue_scalar<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, m0, wm, zc);
}

// =================================================================================================
// LDS-DMA variant for plain operands (no A prologue): global_load_lds_dwordx4 writes the slab straight
// into LDS, so the ~13-cycle-per-instruction ds_write_b128 path that register staging needs -- measured
// as 35-45 % of the K loop of the register-staged kernel (tools/gemm_tune.py, noloads_noepi vs
// noloads_noepi_nowrite) -- disappears, and no VGPRs are spent on staging.  Three LDS stages; slab s+2 is
// issued right behind the barrier that publishes slab s (its stage was last read two slabs ago); each
// wave waits for its own copies with a COUNTED vmcnt (the youngest slab stays in flight) before a raw
// s_barrier.  The LDS image is the same XOR-swizzled one: the DMA writes 64 lanes x 16 B linearly, so the
// swizzle is applied to each lane's SOURCE chunk (same involution as the fragment reads).
// Requires K % (128 / sizeof(T)) == 0; rows beyond M / N read a clamped (valid) row whose products are
// never stored.
// =================================================================================================
template <typename T, int ROWS>
__device__ __forceinline__ void glds_slab(const T *base, int64_t ld, int rows, int r0, int k0, unsigned char *tile, int w, int lane)
{
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PER_WAVE = ROWS / 32;                  // 1 KiB pieces (8 rows) per wave
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int rg = (w * PER_WAVE + i) * 8;           // first tile row of this piece
        const int r = rg + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);       // source chunk that belongs in LDS slot lane&7 of row r
        const T *src = base + (int64_t)min(r0 + r, rows - 1) * ld + k0 + c * EPC;
        lds_dma16(src, tile + rg * ROWB);
    }
}

// Diagnostic build only (tools/gemm_stamp.py compiles this file with -DPPT_GEMM_STAMP into its own library): lane 0 of every
// wave of the 64x64 LDS-DMA kernel stores s_memtime at entry / loads issued / first slab readable / K loop done / stores done
// into the buffer p.pool_min points at (unused by these launches).
#ifdef PPT_GEMM_STAMP
#define GEMM_STAMP(slot) do { if (lane == 0 && p.pool_min && !p.pool_max) reinterpret_cast<unsigned long long *>(p.pool_min)[((size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEMM_STAMP(slot) do { } while (0)
#endif
// NSTAGE: LDS stages of the ring = slabs in flight + 1.  Three (48 KiB: three workgroups of the 64 x 64 tile per CU).  FOUR
// (64 KiB, three slabs in flight) was measured for the prompt chain's small grids (<= 256 workgroups: one per CU whatever the
// LDS size) after the LDS-DMA ring actually ran as a ring (ppt_common.h lds_dma16): 6.7 -> 6.9 us at K = 512, 13.0 -> 13.4 us at
// K = 2048, C2 3.05 -> 3.10 ms -- the K loop is not short of bytes in flight, it runs at what ONE CU's LDS-DMA path takes
// (~27 B/clk).  PPT_GEMM_DEEP_BELOW=<workgroups> selects it (default 0: never).  (Also measured and removed: four HELPER waves per
// workgroup that only issue half of the DMA pieces and meet the barriers -- 12.6 -> 12.0 us at K = 2048 alone, C2 3.11 -> 3.19 ms in
// the step: the rate is a property of the CU, not of how many waves ask, and 512-thread workgroups are harder to place beside the
// tower.  And 32 x 64 tiles -- 208 two-wave workgroups, five 12 KiB stages, bit-identical results -- to put these 104-workgroup
// launches on more CUs: 12.2 -> 12.8 us plain, 12.6 -> 17.1 us with the residual epilogue, C2 3.08 -> 3.35 ms.  Two waves do not
// keep a CU's DMA path as busy as four.  The 64 x 64 / four-wave / three-stage kernel is where these shapes stay.)
template <typename T, int BM, int BN, int NSTAGE = 3>
__global__ __launch_bounds__(NT, BM == 64 ? 3 : 1) void gemm_kernel_glds(const ppt_gemm_params p)   // (a waves-per-SIMD floor keeps the accumulators out of AGPRs)
{
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 32, TJ = WN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int LOADS_PER_SLAB = BM / 32 + BN / 32;    // LDS-DMA instructions per wave per slab
    constexpr int PARK_BYTES = 4 * WM * WN * 4;
    __shared__ __align__(16) unsigned char smem[NSTAGE * STAGE > PARK_BYTES ? NSTAGE * STAGE : PARK_BYTES];
    constexpr int BK = ROWB / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    GEMM_STAMP(0);
    PPT_PRIO(p.wave_prio);
    const int wm = w >> 1, wn = w & 1;
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    const int n0 = (lin % gridDim.x) * BN, m0 = (lin / gridDim.x) * BM;
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)blockIdx.z * p.strideA;
    const T *B = reinterpret_cast<const T *>(p.B) + (int64_t)blockIdx.z * p.strideB;

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nslab = p.K / BK, last = nslab - 1;
    auto issue = [&](int slab, int stage) {
        glds_slab<T, BM>(A, p.lda, p.M, m0, slab * BK, smem + stage * STAGE, w, lane);
        glds_slab<T, BN>(B, p.ldb, p.N, n0, slab * BK, smem + stage * STAGE + A_BYTES, w, lane);
    };
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) issue(min(i, last), i);
    EpiPre<TI, TJ> epre;                                  // (behind the first slabs, as in gemm_kernel_glds_h)
    epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);
    GEMM_STAMP(1);
    int stage = 0;
    for (int s = 0; s < nslab; ++s) {
        // own copies of slab s have landed (the LOADS_PER_SLAB * (NSTAGE - 2) youngest, slabs s+1 .., may still fly) ...
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS_PER_SLAB * (NSTAGE - 2)) : "memory");
        __builtin_amdgcn_s_barrier();                     // ... and so have everybody else's: slab s is readable
        if (s == 0) GEMM_STAMP(2);
        int nstage = stage + NSTAGE - 1; if (nstage >= NSTAGE) nstage -= NSTAGE;
        issue(min(s + NSTAGE - 1, last), nstage);         // that stage was last read at slab s-1, before this barrier
        mma_slab<T, TI, TJ>(smem + stage * STAGE, smem + stage * STAGE + A_BYTES, wm * WM, wn * WN, lane, acc);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // surplus prefetches must not land on the parked accumulators
    __builtin_amdgcn_s_barrier();
    GEMM_STAMP(3);

    const int64_t zc = (int64_t)blockIdx.z * p.strideC;
#ifndef PPT_DBG_NO_REG_EPILOGUE
    if (reg_epilogue_ok<TI>(p, zc)) {
        epilogue_regs<TI, TJ>(p, acc, epre, smem + w * (WM * WN * 2), lane, m0 + wm * WM, n0 + wn * WN, zc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_STAMP(4);
        return;
    }
#endif
    float *ct = reinterpret_cast<float *>(smem) + w * (WM * WN);
    {
        const int h = lane >> 5, cl = lane & 31;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * WN + j * 32 + cl] = acc[i][j][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vec_epilogue_ok(p, zc)) epilogue_vec8<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, zc);
    else epilogue_scalar<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, m0, wm, zc);
#ifdef PPT_GEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_STAMP(4);
#endif
}

// (Tried and removed: a split-K form of this kernel for the prompt chain's skinny linears -- 817 rows, 104 ... 416 tiles --
// with blockIdx.z cutting K, fp32 partial tiles in a caller-lent workspace and the last workgroup to arrive at the tile's
// counter summing them in split order.  Correct and repeatable, never faster: with release / acquire fences every workgroup
// writes back and invalidates its XCD's whole L2 (52 ... 130 us per launch); with fence-free write-through stores and
// L2-bypassing loads (relaxed agent-scope atomics) the three dependent memory round trips -- partial out, counter, partials
// in -- cost the 4 ... 6 us the shorter K loop saves: 12.5 -> 12.5 us at two splits of K = 2048, slower beyond.  In-kernel
// stamps of the unsplit kernel (tools/gemm_stamp.py): ~1.2 us from entry to the first loads issued, 585 clocks per 16 KiB
// slab = 28 B/clk per CU, the L2 -> LDS rate of one CU.)
// =================================================================================================
// Half-slab LDS-DMA kernel for the big plain-operand problems (qkv, fc1, conv3): 128x128 tiles, 64-byte K slabs
// (32 bf16), 16 KiB stages: two of them (default; FOUR workgroups = 16 waves share a CU) or three (three workgroups).
// Why: tools/lds_fill_bench.hip shows that what a CU can pull out of L2 depends on how many waves are issuing --
// 4 waves 6.5, 8 waves 11, 12 waves 14, 16 waves 16 TB/s over the chip -- and hardly on the bytes each keeps in
// flight, and both existing tile loops sit on that line: 64x64 tiles (12 waves, 32 flop per LDS-fill byte) at
// 11.4 of 14 TB/s, register-staged 128x128 (8 waves, 64 flop/B) at 10 of 11 TB/s.  This kernel takes the 64 flop/B
// of the big tile AND the 12 waves.  It has only the register-layout epilogue (the fp32 park of epilogue_vec8 would
// need 64 KiB): the host sends it launches for which reg_epilogue_ok holds.
// LDS image: 64-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3): four consecutive rows fill one
// 256-byte bank row, and the lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) then touch 16 distinct slots.
// =================================================================================================
constexpr int ROWH = 64;
__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * ROWH + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <typename T, int ROWS>
__device__ __forceinline__ void glds_half(const T *base, int64_t ld, int rows, int r0, int k0, unsigned char *tile, int w, int lane)
{
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PER_WAVE = ROWS / 64;                  // 1 KiB pieces (16 rows) per wave
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int rg = (w * PER_WAVE + i) * 16;
        const int r = rg + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);       // source chunk that belongs in LDS slot lane&3 of row r
        const T *src = base + (int64_t)min(r0 + r, rows - 1) * ld + k0 + c * EPC;
        lds_dma16(src, tile + rg * ROWH);
    }
}

template <typename T, int TI, int TJ>
__device__ __forceinline__ void mma_half(const unsigned char *As, const unsigned char *Bs, int arow0, int brow0, int lane,
                                         f32x16_t (&acc)[TI][TJ])
{
    const int r = lane & 31, h = lane >> 5;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                a[i] = *reinterpret_cast<const uint4 *>(As + lds_off_h(arow0 + i * 32 + r, kk * 2 + h));
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                b[j] = *reinterpret_cast<const uint4 *>(Bs + lds_off_h(brow0 + j * 32 + r, kk * 2 + h));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = h16<T>::mfma32(a[i], b[j], acc[i][j]);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int kq = 2 * kk + h;      // k index (in floats) inside the 16-float slab
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                a[i] = *reinterpret_cast<const float *>(As + lds_off_h(arow0 + i * 32 + r, kq >> 2) + (kq & 3) * 4);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                b[j] = *reinterpret_cast<const float *>(Bs + lds_off_h(brow0 + j * 32 + r, kq >> 2) + (kq & 3) * 4);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

template <typename T, int BM, int BN, int NSTAGE = 3, int EPI = -1>
__global__ __launch_bounds__(NT, NSTAGE == 3 ? 3 : 4) void gemm_kernel_glds_h(const ppt_gemm_params p)
{
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 32, TJ = WN / 32;
    constexpr int A_BYTES = BM * ROWH, B_BYTES = BN * ROWH, STAGE = A_BYTES + B_BYTES;
    constexpr int LOADS_PER_SLAB = BM / 64 + BN / 64;    // LDS-DMA instructions per wave per slab
    static_assert(LOADS_PER_SLAB == 4, "counted vmcnt below");
    constexpr int PARK_BYTES = 4 * WM * WN * 2;          // bf16 park of epilogue_regs
    __shared__ __align__(16) unsigned char smem[NSTAGE * STAGE > PARK_BYTES ? NSTAGE * STAGE : PARK_BYTES];
    constexpr int BK = ROWH / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;                   // same XCD-aware remap as gemm_kernel
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    const int n0 = (lin % gridDim.x) * BN, m0 = (lin / gridDim.x) * BM;
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)blockIdx.z * p.strideA;
    const T *B = reinterpret_cast<const T *>(p.B) + (int64_t)blockIdx.z * p.strideB;

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nslab = p.K / BK, last = nslab - 1;
    auto issue = [&](int slab, int stage) {
        glds_half<T, BM>(A, p.lda, p.M, m0, slab * BK, smem + stage * STAGE, w, lane);
        glds_half<T, BN>(B, p.ldb, p.N, n0, slab * BK, smem + stage * STAGE + A_BYTES, w, lane);
    };
    issue(0, 0);
    if constexpr (NSTAGE == 3) issue(min(1, last), 1);
    EpiPre<TI, TJ> epre;                                  // (behind the first slabs: the K loop must not start later for it)
    epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);
    int stage = 0;
    for (int s = 0; s < nslab; ++s) {
        if constexpr (NSTAGE == 3) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // own copies of slab s landed (slab s+1's four may still fly)
            __builtin_amdgcn_s_barrier();                     // ... and everybody else's: slab s is readable
            int nstage = stage + 2; if (nstage >= NSTAGE) nstage -= NSTAGE;
            issue(min(s + 2, last), nstage);                  // stage (s+2)%3 was last read at slab s-1, before this barrier
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(min(s + 1, last), stage ^ 1);               // the other stage was last read at slab s-1, before this barrier
        }
        mma_half<T, TI, TJ>(smem + stage * STAGE, smem + stage * STAGE + A_BYTES, wm * WM, wn * WN, lane, acc);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // surplus prefetches must not land on the parked tile
    __builtin_amdgcn_s_barrier();
    epilogue_regs<TI, TJ, EPI, EPI < 0 ? -1 : (sizeof(T) == 2 ? 1 : 0)>(p, acc, epre, smem + w * (WM * WN * 2), lane, m0 + wm * WM,
                                                                        n0 + wn * WN, (int64_t)blockIdx.z * p.strideC);
}

// host side of the choice: plain operands, K in whole half-slabs, an epilogue the register path covers for every
// batch slice, and enough 128x128 tiles to give each CU its three workgroups
template <typename T>
bool glds_h_ok(const ppt_gemm_params &p, int64_t tiles128)
{
    static const int min_tiles = [] { const char *e = getenv("PPT_GEMM_H128_MIN"); return e ? atoi(e) : 768; }();
    constexpr int BKH = ROWH / sizeof(T);
    if (p.a_mode != PPT_A_PLAIN || (p.K % BKH) != 0 || tiles128 < min_tiles) return false;
    if (p.batch > 1 && (p.strideC % 8) != 0) return false;
    return reg_epilogue_ok<2>(p, 0);
}

template <typename T, int BM, int BN>
int launch_gemm_tile(const ppt_gemm_params &p, hipStream_t s)
{
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.batch > 0 ? p.batch : 1);
    if (grid.y > 65535 || grid.z > 65535) return PPT_EUNSUPPORTED;
    static const int use_glds = [] { const char *e = getenv("PPT_GEMM_GLDS"); return e ? atoi(e) : 1; }();
    constexpr int BKE = ROWB / sizeof(T);
    // (three 32 KiB stages of the 128x128 tile leave one workgroup per CU: measured slower than register staging)
    if (use_glds && BM == 64 && p.a_mode == PPT_A_PLAIN && p.K % BKE == 0) {
        static const int deep_below = [] { const char *e = getenv("PPT_GEMM_DEEP_BELOW"); return e ? atoi(e) : 0; }();
        if constexpr (BM == 64 && BN == 64) {
            if ((int)(grid.x * grid.y * grid.z) < deep_below && p.K / BKE >= 4) {
                hipLaunchKernelGGL((gemm_kernel_glds<T, 64, 64, 4>), grid, dim3(NT), 0, s, p);
                PPT_CHECK_LAUNCH();
                return PPT_OK;
            }
        }
        hipLaunchKernelGGL((gemm_kernel_glds<T, BM, BN>), grid, dim3(NT), 0, s, p);
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    switch (p.a_mode) {
    case PPT_A_PLAIN: hipLaunchKernelGGL((gemm_kernel<T, PPT_A_PLAIN, BM, BN>), grid, dim3(NT), 0, s, p); break;
    case PPT_A_AFFINE_RELU: hipLaunchKernelGGL((gemm_kernel<T, PPT_A_AFFINE_RELU, BM, BN>), grid, dim3(NT), 0, s, p); break;
    case PPT_A_CONV1: hipLaunchKernelGGL((gemm_kernel<T, PPT_A_CONV1, BM, BN>), grid, dim3(NT), 0, s, p); break;
    default: return PPT_EINVAL;
    }
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// tile choice (measured in one process on one MI355X, tools/gemm_shapes.py + bench.py with
// PPT_GEMM_SMALL_BELOW = 0 / 512 / 1e9 -> 8.14 / 7.50 / 7.08 ms per C2 step): 128x128 tiles read one LDS
// fragment per MFMA but keep only 2 workgroups (8 waves) on a CU; 64x64 tiles read two but run 4-5
// workgroups per CU, and for these short-K problems (K = 128..2048, every tile pays a load prologue and a
// store epilogue) the extra latency hiding wins up to several thousand tiles.  Only the half-gigabyte
// mini-PointNet GEMMs stay on 128x128 (and those that emit 64-row BatchNorm partials / 32-row pools).
template <typename T>
int launch_gemm(const ppt_gemm_params &p, hipStream_t s)
{
    static const int small_below = [] { const char *e = getenv("PPT_GEMM_SMALL_BELOW"); return e ? atoi(e) : 4096; }();
    const int64_t tiles128 = (int64_t)((p.N + 127) / 128) * ((p.M + 127) / 128) * (p.batch > 0 ? p.batch : 1);
    const bool need128 = p.col_sum || p.pool_max;       // 32-row chunk partials / pools need 64-wide wave tiles
    if (glds_h_ok<T>(p, tiles128)) {
        dim3 grid((p.N + 127) / 128, (p.M + 127) / 128, p.batch > 0 ? p.batch : 1);
        if (grid.y > 65535 || grid.z > 65535) return PPT_EUNSUPPORTED;
        // two stages (32 KiB, 4 workgroups = 16 waves per CU, prefetch distance 1) beat three (48 KiB, 3 workgroups,
        // distance 2) in the step: 4.73 vs 4.82 ms on C2, conv3 346 -> 325 us -- the fill rate follows the wave count
        // (tools/lds_fill_bench.hip) and a fourth co-resident workgroup hides more of the others' epilogues
        static const int two_stage = [] { const char *e = getenv("PPT_GEMM_H128_STAGES"); return !(e && atoi(e) == 3); }();
        if (two_stage) {
            switch (p.C ? epi_mask(p) : -1) {           // the three hot epilogues have their own small kernels
            case 0: hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2, 0>), grid, dim3(NT), 0, s, p); break;
            case EPI_GELU: hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2, EPI_GELU>), grid, dim3(NT), 0, s, p); break;
            case EPI_GROUP | EPI_STATS:
                hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2, EPI_GROUP | EPI_STATS>), grid, dim3(NT), 0, s, p); break;
            default: hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2>), grid, dim3(NT), 0, s, p); break;
            }
        } else {
            hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128>), grid, dim3(NT), 0, s, p);
        }
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    if (!need128 && tiles128 < small_below) return launch_gemm_tile<T, 64, 64>(p, s);
    return launch_gemm_tile<T, 128, 128>(p, s);
}

}  // namespace

extern "C" int ppt_gemm(const ppt_gemm_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_gemm_params q = *pp;
    if (!q.wave_prio) q.wave_prio = ppt_get_wave_priority();
    const ppt_gemm_params &p = q;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.B) return PPT_EINVAL;
    if (p.dtype != PPT_F32 && p.dtype != PPT_BF16 && p.dtype != PPT_F16) return PPT_EINVAL;
    // one 16-bit format per GEMM: outputs / saved pre-activations / pooled rows are fp32 or the operands' format
    if (p.dtype != PPT_F32) {
        if ((p.C && p.c_dtype != PPT_F32 && p.c_dtype != p.dtype) || (p.C2 && p.c2_dtype != PPT_F32 && p.c2_dtype != p.dtype) ||
            (p.pool_max && p.pool_dtype != PPT_F32 && p.pool_dtype != p.dtype))
            return PPT_EINVAL;
    }
    const int epc = p.dtype != PPT_F32 ? 8 : 4;
    if (p.K % epc || p.ldb % epc || ((uintptr_t)p.B & 15)) return PPT_EINVAL;
    if (p.a_mode != PPT_A_PLAIN && p.K > 1024) return PPT_EUNSUPPORTED;      // prologue constants live in an LDS table
    if (p.a_mode == PPT_A_CONV1) {
        if (!p.pts || !p.w1 || !p.b1) return PPT_EINVAL;
    } else {
        if (!p.A || p.lda % epc || ((uintptr_t)p.A & 15)) return PPT_EINVAL;
        if (p.a_mode == PPT_A_AFFINE_RELU && (!p.a_scale || !p.a_shift)) return PPT_EINVAL;
    }
    if ((p.col_sum == nullptr) != (p.col_sqsum == nullptr)) return PPT_EINVAL;
    if (p.pool_max && p.pool_rows != 0 && p.pool_rows != 16 && p.pool_rows != 32 && p.pool_rows != 64) return PPT_EINVAL;
#ifndef PPT_GEMM_STAMP
    if (p.pool_min && !p.pool_max) return PPT_EINVAL;
#endif
    if ((p.col_sum || p.pool_max) && ((p.N % 8) || ((uintptr_t)p.pool_min & 15) || ((uintptr_t)p.col_sum & 15) || ((uintptr_t)p.col_sqsum & 15) || ((uintptr_t)p.pool_max & 15)))
        return PPT_EUNSUPPORTED;                       // statistics / pooling exist only in the 16-byte epilogue
    if (p.group_add && p.group_rows <= 0) return PPT_EINVAL;
    if (p.row_scale && p.row_scale_rows <= 0) return PPT_EINVAL;
    if (p.batch > 1 && (p.C2 || p.col_sum || p.pool_max || p.residual || p.residual2 || p.dact_pre || p.group_add))
        return PPT_EUNSUPPORTED;
    hipStream_t s = ppt_stream(stream);
    return p.dtype == PPT_BF16 ? launch_gemm<bf16_t>(p, s) : p.dtype == PPT_F16 ? launch_gemm<f16_t>(p, s) : launch_gemm<float>(p, s);
}
