// ppt_common.h -- shared device helpers for the gfx950 kernels of libppt_hip.so.
// gfx950 only: wave64, DPP row_bcast, MFMA 32x32x16 bf16.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ppt_hip.h"

#define PPT_WAVE 64

#define PPT_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return PPT_ELAUNCH;          \
    } while (0)

typedef uint16_t bf16_t;   // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; NaN stays NaN through the plain cast path hipcc lowers to v_cvt_pk_bf16_f32
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}

// two floats -> one dword of bf16 (round-to-nearest-even): the vector conversion lowers to ONE v_cvt_pk_bf16_f32
// (two scalar casts cost two of them plus a shift and an or)
typedef float ppt_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ppt_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    const ppt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ppt_bf16x2));
}

// ---- IEEE half (PPT_F16): raw bits in a type of its own so that kernels can be instantiated per 16-bit format -------------------
struct f16_t { uint16_t b; };
typedef _Float16 ppt_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float f16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }      // round-to-nearest-even
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi)                                                  // one v_cvt_pk_f16_f32
{
    const ppt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ppt_f16x2));
}

template <typename T> struct dt;
template <> struct dt<float> {
    static __device__ __forceinline__ float load(const float *p) { return *p; }
    static __device__ __forceinline__ void store(float *p, float v) { *p = v; }
    static constexpr int code = PPT_F32;
};
template <> struct dt<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t *p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t *p, float v) { *p = f32_to_bf16(v); }
    static constexpr int code = PPT_BF16;
};

template <> struct dt<f16_t> {
    static __device__ __forceinline__ float load(const f16_t *p) { return f16_to_f32(p->b); }
    static __device__ __forceinline__ void store(f16_t *p, float v) { p->b = f32_to_f16(v); }
    static constexpr int code = PPT_F16;
};

// The two 16-bit operand formats behind one interface (T = bf16_t or f16_t): conversions of a packed pair and the MFMA forms.
// Fragments travel as uint4 (8 values); the matrix instructions of both formats run at the same rate.
typedef __attribute__((ext_vector_type(8))) __bf16 ppt_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 ppt_f16x8;
typedef __attribute__((ext_vector_type(16))) float ppt_f32x16;
typedef __attribute__((ext_vector_type(4))) float ppt_f32x4;
template <typename T> struct h16;
template <> struct h16<bf16_t> {
    static constexpr int code = PPT_BF16;
    static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return bf16_to_f32(v); }
    static __device__ __forceinline__ uint16_t from_f32(float v) { return f32_to_bf16(v); }
    static __device__ __forceinline__ ppt_f32x16 mfma32(uint4 a, uint4 b, ppt_f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ppt_bf16x8, a), __builtin_bit_cast(ppt_bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ ppt_f32x4 mfma16(uint4 a, uint4 b, ppt_f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ppt_bf16x8, a), __builtin_bit_cast(ppt_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct h16<f16_t> {
    static constexpr int code = PPT_F16;
    static __device__ __forceinline__ float lo(uint32_t w) { return f16_to_f32((uint16_t)(w & 0xffffu)); }
    static __device__ __forceinline__ float hi(uint32_t w) { return f16_to_f32((uint16_t)(w >> 16)); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_f16x2(a, b); }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return f16_to_f32(v); }
    static __device__ __forceinline__ uint16_t from_f32(float v) { return f32_to_f16(v); }
    static __device__ __forceinline__ ppt_f32x16 mfma32(uint4 a, uint4 b, ppt_f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ppt_f16x8, a), __builtin_bit_cast(ppt_f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ ppt_f32x4 mfma16(uint4 a, uint4 b, ppt_f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ppt_f16x8, a), __builtin_bit_cast(ppt_f16x8, b), c, 0, 0, 0);
    }
};
// run-time forms for code that carries a dtype code instead of a template parameter (code != PPT_F32)
__device__ __forceinline__ uint32_t pack2_dt(int code, float a, float b) { return code == PPT_F16 ? pack_f16x2(a, b) : pack_bf16x2(a, b); }
__device__ __forceinline__ float lo_dt(int code, uint32_t w) { return code == PPT_F16 ? h16<f16_t>::lo(w) : h16<bf16_t>::lo(w); }
__device__ __forceinline__ float hi_dt(int code, uint32_t w) { return code == PPT_F16 ? h16<f16_t>::hi(w) : h16<bf16_t>::hi(w); }
__device__ __forceinline__ uint16_t from_f32_dt(int code, float v) { return code == PPT_F16 ? f32_to_f16(v) : f32_to_bf16(v); }
__device__ __forceinline__ float to_f32_dt(int code, uint16_t v) { return code == PPT_F16 ? f16_to_f32(v) : bf16_to_f32(v); }

// ---- wave64 reductions over DPP (no LDS, no ds_bpermute) -----------------------------------
// butterfly inside each row of 16 lanes, then row_bcast15 / row_bcast31 fold the four rows;
// lane 63 ends with the full result, which v_readlane turns into a wave-uniform SGPR value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// Cross-half (lane ^ 32) reductions on v_permlane32_swap, with explicit wait states.  Found the hard way
// (tools/gemm_determinism.py): when hipcc (ROCm 7.2) feeds the swap from a chain of packed-fp32 ops (v_pk_add_f32 ...
// s_nop 0, v_mov, s_nop 1, v_permlane32_swap, VALU read) the BatchNorm chunk sums came out wrong in lanes 16-31 of
// ~1e-4 of the chunks, different chunks on every run; the builtin leaves 0-2 wait states on either side of the swap.
// The helper therefore does the copy and the swap itself: 8 wait states after whatever produced the value, 4 between
// copy and swap, 8 after the swap -- ~25 cycles per reduction, nothing next to the MFMA work around it.
// permlane32_halves(v, lo, hi): every lane l gets lo = v[l & 31], hi = v[32 + (l & 31)].
__device__ __forceinline__ void permlane32_halves(float v, float &lo, float &hi)
{
    uint32_t a = __float_as_uint(v), b;
    asm volatile("s_nop 7\n\tv_mov_b32 %1, %0\n\ts_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 7" : "+v"(a), "=&v"(b));
    lo = __uint_as_float(a); hi = __uint_as_float(b);
}
__device__ __forceinline__ float xor32_sum(float v) { float a, b; permlane32_halves(v, a, b); return a + b; }
__device__ __forceinline__ float xor32_max(float v) { float a, b; permlane32_halves(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_min(float v) { float a, b; permlane32_halves(v, a, b); return fminf(a, b); }

// Full-wave reductions with the DPP modifier ON the arithmetic instruction (v_max_u32_dpp ...): one VALU op per butterfly
// step instead of copy + DPP move + op.  Inside each row of 16 lanes: quad_perm, quad_perm, row_half_mirror, row_mirror;
// then row_bcast15 folds row 0 into 1 and 2 into 3, row_bcast31 folds rows 0-1 into 2-3: lane 63 holds the result, which
// v_readlane returns as a wave-uniform value.  (s_nop 1: a VALU write needs two wait states before a DPP read.)
#define PPT_DPP_REDUCE_ASM(INSN)                                                                     \
    "s_nop 1\n\t" INSN " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" INSN " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" INSN " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                \
    "s_nop 1\n\t" INSN " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                     \
    "s_nop 1\n\t" INSN " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                   \
    "s_nop 1\n\t" INSN " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                   \
    "s_nop 1"

__device__ __forceinline__ uint32_t wave_reduce_umax(uint32_t v)
{
    asm volatile(PPT_DPP_REDUCE_ASM("v_max_u32_dpp") : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_reduce_umin(uint32_t v)
{
    asm volatile(PPT_DPP_REDUCE_ASM("v_min_u32_dpp") : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t ppt_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t ppt_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

__device__ __forceinline__ float wave_reduce_sum(float v)
{
    asm volatile(PPT_DPP_REDUCE_ASM("v_add_f32_dpp") : "+v"(v));
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ float wave_reduce_max(float v)
{
    asm volatile(PPT_DPP_REDUCE_ASM("v_max_f32_dpp") : "+v"(v));
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

// ---- the reference's expanded-form squared distance ---------------------
// dvae.py:146-148:  ((-2 * dot) + |src|^2) + |dst|^2, dot = fma(a2,b2, fma(a1,b1, a0*b0)),
// |a|^2 = (x*x + y*y) + z*z; every step rounded to fp32 -- explicit _rn intrinsics so that no
// compiler flag can contract or reassociate them.
__device__ __forceinline__ float sqnorm3_rn(float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}
__device__ __forceinline__ float expanded_sqdist_rn(float qx, float qy, float qz, float nq, float px,
                                                    float py, float pz, float np)
{
    const float dot = __fmaf_rn(qz, pz, __fmaf_rn(qy, py, __fmul_rn(qx, px)));
    return __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), nq), np);
}
// order-preserving map float -> uint32 (handles the slightly negative distances of the expanded form)
__device__ __forceinline__ uint32_t float_order_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- LDS-DMA: global_load_lds_dwordx4 (64 lanes x 16 B -> 1 KiB of LDS starting at `lds`, in lane order) ----------------
// Issued from inline asm, NOT through __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) cannot tell which LDS bytes an LDS-DMA
// writes, so its wait-count pass puts `s_waitcnt vmcnt(0)` in front of the first ds_read that follows one -- found in the
// ISA of the 64x64 GEMM's K loop: the slab issued two stages AHEAD was waited for before the current slab's MFMAs, i.e. the
// three-stage ring ran as one load latency per slab.  The asm form is invisible to that pass; the kernels already place their
// own counted vmcnt waits (in-order completion), and a compiler-inserted vmcnt(N) for one of ITS loads stays safe with
// unknown younger operations in the queue (it then waits for more, never for less).
// `lds` must be wave-uniform; PPT_LDS_DMA_BUILTIN restores the builtin (A/B runs).
__device__ __forceinline__ void lds_dma16(const void *gsrc, void *lds)
{
#ifdef PPT_LDS_DMA_BUILTIN
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
#else
    const uint32_t m0v = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(m0v) : "memory", "m0");
#endif
}

static inline hipStream_t ppt_stream(void *s) { return (hipStream_t)s; }

// Compute units of the device the launch goes to -- taken from the STREAM (hipStreamGetDevice; the NULL stream: the current device),
// not from process-global state: one process may drive several GPUs ("device selected from the pointer / stream").
// Cached per device ordinal; a race on the cache writes the same value twice.
static inline int ppt_cu_count(hipStream_t s)
{
    static int cache[64] = {0};
    int dev = 0;
    if (s == nullptr || hipStreamGetDevice(s, &dev) != hipSuccess) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev < 0 || dev >= 64) dev = 0;
    int n = cache[dev];
    if (n <= 0) {
        n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (n <= 0) n = 256;
        cache[dev] = n;
    }
    return n;
}

// 16-byte NON-TEMPORAL store of a kernel's bulk output (the mini-PointNet activations: hundreds of MB per step, read back by
// the next kernel from HBM anyway): the lines do not linger dirty in L2, so the end-of-kernel write-back of the small kernels
// running beside this one (the prompt chain) has less to flush.  Same-box A/B against plain stores: C2 3.718 -> 3.702 ms,
// C3 6.787 -> 6.749 ms per step (tools/build_variant.sh + tools/ab_env.py).
__device__ __forceinline__ void ppt_store16_stream(void *p, uint4 v)
{
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), reinterpret_cast<u32x4_t *>(p));
}

// Wave priority of the prompt chain's kernels (s_setprio: issue arbitration among the waves of a SIMD; 0 is the default).  When
// only the prompt trains, the chain -- text backward, AdamW, text forward, head: ~200 small dependent kernels -- is the critical
// path of the step and its waves share SIMDs with the point tower's; with priority 3 they are issued first.  Same-box A/B
// (tools/build_variant.sh + tools/ab_env.py): C2 3.716 -> 3.676 ms, C4 2.989 -> 2.821 ms per step; where the point side trains
// (C3, C5) the tower is the critical path and the same setting costs 0.2-0.5 %, so it is a RUN-TIME property of the launch:
// ppt_set_wave_priority(n) (per host thread) applies to every launch that follows, the kernels take it as an argument.
// (HIP STREAM priority, by contrast, changed nothing: it orders kernel dispatch, not the waves already resident.)
extern "C" int ppt_get_wave_priority(void);
#define PPT_PRIO(prio) do { if (prio) __builtin_amdgcn_s_setprio(3); } while (0)

// Share of the chip the PERSISTENT point-tower kernels (one long-lived workgroup per CU: the mini-PointNet kernels) size their
// grids for, in percent; see ppt_set_persistent_occupancy in include/ppt_hip.h.
extern "C" int ppt_get_persistent_occupancy(void);
