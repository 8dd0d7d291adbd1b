// text_mlp.hip -- round 6: the MLP half of a CLIP text-tower layer as ONE launch per direction (ULIP_models.py:41-42, 49-51:
//     mlp = Sequential(c_fc: Linear(512, 2048), QuickGELU, c_proj: Linear(2048, 512));  x = x + mlp(ln_2(x))).
// On the prompt chain (817 rows with the shared prefix) the two linears were two dependent launches of the 64 x 64 tile loop --
// c_fc 12 us, c_proj (split-K 4) 10 us, and the 817 x 2048 hidden tensor written and read back in between -- in each direction,
// twelve layers deep, on the step's critical path.  What bounds such a launch is not FLOPs (13.7 GFLOP) but what ONE CU can pull
// out of L2 (48 B/clk with eight waves, tools/wstream_bench.hip) times how little of the weight each workgroup needs.
//
// Here a workgroup takes a 32-row block AND a 256-unit slice of the hidden dimension:
//     U[32, 256]   = act( A[32, 512] . W1[slice, :]^T (+ b1) )           16 k-steps, a wave owns 32 hidden units (2 x 1 register blocking)
//     P_s[32, 512] = U . W2[:, slice]^T                                  8 k-steps, a wave owns 64 output columns
// so it streams 2 x 256 KB of weights (5 us at the CU's rate), the hidden activation never leaves LDS, and the eight slices'
// partial products [8, M, 512] fp32 are added up -- with the residual and the bias -- by the LayerNorm kernel that reads the result
// anyway (ppt_layernorm_fwd_sum / ppt_layernorm_bwd_sum: fixed slice order, no atomics).  26 row blocks x 8 slices = 208 workgroups;
// blockIdx % 8 is the slice, so the workgroups of one XCD share one slice of the weights in its L2.  Measured alone at 817 rows
// (tools/text_mlp_bench.py): 20.2 us for the two launches -> 15.9 us with 64-row blocks (104 workgroups) -> 11.4 us with 32-row
// blocks; ring depths 4 / 8 make no difference (the workgroup's lifetime is its 2 x 128 MFMAs per wave plus four memory round trips).
//   forward  (mode 0): act = QuickGELU, W1 = c_fc.weight [2048, 512], W2 = c_proj.weight [512, 2048]; the pre-activation is saved
//                      (`pre`, 16-bit) when a backward will follow;
//   backward (mode 1): A = d out (16-bit), W1 = c_proj.weight^T [2048, 512], W2 = c_fc.weight^T [512, 2048],
//                      U = (A . W1^T) * QuickGELU'(pre): the input gradient of the whole branch (the tower is frozen: no dW).
// Both weights arrive in fragment order (ppt_text_mlp_retile), each wave's 32 KB per slice contiguous, through register rings fed
// from one running scalar offset (csrc/mlp_fused3.hip).
#include "ppt_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int D = 512, HID = 2048, SL = 256, NS = HID / SL;        // model width, hidden width, hidden slice, slices
#ifndef PPT_TMLP_RB
#define PPT_TMLP_RB 2
#endif
#ifndef PPT_TMLP_D1
#define PPT_TMLP_D1 4
#endif
#ifndef PPT_TMLP_D2
#define PPT_TMLP_D2 4
#endif
constexpr int RB = PPT_TMLP_RB, R = 16 * RB;                       // row blocks / rows per workgroup (tools/build_variant.sh for A/B)
constexpr int AP = 2 * D + 32, UP = 2 * SL + 32;                   // LDS pitches (bytes): = 32 mod 256
constexpr int A_BYTES = R * AP, U_BYTES = R * UP;
constexpr int LDS_BYTES = A_BYTES + U_BYTES;
constexpr int K1 = D / 32, K2 = SL / 32;                           // k-steps of the two products (16 / 8)
constexpr int D1 = PPT_TMLP_D1, D2 = PPT_TMLP_D2;                  // ring depths in k-steps (must divide 16 / 8)
constexpr int WAVE_SLICE = 32 * 1024;                              // bytes of one wave's fragments per slice (either weight)
constexpr int W_BYTES = HID * D * 2;

__device__ __forceinline__ void lds_barrier_t()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float row16_sum_t(float v)
{
    v += __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x4E, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x141, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x140, 0xf>(__float_as_uint(v)));
    return v;
}

// LN (forward only): A is the fp32 residual stream x_mid and ln_2 (ULIP_models.py:21-27, 50: LayerNorm computed in fp32) is applied
// while the block's rows are staged -- 16 threads per row, two-pass statistics over DPP adds as in rowgemm.hip -- so the LayerNorm
// launch in front of this kernel goes too; the eight slices' workgroups of a row block recompute it (64 KB of rows each, L2-warm),
// slice 0 writes the statistics the LayerNorm backward needs.
template <typename F, int MODE, bool LN>
__global__ __launch_bounds__(512, 2) void text_mlp_kernel(const ppt_text_mlp_params p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ai = smem, *ui = smem + A_BYTES;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4, lo16 = lane * 16;
    PPT_PRIO(p.wave_prio);
    const int s = blockIdx.x % NS, row0 = (blockIdx.x / NS) * R;
    const int nrow = min(R, p.M - row0);

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W1), 0, W_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W2), 0, W_BYTES, 0x00020000);
    int o1 = (s * 8 + w) * WAVE_SLICE, o2 = o1;
    auto next1 = [&](uint4 &fa, uint4 &fb) {
        fa = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lo16, o1, 0));
        fb = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lo16 + 1024, o1, 0));
        o1 += 2048;
    };
    auto next2 = [&](uint4 (&f)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) f[nb] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, lo16 + 1024 * nb, o2, 0));
        o2 += 4096;
    };
    // the first k-steps of W1 are requested before anything else: they do not depend on the rows
    uint4 g1[D1][2];
#pragma unroll
    for (int i = 0; i < D1; ++i) next1(g1[i][0], g1[i][1]);

    // ---- the block's rows of A -> LDS image (rows past M: zeros)
    if constexpr (LN) {
        static_assert(R == 32 && D == 512, "16 threads per row x 32 rows = the workgroup");
        const int r = threadIdx.x >> 4, j = threadIdx.x & 15;
        const float *src = (const float *)p.A + (size_t)(row0 + min(r, nrow - 1)) * p.lda;
        float4 xf[D / 64];
#pragma unroll
        for (int i = 0; i < D / 64; ++i) xf[i] = *reinterpret_cast<const float4 *>(src + 4 * (j + 16 * i));
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < D / 64; ++i) sm += (xf[i].x + xf[i].y) + (xf[i].z + xf[i].w);
        const float mean = row16_sum_t(sm) * (1.0f / (float)D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
            const float d0 = xf[i].x - mean, d1 = xf[i].y - mean, d2 = xf[i].z - mean, d3 = xf[i].w - mean;
            q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
        }
        const float rstd = 1.0f / sqrtf(row16_sum_t(q) * (1.0f / (float)D) + p.ln_eps);
        if (p.ln_mean && s == 0 && j == 0 && r < nrow) { p.ln_mean[row0 + r] = mean; p.ln_rstd[row0 + r] = rstd; }
        unsigned char *dst = ai + r * AP;
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
            const int c = 4 * (j + 16 * i);
            const float4 g = *reinterpret_cast<const float4 *>(p.ln_w + c), b = *reinterpret_cast<const float4 *>(p.ln_b + c);
            uint2 o = make_uint2(0u, 0u);
            if (r < nrow)
                o = make_uint2(h16<F>::pack2((xf[i].x - mean) * rstd * g.x + b.x, (xf[i].y - mean) * rstd * g.y + b.y),
                               h16<F>::pack2((xf[i].z - mean) * rstd * g.z + b.z, (xf[i].w - mean) * rstd * g.w + b.w));
            *reinterpret_cast<uint2 *>(dst + 2 * c) = o;
        }
    } else {
        const F *A = (const F *)p.A;
#pragma unroll
        for (int it = 0; it < (R * (D / 8) + 511) / 512; ++it) {
            const int i = threadIdx.x + 512 * it, lr = i / (D / 8), c8 = i % (D / 8);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (lr < nrow) v = *reinterpret_cast<const uint4 *>(A + (size_t)(row0 + lr) * p.lda + 8 * c8);
            if (lr < R) *reinterpret_cast<uint4 *>(ai + lr * AP + 16 * c8) = v;
        }
    }
    // the saved pre-activation (backward) and the bias (forward) of this lane's hidden units: requested now
    uint2 prev[RB][2];
    float4 bv[2];
    const int hcol = SL * s + 32 * w + 4 * kg;                       // + 16 h: this lane's four hidden units of half h
    if (MODE == 1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = row0 + min(16 * rb + l15, nrow - 1);
                prev[rb][h] = *reinterpret_cast<const uint2 *>((const F *)p.pre + (size_t)m * HID + hcol + 16 * h);
            }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[h] = p.b1 ? *reinterpret_cast<const float4 *>(p.b1 + hcol + 16 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    lds_barrier_t();

    // ---- product 1: a1[rb][h] = W1[slice, this wave's 32 units] . A^T (transposed product: a lane holds four hidden units of a row)
    f32x4_t a1[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int h = 0; h < 2; ++h) a1[rb][h] = MODE == 0 ? f32x4_t{bv[h].x, bv[h].y, bv[h].z, bv[h].w} : f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        const unsigned char *ha = ai + l15 * AP + 16 * kg;
        uint4 fa[2][RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fa[0][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * AP);
#pragma unroll
        for (int ks = 0; ks < K1; ++ks) {
            if (ks + 1 < K1) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * AP + 64 * (ks + 1));
            }
            const uint4 wa = g1[ks % D1][0], wb = g1[ks % D1][1];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                a1[rb][0] = h16<F>::mfma16(wa, fa[ks & 1][rb], a1[rb][0]);
                a1[rb][1] = h16<F>::mfma16(wb, fa[ks & 1][rb], a1[rb][1]);
            }
            if (ks + D1 < K1) next1(g1[ks % D1][0], g1[ks % D1][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the first k-steps of W2 fly under the activation
    uint4 g2[D2][4];
#pragma unroll
    for (int i = 0; i < D2; ++i) next2(g2[i]);
    // ---- activation -> the U image; forward: the pre-activation is saved for the backward
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[4] = {a1[rb][h][0], a1[rb][h][1], a1[rb][h][2], a1[rb][h][3]};
            const int lr = 16 * rb + l15;
            if (MODE == 0) {
                if (p.pre && lr < nrow)
                    *reinterpret_cast<uint2 *>((F *)p.pre + (size_t)(row0 + lr) * HID + hcol + 16 * h) =
                        make_uint2(h16<F>::pack2(v[0], v[1]), h16<F>::pack2(v[2], v[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] / (1.0f + __expf(-1.702f * v[i]));                 // QuickGELU (ULIP_models.py:30-32)
            } else {
                const float x[4] = {h16<F>::lo(prev[rb][h].x), h16<F>::hi(prev[rb][h].x), h16<F>::lo(prev[rb][h].y), h16<F>::hi(prev[rb][h].y)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sg = 1.0f / (1.0f + __expf(-1.702f * x[i]));
                    v[i] *= sg * (1.0f + 1.702f * x[i] * (1.0f - sg));
                }
            }
            *reinterpret_cast<uint2 *>(ui + lr * UP + (32 * w + 16 * h + 4 * kg) * 2) =
                make_uint2(h16<F>::pack2(v[0], v[1]), h16<F>::pack2(v[2], v[3]));
        }
    lds_barrier_t();

    // ---- product 2: acc[rb][nb] = W2[this wave's 64 columns, slice] . U^T
    f32x4_t acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[rb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        const unsigned char *ua = ui + l15 * UP + 16 * kg;
        uint4 fu[2][RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fu[0][rb] = *reinterpret_cast<const uint4 *>(ua + rb * 16 * UP);
#pragma unroll
        for (int ks = 0; ks < K2; ++ks) {
            if (ks + 1 < K2) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fu[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ua + rb * 16 * UP + 64 * (ks + 1));
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[rb][nb] = h16<F>::mfma16(g2[ks % D2][nb], fu[ks & 1][rb], acc[rb][nb]);
            if (ks + D2 < K2) next2(g2[ks % D2]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- the slice's partial product: parts[s][row][64 w + 16 nb + 4 kg ..]
    float *out = p.parts + (size_t)s * p.M * D;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int lr = 16 * rb + l15;
        if (lr < nrow) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                *reinterpret_cast<float4 *>(out + (size_t)(row0 + lr) * D + 64 * w + 16 * nb + 4 * kg) =
                    make_float4(acc[rb][nb][0], acc[rb][nb][1], acc[rb][nb][2], acc[rb][nb][3]);
        }
    }
}

// fragment order (see next1 / next2): thread -> one 16-byte piece of each weight
//   W1t[s][w][ks < 16][h < 2][lane][8] = W1[256 s + 32 w + 16 h + l15][32 ks + 8 kg ..)      W1 [2048, 512] row-major
//   W2t[s][w][ks < 8][nb < 4][lane][8] = W2[64 w + 16 nb + l15][256 s + 32 ks + 8 kg ..)       W2 [512, 2048] row-major
__global__ __launch_bounds__(256) void text_mlp_retile_kernel(const bf16_t *__restrict__ W1, const bf16_t *__restrict__ W2,
                                                              bf16_t *__restrict__ W1t, bf16_t *__restrict__ W2t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                        // over NS * 8 * 32 * 64 pieces (both weights)
    if (i >= NS * 8 * 32 * 64) return;
    const int lane = i & 63, f = (i >> 6) & 31, w = (i >> 11) & 7, s = i >> 14;
    const int l15 = lane & 15, kg = lane >> 4;
    {
        const int ks = f >> 1, h = f & 1;
        *reinterpret_cast<uint4 *>(W1t + (size_t)i * 8) =
            *reinterpret_cast<const uint4 *>(W1 + (size_t)(SL * s + 32 * w + 16 * h + l15) * D + 32 * ks + 8 * kg);
    }
    {
        const int ks = f >> 2, nb = f & 3;
        *reinterpret_cast<uint4 *>(W2t + (size_t)i * 8) =
            *reinterpret_cast<const uint4 *>(W2 + (size_t)(64 * w + 16 * nb + l15) * HID + SL * s + 32 * ks + 8 * kg);
    }
}

}  // namespace

extern "C" int ppt_text_mlp_retile(const void *W1, const void *W2, void *W1t, void *W2t, void *stream)
{
    if (!W1 || !W2 || !W1t || !W2t || (((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W1t | (uintptr_t)W2t) & 15)) return PPT_EINVAL;
    hipLaunchKernelGGL(text_mlp_retile_kernel, dim3((NS * 8 * 32 * 64 + 255) / 256), dim3(256), 0, ppt_stream(stream), (const bf16_t *)W1,
                       (const bf16_t *)W2, (bf16_t *)W1t, (bf16_t *)W2t);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_text_mlp_pair_split(const ppt_text_mlp_params *pp, void *stream);     // text_mlp_split.hip

extern "C" int ppt_text_mlp_pair(const ppt_text_mlp_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_text_mlp_params p = *pp;
    if (!p.A || !p.W1 || !p.W2 || !p.parts || p.M <= 0 || p.lda < D) return PPT_EINVAL;
    if (p.D != D || p.hidden != HID) return PPT_EUNSUPPORTED;
    if (p.dtype != PPT_BF16 && p.dtype != PPT_F16 && p.dtype != PPT_F32) return PPT_EINVAL;
    if (p.mode != 0 && p.mode != 1) return PPT_EINVAL;
    if (p.mode == 1 && !p.pre) return PPT_EINVAL;
    if (((uintptr_t)p.A | (uintptr_t)p.W1 | (uintptr_t)p.W2 | (uintptr_t)p.parts | (uintptr_t)p.pre | (uintptr_t)p.b1) & 15) return PPT_EINVAL;
    if (p.wave_prio == 0) p.wave_prio = ppt_get_wave_priority();
    if (p.dtype == PPT_F32) return ppt_text_mlp_pair_split(&p, stream);          // fp32 operands as hi + lo half pairs (text_mlp_split.hip)
    const bool ln = p.ln_w != nullptr;
    if (ln && (p.mode != 0 || !p.ln_b || R != 32 || (p.lda % 4) || (((uintptr_t)p.ln_w | (uintptr_t)p.ln_b) & 15))) return PPT_EINVAL;
    if (!ln && (p.lda % 8)) return PPT_EINVAL;
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)text_mlp_kernel<bf16_t, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_mlp_kernel<bf16_t, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_mlp_kernel<f16_t, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_mlp_kernel<f16_t, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_mlp_kernel<bf16_t, 0, (R == 32)>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_mlp_kernel<f16_t, 0, (R == 32)>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return 0;
    }();
    (void)attrs_once;
    const int grid = NS * ((p.M + R - 1) / R);
    hipStream_t st = ppt_stream(stream);
    if (ln) {
        if (p.dtype == PPT_F16) hipLaunchKernelGGL((text_mlp_kernel<f16_t, 0, (R == 32)>), dim3(grid), dim3(512), LDS_BYTES, st, p);
        else hipLaunchKernelGGL((text_mlp_kernel<bf16_t, 0, (R == 32)>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    } else if (p.dtype == PPT_F16) {
        if (p.mode == 0) hipLaunchKernelGGL((text_mlp_kernel<f16_t, 0, false>), dim3(grid), dim3(512), LDS_BYTES, st, p);
        else hipLaunchKernelGGL((text_mlp_kernel<f16_t, 1, false>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    } else {
        if (p.mode == 0) hipLaunchKernelGGL((text_mlp_kernel<bf16_t, 0, false>), dim3(grid), dim3(512), LDS_BYTES, st, p);
        else hipLaunchKernelGGL((text_mlp_kernel<bf16_t, 1, false>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    }
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
