// text_mlp_split.hip -- round 6: csrc/text_mlp.hip for the SPLIT16 products (fp32 operands multiplied as hi + lo IEEE-half pairs,
// gemm_common.h): the MLP half of a CLIP text-tower layer as one launch per direction when the text tower runs on fp32 operands --
// the whole-model split16 mode, and the mixed mode on weights whose text tower failed its load-time self-check
// (ULIP_WITH_IMAGE.calibrate_text_precision: checkpoint-like magnitudes; tools/ckpt_like_text_halves.py shows that BOTH halves of a
// layer need the fp32-grade products there).  On that path the prompt chain is 96 split16 tile GEMMs of 817 rows at ~22.7 us each:
// 3.2 ms of the 4.6 ms step.  This kernel takes 48 of them (c_fc, c_proj and their two input-gradient products per layer).
//
// Same decomposition as text_mlp.hip -- a workgroup = a 32-row block x a 256-unit slice of the hidden dimension, the hidden
// activation never leaves LDS, the eight slices' fp32 partial products are added up by the LayerNorm that reads them anyway -- with
//   * the rows of A (fp32) multiplied by 2^a_pow2, saturated to half's range (counted: ppt_text_mlp_params.split_overflow) and
//     split ONCE while they are staged: a hi image and a lo image in LDS;
//   * both weights split ONCE per weight version by ppt_text_mlp_retile_split (x 2^b_pow2) into fragment order, a fragment's hi
//     KiB followed by its lo KiB, streamed through the register rings;
//   * every product as three MFMAs (w_hi a_lo + w_lo a_hi + w_hi a_hi; lo x lo, < 2^-22 relative, dropped as in gemm_common.h),
//     fp32 accumulation, the accumulator multiplied by 2^-(a_pow2 + b_pow2) afterwards;
//   * the activation's output split the same way into the U images; the pre-activation is saved / read as fp32.
// A workgroup streams 2 x 512 KB of weight halves; 102 KB of LDS: one workgroup per CU, 208 of them at 817 rows.
#include "ppt_common.h"
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int D = 512, HID = 2048, SL = 256, NS = HID / SL;
constexpr int RB = 2, R = 16 * RB;
constexpr int AP = 2 * D + 32, UP = 2 * SL + 32;                   // LDS pitches (bytes): = 32 mod 256
constexpr int A_BYTES = R * AP, U_BYTES = R * UP;                  // ONE image (hi or lo)
constexpr int LDS_BYTES = 2 * A_BYTES + 2 * U_BYTES;
constexpr int K1 = D / 32, K2 = SL / 32;                           // k-steps of the two products (16 / 8)
constexpr int D1 = 4, D2 = 2;                                      // ring depths in k-steps
constexpr int WAVE_SLICE = 64 * 1024;                              // bytes of one wave's fragments per slice (either weight: hi + lo)
constexpr int W_BYTES = HID * D * 4;

__device__ __forceinline__ void lds_barrier_s()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// four fp32 values (already scaled and saturated) -> 8 bytes of hi halves, 8 bytes of lo halves
__device__ __forceinline__ void split4(const float (&x)[4], uint2 &H, uint2 &L)
{
    uint32_t h[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const _Float16 h0 = (_Float16)x[2 * q], h1 = (_Float16)x[2 * q + 1];
        const ppt_h2 hh = {h0, h1};
        const ppt_h2 ll = {(_Float16)(x[2 * q] - (float)h0), (_Float16)(x[2 * q + 1] - (float)h1)};
        h[q] = __builtin_bit_cast(uint32_t, hh);
        l[q] = __builtin_bit_cast(uint32_t, ll);
    }
    H = make_uint2(h[0], h[1]);
    L = make_uint2(l[0], l[1]);
}

__device__ __forceinline__ float row16_sum_s(float v)
{
    v += __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x4E, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x141, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x140, 0xf>(__float_as_uint(v)));
    return v;
}

// LN (forward only): A is the fp32 residual stream x_mid and ln_2 is applied while the block's rows are staged, as in text_mlp.hip
// (16 threads per row, two-pass statistics over DPP adds; slice 0 writes the statistics the LayerNorm backward needs).
template <int MODE, bool LN>
__global__ __launch_bounds__(512, 2) void text_mlp_split_kernel(const ppt_text_mlp_params p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ai = smem, *ui = smem + 2 * A_BYTES;            // hi image, lo image at + A_BYTES / + U_BYTES
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4, lo16 = lane * 16;
    PPT_PRIO(p.wave_prio);
    const int s = blockIdx.x % NS, row0 = (blockIdx.x / NS) * R;
    const int nrow = min(R, p.M - row0);
    const float sa = pow2f(p.split_a_pow2), inv = pow2f(-(p.split_a_pow2 + p.split_b_pow2));
    uint32_t over = 0;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W1), 0, W_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W2), 0, W_BYTES, 0x00020000);
    int o1 = (s * 8 + w) * WAVE_SLICE, o2 = o1;
    // one k-step of W1: [h < 2][hi, lo] x 1 KiB; of W2: [nb < 4][hi, lo] x 1 KiB
    auto next1 = [&](uint4 (&f)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lo16 + 1024 * i, o1, 0));
        o1 += 4096;
    };
    auto next2 = [&](uint4 (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, lo16 + 1024 * i, o2, 0));
        o2 += 8192;
    };
    uint4 g1[D1][4];
#pragma unroll
    for (int i = 0; i < D1; ++i) next1(g1[i]);

    // ---- the block's rows of A (fp32) -> scaled, saturated, split -> the hi and lo images (rows past M: zeros)
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
        const float mean = row16_sum_s(sm) * (1.0f / (float)D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
            const float d0 = xf[i].x - mean, d1 = xf[i].y - mean, d2 = xf[i].z - mean, d3 = xf[i].w - mean;
            q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
        }
        const float rstd = 1.0f / sqrtf(row16_sum_s(q) * (1.0f / (float)D) + p.ln_eps);
        if (p.ln_mean && s == 0 && j == 0 && r < nrow) { p.ln_mean[row0 + r] = mean; p.ln_rstd[row0 + r] = rstd; }
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
            const int c = 4 * (j + 16 * i);
            const float4 g = *reinterpret_cast<const float4 *>(p.ln_w + c), b = *reinterpret_cast<const float4 *>(p.ln_b + c);
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < nrow) {
                x[0] = split_saturate(((xf[i].x - mean) * rstd * g.x + b.x) * sa, over); x[1] = split_saturate(((xf[i].y - mean) * rstd * g.y + b.y) * sa, over);
                x[2] = split_saturate(((xf[i].z - mean) * rstd * g.z + b.z) * sa, over); x[3] = split_saturate(((xf[i].w - mean) * rstd * g.w + b.w) * sa, over);
            }
            uint2 H, L;
            split4(x, H, L);
            *reinterpret_cast<uint2 *>(ai + r * AP + 2 * c) = H;
            *reinterpret_cast<uint2 *>(ai + A_BYTES + r * AP + 2 * c) = L;
        }
    } else {
        const float *A = (const float *)p.A;
        constexpr int PIECES = R * (D / 4);                          // float4 pieces of the block (4096)
        float4 v[PIECES / 512];
#pragma unroll
        for (int it = 0; it < PIECES / 512; ++it) {
            const int i = threadIdx.x + 512 * it, lr = i / (D / 4), c4 = i % (D / 4);
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lr < nrow) v[it] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + lr) * p.lda + 4 * c4);
        }
#pragma unroll
        for (int it = 0; it < PIECES / 512; ++it) {
            const int i = threadIdx.x + 512 * it, lr = i / (D / 4), c4 = i % (D / 4);
            const float x[4] = {split_saturate(v[it].x * sa, over), split_saturate(v[it].y * sa, over),
                                split_saturate(v[it].z * sa, over), split_saturate(v[it].w * sa, over)};
            uint2 H, L;
            split4(x, H, L);
            *reinterpret_cast<uint2 *>(ai + lr * AP + 8 * c4) = H;
            *reinterpret_cast<uint2 *>(ai + A_BYTES + lr * AP + 8 * c4) = L;
        }
    }
    // the saved pre-activation (backward) and the bias (forward) of this lane's hidden units: requested now
    float4 prev[RB][2];
    float4 bv[2];
    const int hcol = SL * s + 32 * w + 4 * kg;                       // + 16 h: this lane's four hidden units of half h
    if (MODE == 1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = row0 + min(16 * rb + l15, nrow - 1);
                prev[rb][h] = *reinterpret_cast<const float4 *>((const float *)p.pre + (size_t)m * HID + hcol + 16 * h);
            }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[h] = p.b1 ? *reinterpret_cast<const float4 *>(p.b1 + hcol + 16 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    lds_barrier_s();

    // ---- product 1: a1[rb][h] = W1[slice, this wave's 32 units] . A^T, three MFMAs per fragment pair
    f32x4_t a1[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int h = 0; h < 2; ++h) a1[rb][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        const unsigned char *ha = ai + l15 * AP + 16 * kg;
        uint4 fh[2][RB], fl[2][RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            fh[0][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * AP);
            fl[0][rb] = *reinterpret_cast<const uint4 *>(ha + A_BYTES + rb * 16 * AP);
        }
#pragma unroll
        for (int ks = 0; ks < K1; ++ks) {
            if (ks + 1 < K1) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    fh[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * AP + 64 * (ks + 1));
                    fl[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ha + A_BYTES + rb * 16 * AP + 64 * (ks + 1));
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 wh = g1[ks % D1][2 * h], wl = g1[ks % D1][2 * h + 1];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    a1[rb][h] = h16<f16_t>::mfma16(wh, fl[ks & 1][rb], a1[rb][h]);
                    a1[rb][h] = h16<f16_t>::mfma16(wl, fh[ks & 1][rb], a1[rb][h]);
                    a1[rb][h] = h16<f16_t>::mfma16(wh, fh[ks & 1][rb], a1[rb][h]);
                }
            }
            if (ks + D1 < K1) next1(g1[ks % D1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the first k-steps of W2 fly under the activation
    uint4 g2[D2][8];
#pragma unroll
    for (int i = 0; i < D2; ++i) next2(g2[i]);
    // ---- un-scale, bias, activation -> split -> the U images; forward: the pre-activation is saved (fp32) for the backward
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[4] = {a1[rb][h][0] * inv, a1[rb][h][1] * inv, a1[rb][h][2] * inv, a1[rb][h][3] * inv};
            const int lr = 16 * rb + l15;
            if (MODE == 0) {
                v[0] += bv[h].x; v[1] += bv[h].y; v[2] += bv[h].z; v[3] += bv[h].w;
                if (p.pre && lr < nrow)
                    *reinterpret_cast<float4 *>((float *)p.pre + (size_t)(row0 + lr) * HID + hcol + 16 * h) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] / (1.0f + __expf(-1.702f * v[i]));                 // QuickGELU (ULIP_models.py:30-32)
            } else {
                const float x[4] = {prev[rb][h].x, prev[rb][h].y, prev[rb][h].z, prev[rb][h].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sg = 1.0f / (1.0f + __expf(-1.702f * x[i]));
                    v[i] *= sg * (1.0f + 1.702f * x[i] * (1.0f - sg));
                }
            }
            const float xs[4] = {split_saturate(v[0] * sa, over), split_saturate(v[1] * sa, over), split_saturate(v[2] * sa, over),
                                 split_saturate(v[3] * sa, over)};
            uint2 H, L;
            split4(xs, H, L);
            *reinterpret_cast<uint2 *>(ui + lr * UP + (32 * w + 16 * h + 4 * kg) * 2) = H;
            *reinterpret_cast<uint2 *>(ui + U_BYTES + lr * UP + (32 * w + 16 * h + 4 * kg) * 2) = L;
        }
    lds_barrier_s();

    // ---- product 2: acc[rb][nb] = W2[this wave's 64 columns, slice] . U^T
    f32x4_t acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[rb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        const unsigned char *ua = ui + l15 * UP + 16 * kg;
        uint4 fh[2][RB], fl[2][RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            fh[0][rb] = *reinterpret_cast<const uint4 *>(ua + rb * 16 * UP);
            fl[0][rb] = *reinterpret_cast<const uint4 *>(ua + U_BYTES + rb * 16 * UP);
        }
#pragma unroll
        for (int ks = 0; ks < K2; ++ks) {
            if (ks + 1 < K2) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    fh[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ua + rb * 16 * UP + 64 * (ks + 1));
                    fl[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ua + U_BYTES + rb * 16 * UP + 64 * (ks + 1));
                }
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const uint4 wh = g2[ks % D2][2 * nb], wl = g2[ks % D2][2 * nb + 1];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb][nb] = h16<f16_t>::mfma16(wh, fl[ks & 1][rb], acc[rb][nb]);
                    acc[rb][nb] = h16<f16_t>::mfma16(wl, fh[ks & 1][rb], acc[rb][nb]);
                    acc[rb][nb] = h16<f16_t>::mfma16(wh, fh[ks & 1][rb], acc[rb][nb]);
                }
            }
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
                    make_float4(acc[rb][nb][0] * inv, acc[rb][nb][1] * inv, acc[rb][nb][2] * inv, acc[rb][nb][3] * inv);
        }
    }
    split_report(over, p.split_overflow);
}

// fragment order: thread -> the hi and the lo 16-byte piece of 8 consecutive k of one weight row, both weights
//   W1t[s][w][ks < 16][h < 2][hi, lo][lane][8] <- W1[256 s + 32 w + 16 h + l15][32 ks + 8 kg ..) * 2^b_pow2      W1 [2048, 512] f32
//   W2t[s][w][ks < 8][nb < 4][hi, lo][lane][8] <- W2[64 w + 16 nb + l15][256 s + 32 ks + 8 kg ..) * 2^b_pow2      W2 [512, 2048] f32
__device__ __forceinline__ void split8(const float *src, float sb, uint4 &H, uint4 &L)
{
    uint32_t over = 0;          // (weights are fitted into half's range by the caller, ULIP_WITH_IMAGE._fit_split16_range: saturated, not counted)
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    const float x0[4] = {split_saturate(a.x * sb, over), split_saturate(a.y * sb, over), split_saturate(a.z * sb, over), split_saturate(a.w * sb, over)};
    const float x1[4] = {split_saturate(b.x * sb, over), split_saturate(b.y * sb, over), split_saturate(b.z * sb, over), split_saturate(b.w * sb, over)};
    uint2 h0, l0, h1, l1;
    split4(x0, h0, l0);
    split4(x1, h1, l1);
    H = make_uint4(h0.x, h0.y, h1.x, h1.y);
    L = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

__global__ __launch_bounds__(256) void text_mlp_retile_split_kernel(const float *__restrict__ W1, const float *__restrict__ W2,
                                                                    unsigned char *__restrict__ W1t, unsigned char *__restrict__ W2t, int b_pow2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                        // over NS * 8 * 32 * 64 pieces (both weights)
    if (i >= NS * 8 * 32 * 64) return;
    const int lane = i & 63, f = (i >> 6) & 31, w = (i >> 11) & 7, s = i >> 14;
    const int l15 = lane & 15, kg = lane >> 4;
    const float sb = pow2f(b_pow2);
    uint4 H, L;
    {
        const int ks = f >> 1, h = f & 1;
        split8(W1 + (size_t)(SL * s + 32 * w + 16 * h + l15) * D + 32 * ks + 8 * kg, sb, H, L);
        unsigned char *dst = W1t + (size_t)(s * 8 + w) * WAVE_SLICE + (size_t)(ks * 2 + h) * 2048 + lane * 16;
        *reinterpret_cast<uint4 *>(dst) = H;
        *reinterpret_cast<uint4 *>(dst + 1024) = L;
    }
    {
        const int ks = f >> 2, nb = f & 3;
        split8(W2 + (size_t)(64 * w + 16 * nb + l15) * HID + SL * s + 32 * ks + 8 * kg, sb, H, L);
        unsigned char *dst = W2t + (size_t)(s * 8 + w) * WAVE_SLICE + (size_t)(ks * 4 + nb) * 2048 + lane * 16;
        *reinterpret_cast<uint4 *>(dst) = H;
        *reinterpret_cast<uint4 *>(dst + 1024) = L;
    }
}

}  // namespace

extern "C" int ppt_text_mlp_retile_split(const float *W1, const float *W2, void *W1t, void *W2t, int b_pow2, void *stream)
{
    if (!W1 || !W2 || !W1t || !W2t || (((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W1t | (uintptr_t)W2t) & 15) || abs(b_pow2) > 24) return PPT_EINVAL;
    hipLaunchKernelGGL(text_mlp_retile_split_kernel, dim3((NS * 8 * 32 * 64 + 255) / 256), dim3(256), 0, ppt_stream(stream), W1, W2,
                       (unsigned char *)W1t, (unsigned char *)W2t, b_pow2);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// (called by ppt_text_mlp_pair for dtype == PPT_F32; csrc/text_mlp.hip has validated the common fields)
extern "C" int ppt_text_mlp_pair_split(const ppt_text_mlp_params *pp, void *stream)
{
    ppt_text_mlp_params p = *pp;
    if ((p.lda % 4) || abs(p.split_a_pow2) > 24 || abs(p.split_b_pow2) > 24) return PPT_EINVAL;
    const bool ln = p.ln_w != nullptr;
    if (ln && (p.mode != 0 || !p.ln_b || (((uintptr_t)p.ln_w | (uintptr_t)p.ln_b) & 15))) return PPT_EINVAL;
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)text_mlp_split_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_mlp_split_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_mlp_split_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return 0;
    }();
    (void)attrs_once;
    const int grid = NS * ((p.M + R - 1) / R);
    hipStream_t st = ppt_stream(stream);
    if (ln) hipLaunchKernelGGL((text_mlp_split_kernel<0, true>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    else if (p.mode == 0) hipLaunchKernelGGL((text_mlp_split_kernel<0, false>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    else hipLaunchKernelGGL((text_mlp_split_kernel<1, false>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
