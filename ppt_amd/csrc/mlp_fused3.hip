// mlp_fused3.hip -- round 6: the second form of the fused frozen-block kernel (attn.proj + DropPath + residual, LayerNorm, fc1, GELU,
// fc2, DropPath, residual (+ pos): point_encoder.py:14-30, 57-58, 69, 76-79, 103).  Same contract as mlp_fused.hip
// (ppt_vit_mlp_params), different schedule.  What round 5's stamps and this round's stream microbenchmark say about mlp_fused.hip:
//   * the weight stream is NOT its bound: tools/wstream_bench.hip pulls 48 B/clk/CU out of L2 with 8 waves and a register ring
//     (53 with 16), whatever the number of CUs in use; the kernel's 21-24 B/clk is simply its demand -- 196 KB per 128-unit slab
//     in the 8 300 cycles the slab's COMPUTE phases take;
//   * those phases are: GEMM1 3 936 cycles for 1 920 of MFMA work per SIMD (every LDS fragment feeds ONE 16x16x32 MFMA: 256 B/clk/CU,
//     the LDS's whole rate), GELU 1 592 cycles of pure vector work with the matrix pipe idle, GEMM2 2 358, barrier 450;
//   * x_mid = x + proj(a) is written to HBM and read back in the epilogue (2 x 25 MB per launch at C2's size) only because
//     nothing holds it during the MLP phase.
// This kernel:
//   (1) slabs of 256 hidden units: a wave owns 32 of them, so every LDS fragment of GEMM1 feeds TWO MFMAs (LDS traffic / 2);
//   (2) the GELU of slab j + 1 is issued BETWEEN the MFMAs of GEMM2(j) (same wave, same basic block): vector work under matrix
//       work; its result goes to the other half of a double-buffered slab, ONE barrier per slab;
//   (3) the fc2 accumulators START at x_mid (the proj prologue's result, or x): the residual never leaves registers -- no x_mid
//       round trip, no residual read in the epilogue.  DropPath's per-sample factor rides on the GELU output (rs * U) W2 = rs * (U W2);
//   (4) the weights arrive through two register rings (fc1: 4 k-steps = 8 KiB, fc2: 4 k-steps = 12 KiB per wave; in the step, beside the other streams, the deeper fc2 ring is worth 2 %) that wrap from one
//       slab into the next and from the last slab back to the first: the stream never stops at a phase boundary.
// Arithmetic differs from mlp_fused.hip only in rounding ORDER (residual first instead of last; rs folded before the 16-bit
// rounding of U): tests/test_kernels_gpu.py holds both to the same bounds against fp32 math on the same operands.
#include "ppt_common.h"
#include "ppt_act.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int D = 384, HID = 1536, HC = 256, NJ = HID / HC;       // model dims, hidden slab
constexpr int RB = 5, R = 16 * RB;                                 // row blocks / rows per chunk
constexpr int HP = 2 * D + 32, UP = 2 * HC + 32;                   // LDS pitches (bytes): = 32 mod 256 -> conflict-free b128 fragment reads
constexpr int H2_BYTES = R * HP, U_BYTES = R * UP;
constexpr int LDS_BYTES = H2_BYTES + 2 * U_BYTES + (2 * D + HID + D) * 4;
constexpr int K1 = D / 32, K2 = HC / 32;                           // k-steps of GEMM1 (12) / GEMM2 (8) per slab
#ifndef PPT_MLP3_D1
#define PPT_MLP3_D1 4
#endif
#ifndef PPT_MLP3_D2
#define PPT_MLP3_D2 4
#endif
constexpr int D1 = PPT_MLP3_D1, D2 = PPT_MLP3_D2;                  // ring depths in k-steps (must divide K1 / K2; tools/build_variant.sh for A/B)
static_assert(K1 % D1 == 0 && K2 % D2 == 0, "ring slots must line up from slab to slab");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
constexpr int W1_BYTES = HID * D * 2, W2_BYTES = D * HID * 2;

__device__ __forceinline__ float row16_sum3(float v)
{
    v += __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x4E, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x141, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x140, 0xf>(__float_as_uint(v)));
    return v;
}

// GELU (exact-erf form as an odd polynomial, ppt_act.h) with the saturation folded into a clamp of the erf argument: the
// polynomial was fitted on |x| < 4 and erf(4 / sqrt 2) = 0.99994, so clamping x to [-4, 4] for the erf factor replaces gelu_poly's
// compare + copysign + select by one v_med3_f32 (|difference| <= 6.3e-5 * |x| / 2 beyond 4: inside the polynomial's own error).
// hs = 0.5 * (the row's DropPath factor): the factor rides on the GELU output for free, (rs * U) W2 = rs * (U W2)
__device__ __forceinline__ float gelu_poly3(float x, float hs)
{
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float u = xc * xc;
    float p = fmaf(-3.161567230e-09f, u, 2.434219084e-07f);
    p = fmaf(p, u, -8.201724995e-06f);
    p = fmaf(p, u, 1.613346976e-04f);
    p = fmaf(p, u, -2.096408280e-03f);
    p = fmaf(p, u, 1.932974532e-02f);
    p = fmaf(p, u, -1.323507577e-01f);
    p = fmaf(p, u, 7.976950407e-01f);
    const float e = p * xc;
    const float h = hs * x;
    return fmaf(h, e, h);
}

// A workgroup barrier for LDS hand-overs ONLY: the LDS operations of this wave are complete (lgkmcnt(0)), global loads stay in
// flight.  __syncthreads() is a workgroup-scope release + acquire around s_barrier, i.e. s_waitcnt vmcnt(0): every barrier of the
// slab loop drained the weight rings (in-kernel stamps: ~1 200 cycles per slab at the barrier), and the barrier at the end of the
// prologue waited for the ring fill it was meant to overlap.  Nothing in this kernel hands GLOBAL data from wave to wave.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#ifdef PPT_MLP3_STAMP
#ifndef PPT_MLP3_STAMP_CHUNK
#define PPT_MLP3_STAMP_CHUNK 0           /* which of the workgroup's chunks is stamped (1: the second -- warm instruction cache / TLBs) */
#endif
#define MLP3_STAMP(j, slot) do { if (lane == 0 && chunk == (int)(blockIdx.x + PPT_MLP3_STAMP_CHUNK * gridDim.x)) stamps[((size_t)(blockIdx.x * 8 + w) * 8 + (j)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MLP3_STAMP(j, slot) do { } while (0)
#endif

template <typename F>
__global__ __launch_bounds__(512, 2) void vit_mlp3_kernel(const ppt_vit_mlp_params p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *h2 = smem, *ub = smem + H2_BYTES;
    float *gam = reinterpret_cast<float *>(smem + H2_BYTES + 2 * U_BYTES), *bet = gam + D, *b1s = bet + D, *b2s = b1s + HID;
    int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int l15 = lane & 15, kg = lane >> 4, lo16 = lane * 16, tid = threadIdx.x;
    // Every phase re-derives its per-lane addresses from an OPAQUE copy of the lane id.  Left alone, hipcc hoists every per-lane
    // address of the whole kernel (LDS fragment bases, row offsets, statistics slots ...) to the entry, finds no room for ~70 of
    // them beside the slab loop's 250 registers, spills them there and reloads them where they are used -- and each scratch
    // reload is followed by s_waitcnt vmcnt(0): in the prologue that drained the residual loads once per row block.
#define RELANE() do { asm volatile("" : "+v"(lane)); l15 = lane & 15; kg = lane >> 4; lo16 = lane * 16; tid = 64 * w + lane; } while (0)

    // fragment-ordered weights (ppt_vit_mlp3_retile): one wave-instruction = 1 KiB of consecutive bytes
    //   W1t[j][w][ks < 12][h < 2][lane][8] = W1[256 j + 32 w + 16 h + l15][32 ks + 8 kg ..)
    //   W2t[j][w][ks < 8][nb < 3][lane][8] = W2[48 w + 16 nb + l15][256 j + 32 ks + 8 kg ..)
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W1), 0, W1_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W2), 0, W2_BYTES, 0x00020000);
    // Each wave consumes its fragments in the order they lie in memory (24 KiB per slab and weight, then 7 x 24 KiB of the other
    // waves' to skip), so each ring is fed from ONE running scalar offset.  (First version: offsets computed from (slab, k-step)
    // at every load -- hipcc kept the ~48 per-k-step constants live in SGPRs across the slab loop, spilled 50 of them into VGPR
    // lanes and those VGPRs to scratch; every scratch reload is followed by s_waitcnt vmcnt(0), which in the prologue drained the
    // residual loads fifteen times: 19 000 cycles to ISSUE 25 loads.)  Past the last slab the offset leaves the buffer: such a
    // buffer load fetches nothing and returns zeros -- no wrap-around prefetch to pay for.
    constexpr int WAVE_SLAB = 24 * 1024;                                 // bytes of one wave's fragments per slab (either weight)
    int o1 = 0, o2 = 0;
    auto next1 = [&](uint4 &fa, uint4 &fb) {                             // the next k-step of the fc1 ring: two fragments
        fa = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lo16, o1, 0));
        fb = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lo16 + 1024, o1, 0));
        o1 += 2048;
    };
    auto next2 = [&](uint4 (&f)[3]) {                                    // the next k-step of the fc2 ring: three fragments
        f[0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, lo16, o2, 0));
        f[1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, lo16 + 1024, o2, 0));
        f[2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r2, lo16 + 2048, o2, 0));
        o2 += 3072;
    };
#ifdef PPT_MLP3_STAMP
    unsigned long long *stamps = (unsigned long long *)p.residual2;
#endif

    for (int chunk = blockIdx.x; chunk < p.n_chunks; chunk += gridDim.x) {
        MLP3_STAMP(6, 0);
        RELANE();
        const int row0 = chunk * p.rows_per_chunk;
        const int nrow = min(p.rows_per_chunk, p.M - row0);
        lds_barrier();                                                   // the previous chunk's readers are done
        MLP3_STAMP(5, 1);
        // acc2[rb][nb]: columns 48 w + 16 nb + 4 kg .. + 3 of row 16 rb + l15.  It STARTS as the residual (x, or x_mid below).
        // The residual rows are requested FIRST, in this layout, and nothing else reads x: the LayerNorm statistics are formed from
        // the accumulator layout too (first version: a separate 16-threads-per-row LayerNorm pass over x in three dependent
        // load -> reduce -> write rounds, then a second read of x for the accumulators -- 41 000 cycles of prologue per chunk).
        f32x4_t acc2[RB][3];
        float hrs[RB], rs1[RB];
        uint4 g1[D1][2], g2[D2][3];                                      // the two weight rings
        {
            // (once per workgroup) the per-channel constants are REQUESTED first and stored to LDS behind the requests below: vmcnt
            // retires in order, so stored behind requests issued EARLIER the ds_write waited for the residual rows to arrive from HBM
            float cst[6];
            const bool first = chunk == (int)blockIdx.x;
            if (first) {
                const int c = tid;
                cst[0] = c < D ? p.ln_w[c] : 0.f; cst[1] = c < D ? p.ln_b[c] : 0.f; cst[2] = (c < D && p.b2) ? p.b2[c] : 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) cst[3 + i] = p.b1 ? p.b1[c + 512 * i] : 0.f;
            }
            // ---- (a) the proj operand: the chunk's rows of the attention output -> LDS image (rows past the chunk: zeros), IN FRONT of
            // every other request -- vmcnt retires in order, and behind the residual rows' requests these stores waited for HBM to
            // deliver those first (9 200 cycles from the first request to the image's barrier).  (Held in registers across the
            // other requests instead, eight uint4 per thread, hipcc spilled 84-136 registers.)
            if (p.proj_a) {
                const bf16_t *A = (const bf16_t *)p.proj_a;
                for (int i = tid; i < R * (D / 8); i += 512) {
                    const int lr = i / (D / 8), c8 = i % (D / 8);
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (lr < nrow) v = *reinterpret_cast<const uint4 *>(A + (size_t)(row0 + lr) * D + 8 * c8);
                    *reinterpret_cast<uint4 *>(h2 + lr * HP + 16 * c8) = v;
                }
            }
            float4 xv[RB][3];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = row0 + min(16 * rb + l15, nrow - 1);
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) xv[rb][nb] = *reinterpret_cast<const float4 *>(p.x + (size_t)m * D + 48 * w + 16 * nb + 4 * kg);
            }
            // DropPath's per-sample factors (both branches), per row block of this lane: requested now, used far below
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = row0 + min(16 * rb + l15, nrow - 1);
                hrs[rb] = p.row_scale ? 0.5f * p.row_scale[m / p.row_scale_rows] : 0.5f;     // HALF the MLP branch's DropPath factor
                rs1[rb] = (p.proj_a && p.proj_row_scale) ? p.proj_row_scale[m / p.proj_row_scale_rows] : 1.0f;
            }
            if (first) {
                const int c = tid;
                if (c < D) { gam[c] = cst[0]; bet[c] = cst[1]; b2s[c] = cst[2]; }
#pragma unroll
                for (int i = 0; i < 3; ++i) b1s[c + 512 * i] = cst[3 + i];
            }
            MLP3_STAMP(5, 2);
            if (p.proj_a) {
                // ---- (b) acc2 = a . Wp^T, Wp in fragment order (ppt_vit_proj_retile: [w][12 nb + ks][lane][8]) four k-steps deep
                const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.proj_W), 0, 8 * 36 * 64 * 16, 0x00020000);
                auto pfrag = [&](int nb, int ks) {
                    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lo16, (w * 36 + nb * 12 + ks) * 1024, 0));
                };
                uint4 wf[4][3];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) wf[ks][nb] = pfrag(nb, ks);
                lds_barrier();
                MLP3_STAMP(5, 3);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) acc2[rb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const unsigned char *ha = h2 + l15 * HP + 16 * kg;
                uint4 fp[2][RB];                                          // (the image fragments of k-step ks + 1 requested before the MFMAs of k-step ks: GEMM1 below)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fp[0][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * HP);
#pragma unroll
                for (int ks = 0; ks < K1; ++ks) {
                    if (ks + 1 < K1) {
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) fp[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * HP + 64 * (ks + 1));
                    }
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb) acc2[rb][nb] = h16<F>::mfma16(wf[ks & 3][nb], fp[ks & 1][rb], acc2[rb][nb]);
                    if (ks + 4 < K1) {
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb) wf[ks & 3][nb] = pfrag(nb, ks + 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                MLP3_STAMP(5, 4);
                // ---- (c) x_mid = x + drop_path1 * (acc + bp): STAYS in acc2
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float r1 = rs1[rb];
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) {
                        const int n = 48 * w + 16 * nb + 4 * kg;
                        const float4 bv = p.proj_b ? *reinterpret_cast<const float4 *>(p.proj_b + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                        acc2[rb][nb][0] = (acc2[rb][nb][0] + bv.x) * r1 + xv[rb][nb].x; acc2[rb][nb][1] = (acc2[rb][nb][1] + bv.y) * r1 + xv[rb][nb].y;
                        acc2[rb][nb][2] = (acc2[rb][nb][2] + bv.z) * r1 + xv[rb][nb].z; acc2[rb][nb][3] = (acc2[rb][nb][3] + bv.w) * r1 + xv[rb][nb].w;
                    }
                }
            } else {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) acc2[rb][nb] = f32x4_t{xv[rb][nb].x, xv[rb][nb].y, xv[rb][nb].z, xv[rb][nb].w};
            }
        }
        MLP3_STAMP(5, 5);
        RELANE();
        // the two rings, filled per chunk, BEHIND the residual / proj phase and in front of the statistics' five barriers (kept live
        // from kernel entry, wrapping from the last slab into the next chunk's first, hipcc spilled all 80 ring registers around
        // the prologue -- 81 scratch stores right behind the loads)
        o1 = o2 = w * WAVE_SLAB;
#pragma unroll
        for (int s = 0; s < D1; ++s) next1(g1[s][0], g1[s][1]);
#pragma unroll
        for (int s = 0; s < D2; ++s) next2(g2[s]);
        // ---- LayerNorm of the residual rows FROM the accumulator layout (two passes, as nn.LayerNorm): a lane holds 12 of a row's 384
        // values per row block; the 32 partials of a row (8 waves x 4 lane groups) meet in LDS (the slab buffers are idle until GEMM1)
        {
            float *psum = reinterpret_cast<float *>(ub);                  // [R][32] partials, then [R] mean / rstd behind them
            float *stat = psum + R * 32;
            if (p.proj_a) lds_barrier();                                  // every wave is done reading `a` from the image
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float sum = 0.f;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) sum += (acc2[rb][nb][0] + acc2[rb][nb][1]) + (acc2[rb][nb][2] + acc2[rb][nb][3]);
                psum[(16 * rb + l15) * 32 + 4 * w + kg] = sum;
            }
            lds_barrier();
            if (tid < R) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) t += psum[tid * 32 + i];
                stat[tid] = t * (1.0f / (float)D);
            }
            lds_barrier();
            float mean_r[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int lr = 16 * rb + l15;
                mean_r[rb] = stat[lr];
                float q = 0.f;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = acc2[rb][nb][i] - mean_r[rb]; q = fmaf(d, d, q); }
                psum[lr * 32 + 4 * w + kg] = q;
            }
            lds_barrier();
            if (tid < R) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) t += psum[tid * 32 + i];
                stat[R + tid] = 1.0f / sqrtf(t * (1.0f / (float)D) + p.ln_eps);
            }
            lds_barrier();
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int lr = 16 * rb + l15;
                const float rstd = stat[R + lr];
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    const int n = 48 * w + 16 * nb + 4 * kg;
                    const float4 g = *reinterpret_cast<const float4 *>(gam + n), b = *reinterpret_cast<const float4 *>(bet + n);
                    uint2 o = make_uint2(0u, 0u);
                    if (lr < nrow)
                        o = make_uint2(h16<F>::pack2((acc2[rb][nb][0] - mean_r[rb]) * rstd * g.x + b.x, (acc2[rb][nb][1] - mean_r[rb]) * rstd * g.y + b.y),
                                       h16<F>::pack2((acc2[rb][nb][2] - mean_r[rb]) * rstd * g.z + b.z, (acc2[rb][nb][3] - mean_r[rb]) * rstd * g.w + b.w));
                    *reinterpret_cast<uint2 *>(h2 + lr * HP + 2 * n) = o;
                }
            }
        }
        MLP3_STAMP(5, 6);
        lds_barrier();                                                   // the image is complete (and psum / stat are dead: ub is free)
        MLP3_STAMP(6, 1);
        RELANE();

        f32x4_t a1[RB][2];
        // GEMM1(jj): a1 = W1[slab jj, this wave's 32 units] . H2^T -- every H2 fragment feeds two MFMAs; the ring slot a k-step used
        // is refilled with the k-step D1 further on (of the next slab -- of slab 0 behind the last: the next chunk's -- past the end)
        auto gemm1 = [&](int jj) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                 // the accumulators START at the fc1 bias (a lane's four hidden units)
                const float4 bv = *reinterpret_cast<const float4 *>(b1s + jj * HC + 32 * w + 16 * h + 4 * kg);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a1[rb][h] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
            }
            const unsigned char *ha = h2 + l15 * HP + 16 * kg;
            // the H2 fragments of k-step ks + 1 are requested before the MFMAs of k-step ks (two register sets): left to itself hipcc
            // emitted read -> wait -> two MFMAs, one LDS latency per pair (in-kernel stamps: 5 400 cycles for 3 840 of matrix work)
            uint4 fa[2][RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) fa[0][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * HP);
#pragma unroll
            for (int ks = 0; ks < K1; ++ks) {
                if (ks + 1 < K1) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) fa[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * HP + 64 * (ks + 1));
                }
                const uint4 wa = g1[ks % D1][0], wb = g1[ks % D1][1];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    a1[rb][0] = h16<F>::mfma16(wa, fa[ks & 1][rb], a1[rb][0]);
                    a1[rb][1] = h16<F>::mfma16(wb, fa[ks & 1][rb], a1[rb][1]);
                }
                if (ks + D1 == K1) o1 += 7 * WAVE_SLAB;                   // the ring moves on to this wave's part of the next slab
                next1(g1[ks % D1][0], g1[ks % D1][1]);
                __builtin_amdgcn_sched_barrier(0);                        // (k-steps stay in order: the scheduler may not pull reads back behind their use)
            }
        };
        // GELU of pair q = (rb, h) of a1 (slab jj) -> U[jj & 1], scaled by the row's DropPath factor
        auto gelu_pair = [&](int jj, int q) {
            const int rb = q >> 1, h = q & 1;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_poly3(a1[rb][h][i], hrs[rb]);
            *reinterpret_cast<uint2 *>(ub + (jj & 1) * U_BYTES + (16 * rb + l15) * UP + (32 * w + 16 * h + 4 * kg) * 2) =
                make_uint2(h16<F>::pack2(v[0], v[1]), h16<F>::pack2(v[2], v[3]));
        };
        // GEMM2(j): acc2 += U[j & 1] . W2[this wave's 48 columns, slab j]^T, with the GELU of slab j + 1 between its MFMAs
        auto gemm2 = [&](int j, auto with_gelu_c) {
            constexpr bool with_gelu = decltype(with_gelu_c)::value;
            const unsigned char *ua = ub + (j & 1) * U_BYTES + l15 * UP + 16 * kg;
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
                    for (int nb = 0; nb < 3; ++nb) acc2[rb][nb] = h16<F>::mfma16(g2[ks % D2][nb], fu[ks & 1][rb], acc2[rb][nb]);
                if (ks + D2 == K2) o2 += 7 * WAVE_SLAB;
                next2(g2[ks % D2]);
                if constexpr (with_gelu) {                               // 2 RB = 10 pairs over 8 k-steps
                    gelu_pair(j + 1, ks);
                    if (ks < 2 * RB - K2) gelu_pair(j + 1, K2 + ks);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        gemm1(0);
#pragma unroll
        for (int q = 0; q < 2 * RB; ++q) gelu_pair(0, q);
        lds_barrier();
        MLP3_STAMP(6, 2);
        // (the last slab is peeled: whether a GELU rides between GEMM2's MFMAs is then a compile-time property of the loop body,
        // which stays ONE basic block per slab -- with a run-time flag every k-step ended in a branch)
        for (int j = 0; j < NJ - 1; ++j) {
            MLP3_STAMP(j, 0);
            gemm1(j + 1);
            MLP3_STAMP(j, 1);
            gemm2(j, std::true_type{});                                  // reads U[j & 1]; writes U[(j + 1) & 1] (last read by GEMM2(j - 1), before the barrier above)
            MLP3_STAMP(j, 2);
            lds_barrier();
            MLP3_STAMP(j, 3);
        }
        MLP3_STAMP(NJ - 1, 0);
        RELANE();
        // the next block's "+ pos" rows (residual2): requested in front of the last slab's GEMM2 (the fc1 ring and a1 are dead: 72
        // registers free), added behind it -- in the epilogue proper each of these loads was a round trip with nothing beside it
        float4 r2v[RB][3];
#ifndef PPT_MLP3_STAMP
        if (p.residual2) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = row0 + min(16 * rb + l15, nrow - 1);
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) r2v[rb][nb] = *reinterpret_cast<const float4 *>(p.residual2 + (size_t)m * D + 48 * w + 16 * nb + 4 * kg);
            }
        } else
#endif
        {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) r2v[rb][nb] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        gemm2(NJ - 1, std::false_type{});
        MLP3_STAMP(7, 0);

        // ---- epilogue: out = acc2 (= residual + rs * U W2) + rs * b2 (+ pos)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int lr = 16 * rb + l15;
            if (lr < nrow) {
                const int m = row0 + lr;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    const int n = 48 * w + 16 * nb + 4 * kg;
                    const float4 bv = *reinterpret_cast<const float4 *>(b2s + n);
                    const float r = 2.0f * hrs[rb];
                    const float4 o = make_float4(fmaf(bv.x, r, acc2[rb][nb][0]) + r2v[rb][nb].x, fmaf(bv.y, r, acc2[rb][nb][1]) + r2v[rb][nb].y,
                                                 fmaf(bv.z, r, acc2[rb][nb][2]) + r2v[rb][nb].z, fmaf(bv.w, r, acc2[rb][nb][3]) + r2v[rb][nb].w);
                    *reinterpret_cast<float4 *>(p.out + (size_t)m * D + n) = o;
                }
            }
        }
        MLP3_STAMP(7, 1);
    }
}

// fragment order of the two weights (see ld1 / ld2 above): thread -> one 16-byte piece
__global__ __launch_bounds__(256) void vit_mlp3_retile_kernel(const bf16_t *__restrict__ W1, const bf16_t *__restrict__ W2,
                                                              bf16_t *__restrict__ W1t, bf16_t *__restrict__ W2t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                        // over NJ * 8 * 24 * 64 pieces (both weights)
    if (i >= NJ * 8 * 24 * 64) return;
    const int lane = i & 63, f = (i >> 6) % 24, w = (i / (64 * 24)) & 7, j = i / (64 * 24 * 8);
    const int l15 = lane & 15, kg = lane >> 4;
    {
        const int ks = f >> 1, h = f & 1;
        *reinterpret_cast<uint4 *>(W1t + (size_t)i * 8) =
            *reinterpret_cast<const uint4 *>(W1 + (size_t)(j * HC + 32 * w + 16 * h + l15) * D + 32 * ks + 8 * kg);
    }
    {
        const int ks = f / 3, nb = f % 3;
        *reinterpret_cast<uint4 *>(W2t + (size_t)i * 8) =
            *reinterpret_cast<const uint4 *>(W2 + (size_t)(48 * w + 16 * nb + l15) * HID + j * HC + 32 * ks + 8 * kg);
    }
}

}  // namespace

extern "C" int ppt_vit_mlp3_retile(const void *W1, const void *W2, void *W1t, void *W2t, void *stream)
{
    if (!W1 || !W2 || !W1t || !W2t || (((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W1t | (uintptr_t)W2t) & 15)) return PPT_EINVAL;
    hipLaunchKernelGGL(vit_mlp3_retile_kernel, dim3((NJ * 8 * 24 * 64 + 255) / 256), dim3(256), 0, ppt_stream(stream), (const bf16_t *)W1,
                       (const bf16_t *)W2, (bf16_t *)W1t, (bf16_t *)W2t);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_vit_mlp3_bf16(const ppt_vit_mlp_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_vit_mlp_params p = *pp;
    if (!p.x || !p.out || !p.W1 || !p.W2 || !p.ln_w || !p.ln_b || p.M <= 0) return PPT_EINVAL;
    if (p.D != D || p.hidden != HID) return PPT_EUNSUPPORTED;
    if (p.dtype != PPT_BF16 && p.dtype != PPT_F16 && p.dtype != 0) return PPT_EINVAL;
    if (p.row_scale && p.row_scale_rows <= 0) return PPT_EINVAL;
    if (((uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.W1 | (uintptr_t)p.W2 | (uintptr_t)p.residual2) & 15) return PPT_EINVAL;
    if (p.proj_a) {
        if (!p.proj_W || (p.proj_row_scale && p.proj_row_scale_rows <= 0)) return PPT_EINVAL;
        if (((uintptr_t)p.proj_a | (uintptr_t)p.proj_W | (uintptr_t)p.proj_b) & 15) return PPT_EINVAL;
    }
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)vit_mlp3_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)vit_mlp3_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return 0;
    }();
    (void)attrs_once;
    const int cus = ppt_cu_count(ppt_stream(stream));
    // chunks of at most R rows, a whole number of rounds over the CUs, every chunk as full as the division allows (mlp_fused.hip)
    int wgs = p.workgroups > 0 ? p.workgroups : cus;
    int rounds = 1;
    while ((int64_t)rounds * wgs * R < p.M) ++rounds;
    if (p.workgroups <= 0 && ppt_get_persistent_occupancy() < 100) wgs = (int)((p.M + (int64_t)rounds * R - 1) / ((int64_t)rounds * R));
    p.n_chunks = rounds * wgs;
    p.rows_per_chunk = (p.M + p.n_chunks - 1) / p.n_chunks;
    p.n_chunks = (p.M + p.rows_per_chunk - 1) / p.rows_per_chunk;
    const int grid = p.n_chunks < wgs ? p.n_chunks : wgs;
    if (p.dtype == PPT_F16) hipLaunchKernelGGL(vit_mlp3_kernel<f16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), p);
    else hipLaunchKernelGGL(vit_mlp3_kernel<bf16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
