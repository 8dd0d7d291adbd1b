// mlp_fused.hip -- the MLP half of a frozen PointBERT block as ONE kernel (point_encoder.py:14-30, 69, 78-79):
//     x <- x + drop_path * ( GELU( LayerNorm(x) W1^T + b1 ) W2^T + b2 ) (+ pos)
// for x [M, 384] fp32 (the residual stream, updated in place), W1 [1536, 384], W2 [384, 1536] bf16.
//
// Unfused (norm.hip + two launches of gemm.hip) this is LayerNorm 6.4 us + fc1 46 us + fc2 53 us = 105 of the 190 us of a
// block at B = 32, and what bounds fc1 / fc2 is not the matrix pipe (23 us of MFMA work for both) but what crosses HBM and
// the L2 -> LDS path: the [M, 1536] hidden tensor is written and read back (2 x 50 MB), the LayerNorm output too, and both
// GEMMs re-fetch their A panels per column tile.  Here a workgroup owns a CHUNK of up to 80 token rows from the LayerNorm to
// the residual add: the normalised rows (bf16) stay in LDS, the hidden activation exists only as a [80 x 128] bf16 slab in
// LDS (double-buffered), the output accumulates in registers across the 12 hidden slabs.  What moves is the weights: every
// workgroup streams W1 and W2 (2.36 MB) once per chunk, straight from L2 into the MFMA operand registers -- each wave
// loads the fragments of its own columns, one slab ahead of their use -- so the launch is bounded by that stream
// (~30 B/clk/CU: ~38 us at B = 32), not by HBM.
//
// Per hidden slab j (128 hidden units) and wave w (8 waves):
//   GEMM1: U[:, 16 w .. 16 w + 15] = GELU(H2[80 x 384] . W1[slab rows 16 w .., :]^T + b1)   5 row blocks x 12 k-steps
//   barrier (the slab is complete)
//   GEMM2: acc[80 x (48 w .. 48 w + 47)] += U[80 x 128] . W2[48 w .., slab]^T                5 x 3 blocks x 4 k-steps
// Products are formed transposed (D[n][m], v_mfma_f32_16x16x32_bf16 with the weight as first operand), so a lane holds
// four consecutive columns of a row: 8-byte LDS writes of the slab, 16-byte accesses of the residual stream.
//
// Round 3 -- the attention branch's tail rides in front (p.proj_a != NULL): the chunk's rows of the attention output `a`
// (16-bit, [M, 384]) go to the LDS image first, x_mid = x + drop_path1 * (a Wp^T + bp) (point_encoder.py:57-58, 77) is formed
// with Wp streamed in fragment order exactly like W2 (same output layout: a lane owns 4 consecutive columns of a row in each
// of its 15 blocks), written back to the residual stream, and normalised IN REGISTERS: the row statistics (two passes, as
// nn.LayerNorm) meet through LDS (the slab buffers are idle until GEMM1), and the normalised rows replace `a` in the image.
// One launch and one read + one write of the fp32 residual stream fewer per block than proj (rowgemm.hip) + this kernel.
#include "ppt_common.h"
#include "ppt_act.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int D = 384, HID = 1536, HC = 128, NJ = HID / HC;       // model dims, hidden slab
#ifndef PPT_MLP_RB
#define PPT_MLP_RB 5
#endif
constexpr int RB = PPT_MLP_RB, R = 16 * RB;                        // row blocks / rows per chunk (capacity)
constexpr int HP = 2 * D + 32, UP = 2 * HC + 32;                   // LDS pitches (bytes): = 32 mod 256 -> conflict-free b128 fragment reads
constexpr int H2_BYTES = R * HP, U_BYTES = R * UP;
constexpr int LDS_BYTES = H2_BYTES + 2 * U_BYTES + (2 * D + HID + D) * 4;

__device__ __forceinline__ float row16_sum(float v)
{
    v += __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x4E, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x141, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x140, 0xf>(__float_as_uint(v)));
    return v;
}

// Diagnostic build only (tools/mlp_stamp.py compiles this file with -DPPT_MLP_STAMP): s_memtime stamps of the first chunk's
// phases go to the buffer passed in `residual2`: [workgroup][wave][slab 0..11 + 2][8].
#ifdef PPT_MLP_STAMP
#define MLP_STAMP(j, slot) do { if (lane == 0 && chunk == (int)blockIdx.x && (j) >= 0) stamps[((size_t)(blockIdx.x * 8 + w) * 14 + (j)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MLP_STAMP(j, slot) do { } while (0)
#endif

template <typename F>
__global__ __launch_bounds__(512, 2) void vit_mlp_kernel(const ppt_vit_mlp_params p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *h2 = smem, *ub = smem + H2_BYTES;
    float *gam = reinterpret_cast<float *>(smem + H2_BYTES + 2 * U_BYTES), *bet = gam + D, *b1s = bet + D, *b2s = b1s + HID;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const bf16_t *W1 = (const bf16_t *)p.W1, *W2 = (const bf16_t *)p.W2;

    for (int c = threadIdx.x; c < D; c += 512) { gam[c] = p.ln_w[c]; bet[c] = p.ln_b[c]; b2s[c] = p.b2 ? p.b2[c] : 0.f; }
    for (int c = threadIdx.x; c < HID; c += 512) b1s[c] = p.b1 ? p.b1[c] : 0.f;

    // W1 / W2 arrive in FRAGMENT ORDER (ppt_vit_mlp_retile): [slab j][wave w][fragment f < 12][lane][8 bf16], so that one
    // load instruction of a wave is 1 KB of consecutive bytes -- eight full cache lines.  Read from the row-major weights a
    // fragment is sixteen 64-byte row pieces (half lines): measured 14 B/clk/CU out of L2 instead of ~30 (in-kernel stamps:
    // 14 300 cycles per slab for 196 KB against 9 300 in fragment order).
    auto frag = [&](const bf16_t *Wt, int j, int f) {
        return *reinterpret_cast<const uint4 *>(Wt + ((size_t)((j * 8 + w) * 12 + f) * 64 + lane) * 8);
    };
    // this wave's weight fragments: B1 = W1 rows (hidden units) 16 w .. + 15 of the slab, B2 = W2 rows (outputs) 48 w .. + 47
    uint4 b1f[D / 32], b2f[3][HC / 32];
    auto load_b1 = [&](int j) {
#pragma unroll
        for (int s = 0; s < D / 32; ++s) b1f[s] = frag(W1, j, s);
    };
#ifdef PPT_MLP_STAMP
    unsigned long long *stamps = (unsigned long long *)p.residual2;
#endif
    for (int chunk = blockIdx.x; chunk < p.n_chunks; chunk += gridDim.x) {
        MLP_STAMP(12, 0);
        const int row0 = chunk * p.rows_per_chunk;
        const int nrow = min(p.rows_per_chunk, p.M - row0);
        __syncthreads();                                                 // constants are in LDS; the previous chunk's readers are done
        if (p.proj_a) {
            // ---- (a) the chunk's rows of the attention output -> LDS image (16 bytes per thread and step; rows past the chunk: zeros)
            {
                const bf16_t *A = (const bf16_t *)p.proj_a;
                for (int i = threadIdx.x; i < R * (D / 8); i += 512) {
                    const int lr = i / (D / 8), c8 = i % (D / 8);
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (lr < nrow) v = *reinterpret_cast<const uint4 *>(A + (size_t)(row0 + lr) * D + 8 * c8);
                    *reinterpret_cast<uint4 *>(h2 + lr * HP + 16 * c8) = v;
                }
            }
            __syncthreads();
            // ---- (b) acc = a . Wp^T: this wave's 48 output columns, weight fragments streamed four k-steps deep (register budget 128)
            // (buffer loads: one VGPR of lane offset + a scalar fragment offset.  With plain pointers hipcc formed the 36 per-lane
            // 64-bit fragment addresses at kernel entry, spilled them around the MLP phase's 248 registers and reloaded one inside
            // every MFMA group of this loop.)
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.proj_W), 0, 8 * 36 * 64 * 16, 0x00020000);
            auto pfrag = [&](int nb, int ks) {
                return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, (w * 36 + nb * 12 + ks) * 1024, 0));
            };
            f32x4_t pa[RB][3];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) pa[rb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            {
                uint4 wf[4][3];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) wf[ks][nb] = pfrag(nb, ks);
                const unsigned char *ha = h2 + l15 * HP + 16 * kg;
#pragma unroll
                for (int ks = 0; ks < D / 32; ++ks) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const uint4 f = *reinterpret_cast<const uint4 *>(ha + rb * 16 * HP + 64 * ks);
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb) pa[rb][nb] = h16<F>::mfma16(wf[ks & 3][nb], f, pa[rb][nb]);
                    }
                    if (ks + 4 < D / 32) {
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb) wf[ks & 3][nb] = pfrag(nb, ks + 4);
                    }
                }
            }
            // ---- (c) x_mid = x + drop_path1 * (acc + bp) -> the residual stream; partial row sums -> LDS
            float *psum = reinterpret_cast<float *>(ub);                  // [R][32] partials, then [R] mean / rstd behind them
            float *stat = psum + R * 32;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int lr = 16 * rb + l15;
                const int m = row0 + min(lr, nrow - 1);
                const float rs1 = p.proj_row_scale ? p.proj_row_scale[m / p.proj_row_scale_rows] : 1.0f;
                float sum = 0.f;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    const int n = 48 * w + 16 * nb + 4 * kg;
                    const float4 xv = *reinterpret_cast<const float4 *>(p.x + (size_t)m * D + n);
                    const float4 bv = p.proj_b ? *reinterpret_cast<const float4 *>(p.proj_b + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    pa[rb][nb][0] = (pa[rb][nb][0] + bv.x) * rs1 + xv.x; pa[rb][nb][1] = (pa[rb][nb][1] + bv.y) * rs1 + xv.y;
                    pa[rb][nb][2] = (pa[rb][nb][2] + bv.z) * rs1 + xv.z; pa[rb][nb][3] = (pa[rb][nb][3] + bv.w) * rs1 + xv.w;
                    if (lr < nrow)
                        *reinterpret_cast<float4 *>(p.out + (size_t)m * D + n) = make_float4(pa[rb][nb][0], pa[rb][nb][1], pa[rb][nb][2], pa[rb][nb][3]);
                    sum += (pa[rb][nb][0] + pa[rb][nb][1]) + (pa[rb][nb][2] + pa[rb][nb][3]);
                }
                psum[lr * 32 + 4 * w + kg] = sum;
            }
            __syncthreads();                                              // (also: every wave is done reading `a` from the image)
            if (threadIdx.x < R) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) t += psum[threadIdx.x * 32 + i];
                stat[threadIdx.x] = t * (1.0f / (float)D);
            }
            __syncthreads();
            float mean_r[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int lr = 16 * rb + l15;
                mean_r[rb] = stat[lr];
                float q = 0.f;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = pa[rb][nb][i] - mean_r[rb]; q = fmaf(d, d, q); }
                psum[lr * 32 + 4 * w + kg] = q;
            }
            __syncthreads();
            if (threadIdx.x < R) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) t += psum[threadIdx.x * 32 + i];
                stat[R + threadIdx.x] = 1.0f / sqrtf(t * (1.0f / (float)D) + p.ln_eps);
            }
            __syncthreads();
            // ---- (d) LayerNorm(x_mid) -> the image (rows past the chunk: zeros, as the plain prologue leaves them)
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
                        o = make_uint2(h16<F>::pack2((pa[rb][nb][0] - mean_r[rb]) * rstd * g.x + b.x, (pa[rb][nb][1] - mean_r[rb]) * rstd * g.y + b.y),
                                       h16<F>::pack2((pa[rb][nb][2] - mean_r[rb]) * rstd * g.z + b.z, (pa[rb][nb][3] - mean_r[rb]) * rstd * g.w + b.w));
                    *reinterpret_cast<uint2 *>(h2 + lr * HP + 2 * n) = o;
                }
            }
        } else
        // ---- LayerNorm of the chunk's rows -> H2 (bf16): 16 threads per row, 32 rows per pass; rows past the chunk are zeros
        {
            const int r = threadIdx.x >> 4, jj = threadIdx.x & 15;
            for (int pass = 0; pass < (R + 31) / 32; ++pass) {
                const int lr = pass * 32 + r;
                if (lr < R) {
                    unsigned char *dst = h2 + lr * HP;
                    if (lr < nrow) {
                        const float *src = p.x + (size_t)(row0 + lr) * D;
                        float4 xf[D / 64];
#pragma unroll
                        for (int i = 0; i < D / 64; ++i) xf[i] = *reinterpret_cast<const float4 *>(src + 4 * (jj + 16 * i));
                        float s = 0.f;
#pragma unroll
                        for (int i = 0; i < D / 64; ++i) s += (xf[i].x + xf[i].y) + (xf[i].z + xf[i].w);
                        const float mean = row16_sum(s) * (1.0f / (float)D);
                        float q = 0.f;
#pragma unroll
                        for (int i = 0; i < D / 64; ++i) {
                            const float d0 = xf[i].x - mean, d1 = xf[i].y - mean, d2 = xf[i].z - mean, d3 = xf[i].w - mean;
                            q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
                        }
                        const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / (float)D) + p.ln_eps);
#pragma unroll
                        for (int i = 0; i < D / 64; ++i) {
                            const int cc = 4 * (jj + 16 * i);
                            const float4 g = *reinterpret_cast<const float4 *>(gam + cc), b = *reinterpret_cast<const float4 *>(bet + cc);
                            const float o0 = (xf[i].x - mean) * rstd * g.x + b.x, o1 = (xf[i].y - mean) * rstd * g.y + b.y;
                            const float o2 = (xf[i].z - mean) * rstd * g.z + b.z, o3 = (xf[i].w - mean) * rstd * g.w + b.w;
                            *reinterpret_cast<uint2 *>(dst + 2 * cc) = make_uint2(h16<F>::pack2(o0, o1), h16<F>::pack2(o2, o3));
                        }
                    } else {
                        // (row16_sum needs the whole DPP row: rows are per 16 lanes, so a missing row skips it as a unit)
#pragma unroll
                        for (int i = 0; i < D / 64; ++i) *reinterpret_cast<uint2 *>(dst + 8 * (jj + 16 * i)) = make_uint2(0u, 0u);
                    }
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        load_b1(0);                                                      // (behind the LayerNorm: in front of it the fragment registers made it spill)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        f32x4_t acc2[RB][3];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) acc2[rb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        // The weights of the NEXT use are fetched inside the MFMA loop of the CURRENT phase, one load per MFMA group: issued
        // as a burst after the loop (first version) the twelve loads stalled the wave ~2 500 cycles at the issue stage while
        // the matrix pipe sat idle (in-kernel stamps).  jb >= 0: load B2 fragments of slab jb (gemm1) / B1 of slab jb (gemm2).
        auto gemm1 = [&](int j, int jb) {                               // -> slab j in ub[j & 1]
            f32x4_t a1[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a1[rb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const unsigned char *ha = h2 + l15 * HP + 16 * kg;
#pragma unroll
            for (int s = 0; s < D / 32; ++s) {
                if (jb >= 0) b2f[s / 4][s % 4] = frag(W2, jb, s);       // 12 k-steps <-> the 12 fragments of B2: (nb, ks) = (s / 4, s % 4)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const uint4 f = *reinterpret_cast<const uint4 *>(ha + rb * 16 * HP + 64 * s);
                    a1[rb] = h16<F>::mfma16(b1f[s], f, a1[rb]);
                }
            }
            MLP_STAMP(j - 1, 4);                                         // (diagnostic build: end of the MFMA loop, start of the GELU)
            const float4 bv = *reinterpret_cast<const float4 *>(b1s + j * HC + 16 * w + 4 * kg);
            unsigned char *ud = ub + (j & 1) * U_BYTES + l15 * UP + (16 * w + 4 * kg) * 2;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float v[4] = {a1[rb][0] + bv.x, a1[rb][1] + bv.y, a1[rb][2] + bv.z, a1[rb][3] + bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = gelu_poly(v[i]);
                *reinterpret_cast<uint2 *>(ud + rb * 16 * UP) = make_uint2(h16<F>::pack2(v[0], v[1]), h16<F>::pack2(v[2], v[3]));
            }
        };
        auto gemm2 = [&](int j, int jb) {                               // acc2 += slab j . W2[:, slab]^T
            const unsigned char *ua = ub + (j & 1) * U_BYTES + l15 * UP + 16 * kg;
#pragma unroll
            for (int s = 0; s < HC / 32; ++s)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    // 4 x 5 MFMA groups <-> the 12 fragments of B1 (the first three row blocks of every k-step carry one)
                    if (jb >= 0 && rb < 3) b1f[3 * s + rb] = frag(W1, jb, 3 * s + rb);
                    const uint4 f = *reinterpret_cast<const uint4 *>(ua + rb * 16 * UP + 64 * s);
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) acc2[rb][nb] = h16<F>::mfma16(b2f[nb][s], f, acc2[rb][nb]);
                }
        };

        // (Tried and dropped: (a) running waves 4-7 half an iteration out of phase so that one half's GELU sits beside the other
        // half's MFMAs -- as two code paths or as a two-trip loop selecting the phase by (trip ^ half), hipcc's register allocation
        // went from ~230 VGPRs to 130-180 spilled ones; (b) producer / consumer roles (4 waves GEMM1 + GELU on 32 hidden units,
        // 4 waves GEMM2 on 96 columns) to halve the LDS fragment traffic -- the consumer's 120 accumulator + 96 weight registers
        // spill 80.  What remains the bound here is that LDS traffic: every A fragment of GEMM1 feeds a single MFMA.)
        MLP_STAMP(12, 1);
        gemm1(0, 0);                                                     // (fetches B2(0) under it)
        __syncthreads();
        MLP_STAMP(12, 2);
        for (int j = 0; j < NJ; ++j) {
            MLP_STAMP(j, 0);
            gemm2(j, j + 1 < NJ ? j + 1 : -1);                           // reads ub[j & 1] (complete at the barrier above); fetches B1(j + 1)
            MLP_STAMP(j, 1);
            if (j + 1 < NJ) gemm1(j + 1, j + 1);                         // writes ub[(j + 1) & 1] (last read by gemm2(j - 1), before that barrier); fetches B2(j + 1)
            MLP_STAMP(j, 2);
            __syncthreads();
            MLP_STAMP(j, 3);
        }
        MLP_STAMP(13, 0);

        // ---- epilogue: x <- x + drop_path * (acc2 + b2) (+ pos); lane holds columns 48 w + 16 nb + 4 kg .. + 3 of row 16 rb + l15.
        // The residual rows are all requested first (the 96 weight registers are free now): consumed where they are issued,
        // every 16 x 16 block paid a memory round trip (12 600 cycles for the 15 blocks of a chunk).
        {
            const float *xres = p.proj_a ? (const float *)p.out : p.x;       // (fused: x_mid, written by these same lanes in (c))
            float4 res[RB][3];
            float rs[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = row0 + min(16 * rb + l15, nrow - 1);
                rs[rb] = p.row_scale ? p.row_scale[m / p.row_scale_rows] : 1.0f;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) res[rb][nb] = *reinterpret_cast<const float4 *>(xres + (size_t)m * D + 48 * w + 16 * nb + 4 * kg);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int lr = 16 * rb + l15;
                if (lr < nrow) {
                    const int m = row0 + lr;
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) {
                        const int n = 48 * w + 16 * nb + 4 * kg;
                        const float4 bv = *reinterpret_cast<const float4 *>(b2s + n);
                        float4 o = make_float4((acc2[rb][nb][0] + bv.x) * rs[rb] + res[rb][nb].x, (acc2[rb][nb][1] + bv.y) * rs[rb] + res[rb][nb].y,
                                               (acc2[rb][nb][2] + bv.z) * rs[rb] + res[rb][nb].z, (acc2[rb][nb][3] + bv.w) * rs[rb] + res[rb][nb].w);
#ifndef PPT_MLP_STAMP
                        if (p.residual2) {
                            const float4 r2 = *reinterpret_cast<const float4 *>(p.residual2 + (size_t)m * D + n);
                            o.x += r2.x; o.y += r2.y; o.z += r2.z; o.w += r2.w;
                        }
#endif
                        *reinterpret_cast<float4 *>(p.out + (size_t)m * D + n) = o;
                    }
                }
            }
        }
        MLP_STAMP(13, 1);
    }
}

// fragment order of the two weights (see frag() above): thread -> one 16-byte piece
//   W1t[j][w][f][lane] = W1[128 j + 16 w + l15][32 f + 8 kg .. + 8)                       (wave w, k-step f)
//   W2t[j][w][f = 4 nb + ks][lane] = W2[48 w + 16 nb + l15][128 j + 32 ks + 8 kg .. + 8)   (wave w, column block nb, k-step ks)
__global__ __launch_bounds__(256) void vit_mlp_retile_kernel(const bf16_t *__restrict__ W1, const bf16_t *__restrict__ W2,
                                                             bf16_t *__restrict__ W1t, bf16_t *__restrict__ W2t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                        // over NJ * 8 * 12 * 64 pieces
    if (i >= NJ * 8 * 12 * 64) return;
    const int lane = i & 63, f = (i >> 6) % 12, w = (i / (64 * 12)) & 7, j = i / (64 * 12 * 8);
    const int l15 = lane & 15, kg = lane >> 4;
    const uint4 a = *reinterpret_cast<const uint4 *>(W1 + (size_t)(j * HC + 16 * w + l15) * D + 32 * f + 8 * kg);
    const uint4 b = *reinterpret_cast<const uint4 *>(W2 + (size_t)(48 * w + 16 * (f / 4) + l15) * HID + j * HC + 32 * (f % 4) + 8 * kg);
    *reinterpret_cast<uint4 *>(W1t + (size_t)i * 8) = a;
    *reinterpret_cast<uint4 *>(W2t + (size_t)i * 8) = b;
}

// fragment order of the proj weight (see pfrag() in the kernel): Wpt[w][f = 12 nb + ks][lane] = Wp[48 w + 16 nb + l15][32 ks + 8 kg .. + 8)
__global__ __launch_bounds__(256) void vit_proj_retile_kernel(const bf16_t *__restrict__ Wp, bf16_t *__restrict__ Wpt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                        // over 8 * 36 * 64 pieces
    if (i >= 8 * 36 * 64) return;
    const int lane = i & 63, f = (i >> 6) % 36, w = i / (64 * 36);
    const int l15 = lane & 15, kg = lane >> 4, nb = f / 12, ks = f % 12;
    *reinterpret_cast<uint4 *>(Wpt + (size_t)i * 8) = *reinterpret_cast<const uint4 *>(Wp + (size_t)(48 * w + 16 * nb + l15) * D + 32 * ks + 8 * kg);
}

}  // namespace

extern "C" int ppt_vit_proj_retile(const void *Wp, void *Wp_tiled, void *stream)
{
    if (!Wp || !Wp_tiled || (((uintptr_t)Wp | (uintptr_t)Wp_tiled) & 15)) return PPT_EINVAL;
    hipLaunchKernelGGL(vit_proj_retile_kernel, dim3((8 * 36 * 64 + 255) / 256), dim3(256), 0, ppt_stream(stream), (const bf16_t *)Wp, (bf16_t *)Wp_tiled);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_vit_mlp_retile(const void *W1, const void *W2, void *W1t, void *W2t, void *stream)
{
    if (!W1 || !W2 || !W1t || !W2t || (((uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)W1t | (uintptr_t)W2t) & 15)) return PPT_EINVAL;
    hipLaunchKernelGGL(vit_mlp_retile_kernel, dim3((NJ * 8 * 12 * 64 + 255) / 256), dim3(256), 0, ppt_stream(stream), (const bf16_t *)W1,
                       (const bf16_t *)W2, (bf16_t *)W1t, (bf16_t *)W2t);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_vit_mlp_bf16(const ppt_vit_mlp_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_vit_mlp_params p = *pp;
    if (!p.x || !p.out || !p.W1 || !p.W2 || !p.ln_w || !p.ln_b || p.M <= 0) return PPT_EINVAL;
    if (p.D != D || p.hidden != HID) return PPT_EUNSUPPORTED;
    if (p.dtype != PPT_BF16 && p.dtype != PPT_F16 && p.dtype != 0) return PPT_EINVAL;          // (0: callers of ABI 2 -- bf16)
    if (p.row_scale && p.row_scale_rows <= 0) return PPT_EINVAL;
    if (((uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.W1 | (uintptr_t)p.W2 | (uintptr_t)p.residual2) & 15) return PPT_EINVAL;
    if (p.proj_a) {                                                      // the fused attention-branch tail (see the file header)
        if (!p.proj_W || (p.proj_row_scale && p.proj_row_scale_rows <= 0)) return PPT_EINVAL;
        if (((uintptr_t)p.proj_a | (uintptr_t)p.proj_W | (uintptr_t)p.proj_b) & 15) return PPT_EINVAL;
    }
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)vit_mlp_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)vit_mlp_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return 0;
    }();
    (void)attrs_once;
    const int cus = ppt_cu_count(ppt_stream(stream));            // (of the stream's device, not process-global state)
    // chunks of at most R rows, a whole number of rounds over the CUs, every chunk as full as the division allows
    int wgs = p.workgroups > 0 ? p.workgroups : cus;
    int rounds = 1;
    while ((int64_t)rounds * wgs * R < p.M) ++rounds;
    // the caller leaves room for the other stream (ppt_set_persistent_occupancy < 100): as FEW workgroups as the same number of
    // rounds allows, i.e. full R-row chunks (C2: 206 workgroups of 80 rows instead of 253 of 65 -- same time, 47 CUs free)
    if (p.workgroups <= 0 && ppt_get_persistent_occupancy() < 100) wgs = (int)((p.M + (int64_t)rounds * R - 1) / ((int64_t)rounds * R));
    p.n_chunks = rounds * wgs;
    p.rows_per_chunk = (p.M + p.n_chunks - 1) / p.n_chunks;
    p.n_chunks = (p.M + p.rows_per_chunk - 1) / p.rows_per_chunk;
    const int grid = p.n_chunks < wgs ? p.n_chunks : wgs;
    if (p.dtype == PPT_F16) hipLaunchKernelGGL(vit_mlp_kernel<f16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), p);
    else hipLaunchKernelGGL(vit_mlp_kernel<bf16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
