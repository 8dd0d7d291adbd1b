// mpn1.hip -- first half of the PointBERT mini-PointNet (Encoder.first_conv, dvae.py:188-193,206-210) as one kernel:
//     pts [M,3] -> Conv1d(3,128) -> BatchNorm1d (folded scale/shift) -> ReLU -> Conv1d(128,256) + bias -> y2 [M,256] bf16
//                                                                        \-> max over each group of 32 rows -> gmax [M/32,256]
// The generic GEMM runs this as an A-prologue (ppt_gemm, PPT_A_CONV1) on 128 x 128 tiles with K = 128: four K slabs per
// tile, so each workgroup spends its time in prologue / barrier / epilogue latency (326 us for M = 524 288: 105 TFLOP/s
// on a product whose only HBM traffic is the 268 MB it writes).  Here nothing is staged and nothing synchronises:
//   * a wave owns 64 output columns for the whole kernel; its B operand -- W2[64 cols][128] -- sits in 64 VGPRs;
//   * one group of 32 points is one MFMA row tile: lane (row, half) computes its own A fragment from the point's three
//     coordinates and a 2 KB LDS table {scale*w1, scale*b1 + shift} -- the same expression, in the same order, as the
//     generic prologue, so y2 and gmax are bit-identical to that path;
//   * 16 MFMA 32x32x16 per group, bias, group max out of the accumulators (one permlane swap), y2 through a wave-private
//     LDS transpose so that rows leave as 128-byte pieces.
// Bound: the y2 write (512 B per point, ~70 us at 4 TB/s); the A fragments are recomputed by the four waves of a group
// (VALU ~45 us per SIMD), which is what buys the absence of any barrier.
#include "ppt_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// C1: channels of the K=3 conv (= K of the MFMA product); a wave owns TJ column tiles of 32; NWN waves side by side cover
// N = 32 * TJ * NWN columns, the 4 / NWN wave rows of a workgroup take different groups of 32 points.
// POOL: gmax[g, :] = max over the 32 rows of the group.  STATS: BatchNorm partials of the output per 32-row chunk,
// (sum, M2 = sum (v - chunk mean)^2), the layout ppt_bn_finalize_ws takes with rows_per_partial = 32.
template <typename F, int C1, int TJ, int NWN, bool POOL, bool STATS>
__global__ __launch_bounds__(256) void mpn1_kernel(const float *__restrict__ pts, int n_tiles, const float *__restrict__ w1,
                                                    const float *__restrict__ b1, const float *__restrict__ a_scale,
                                                    const float *__restrict__ a_shift, const bf16_t *__restrict__ W2,
                                                    const float *__restrict__ bias2, bf16_t *__restrict__ y2,
                                                    bf16_t *__restrict__ gmax, float *__restrict__ part_sum,
                                                    float *__restrict__ part_m2)
{
    constexpr int N = 32 * TJ * NWN, KS = C1 / 16, NWM = 4 / NWN;
    constexpr int PITCH = 64 * TJ + 16;                   // LDS row pitch of a wave's 32 x (32 TJ) bf16 transpose tile
    __shared__ float4 tab[C1];
    __shared__ __align__(16) unsigned char tr_all[4][32 * PITCH];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = w % NWN, wm = w / NWN;
    for (int c = threadIdx.x; c < C1; c += 256) {                     // as gemm.hip's PPT_A_CONV1 table
        const float s = a_scale[c], h = a_shift[c];
        tab[c] = make_float4(s * w1[c * 3 + 0], s * w1[c * 3 + 1], s * w1[c * 3 + 2], fmaf(s, b1[c], h));
    }
    const int col = lane & 31, h = lane >> 5;
    const int n_w = 32 * TJ * wn;                                      // first column of this wave
    uint4 bfrag[TJ][KS];
    float bias[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            bfrag[j][s] = *reinterpret_cast<const uint4 *>(W2 + (size_t)(n_w + 32 * j + col) * C1 + 16 * s + 8 * h);
        bias[j] = bias2 ? bias2[n_w + 32 * j + col] : 0.f;
    }
    unsigned char *tr = tr_all[w];
    __syncthreads();

    for (int t = blockIdx.x * NWM + wm; t < n_tiles; t += gridDim.x * NWM) {
        const float *pp = pts + ((size_t)t * 32 + col) * 3;
        const float x = pp[0], y = pp[1], z = pp[2];
        f32x16_t acc[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint32_t pk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v0 = tab[16 * s + 8 * h + 2 * i], v1 = tab[16 * s + 8 * h + 2 * i + 1];
                const float f0 = fmaxf(fmaf(v0.z, z, fmaf(v0.y, y, fmaf(v0.x, x, v0.w))), 0.0f);
                const float f1 = fmaxf(fmaf(v1.z, z, fmaf(v1.y, y, fmaf(v1.x, x, v1.w))), 0.0f);
                pk[i] = h16<F>::pack2(f0, f1);
            }
            const uint4 a = (make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[j] = h16<F>::mfma32(a, bfrag[j][s], acc[j]);
        }
        // C layout: column (lane & 31), rows (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            float mx = -INFINITY, sm = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[j][e] += bias[j];
                mx = fmaxf(mx, acc[j][e]);
                sm += acc[j][e];
            }
            if constexpr (POOL) {
                mx = xor32_max(mx);
                if (h == 0) gmax[(size_t)t * N + n_w + 32 * j + col] = h16<F>::from_f32(mx);
            }
            if constexpr (STATS) {
                sm = xor32_sum(sm);
                const float mean = sm * (1.0f / 32.0f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = acc[j][e] - mean; q = fmaf(d, d, q); }
                q = xor32_sum(q);
                if (h == 0) {
                    part_sum[(size_t)t * N + n_w + 32 * j + col] = sm;
                    part_m2[(size_t)t * N + n_w + 32 * j + col] = q;
                }
            }
            // neighbour lanes trade one value per register pair, so that a lane owns two adjacent columns of one row:
            // even lanes keep row(e0), odd lanes row(e1) -- 4-byte LDS writes instead of 2-byte ones
#pragma unroll
            for (int q2 = 0; q2 < 8; ++q2) {
                const int e0 = 2 * q2, e1 = 2 * q2 + 1;
                const float send = (lane & 1) ? acc[j][e0] : acc[j][e1];
                const float recv = __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(send)));     // quad_perm [1,0,3,2]
                const uint32_t packed = (lane & 1) ? h16<F>::pack2(recv, acc[j][e1]) : h16<F>::pack2(acc[j][e0], recv);
                const int e = (lane & 1) ? e1 : e0;
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                *reinterpret_cast<uint32_t *>(tr + row * PITCH + (32 * j + (col & ~1)) * 2) = packed;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // rows leave as 64 TJ-byte pieces: 4 TJ lanes x 16 bytes per row
        constexpr int LPR = 4 * TJ, RPI = 64 / LPR, NIT = (32 + RPI - 1) / RPI;      // lanes per row, rows per instruction
#pragma unroll
        for (int q2 = 0; q2 < NIT; ++q2) {
            const int row = RPI * q2 + lane / LPR, ch = lane % LPR;
            if (row < 32 && lane < RPI * LPR) {
                const uint4 v = *reinterpret_cast<const uint4 *>(tr + row * PITCH + ch * 16);
                ppt_store16_stream(y2 + ((size_t)t * 32 + row) * N + n_w + ch * 8, v);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

int mpn1_grid(int64_t tiles, int nwm, hipStream_t st)
{
    const int cus = ppt_cu_count(st);            // (of the stream's device, not process-global state)
    const int64_t blocks = (tiles + nwm - 1) / nwm;
    int64_t want = (int64_t)cus * 3 * ppt_get_persistent_occupancy() / 100;        // (ppt_set_persistent_occupancy)
    want = want < 8 ? 8 : want;
    return (int)(blocks < want ? blocks : want);
}

}  // namespace

extern "C" int ppt_mini_pointnet_conv12_half(const float *pts, int64_t M, const float *w1, const float *b1, const float *a_scale,
                                             const float *a_shift, int C1, const void *W2, const float *bias2, int N, void *y2,
                                             void *gmax, int dtype, void *stream)
{
    if (dtype != PPT_BF16 && dtype != PPT_F16) return PPT_EINVAL;
    if (!pts || !w1 || !b1 || !a_scale || !a_shift || !W2 || !bias2 || !y2 || !gmax || M <= 0) return PPT_EINVAL;
    if (C1 != 128 || N != 256 || M % 32) return PPT_EUNSUPPORTED;
    if (((uintptr_t)W2 | (uintptr_t)y2) & 15) return PPT_EINVAL;
    const int64_t tiles = M / 32;
    if (dtype == PPT_F16)
        hipLaunchKernelGGL((mpn1_kernel<f16_t, 128, 2, 4, true, false>), dim3(mpn1_grid(tiles, 1, ppt_stream(stream))), dim3(256), 0, ppt_stream(stream), pts,
                           (int)tiles, w1, b1, a_scale, a_shift, (const bf16_t *)W2, bias2, (bf16_t *)y2, (bf16_t *)gmax, nullptr, nullptr);
    else
        hipLaunchKernelGGL((mpn1_kernel<bf16_t, 128, 2, 4, true, false>), dim3(mpn1_grid(tiles, 1, ppt_stream(stream))), dim3(256), 0, ppt_stream(stream), pts,
                           (int)tiles, w1, b1, a_scale, a_shift, (const bf16_t *)W2, bias2, (bf16_t *)y2, (bf16_t *)gmax, nullptr, nullptr);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_mini_pointnet_conv12_bf16(const float *pts, int64_t M, const float *w1, const float *b1, const float *a_scale,
                                             const float *a_shift, int C1, const void *W2, const float *bias2, int N, void *y2,
                                             void *gmax, void *stream)
{
    return ppt_mini_pointnet_conv12_half(pts, M, w1, b1, a_scale, a_shift, C1, W2, bias2, N, y2, gmax, PPT_BF16, stream);
}

// the same product with BatchNorm partials of the output instead of the group max: the first two convs of a PointNet2
// set-abstraction branch on raw coordinates (pointnet2_utils.py:168-199, 217-262 with in_channel = 0): C1 in {32, 64},
// N in {32, 64, 96}; part_sum / part_m2 [M/32, N] f32 (rows_per_partial = 32 for ppt_bn_finalize_ws); bias2 may be null.
extern "C" int ppt_conv12_stats_bf16(const float *pts, int64_t M, const float *w1, const float *b1, const float *a_scale,
                                     const float *a_shift, int C1, const void *W2, const float *bias2, int N, void *y2,
                                     float *part_sum, float *part_m2, void *stream)
{
    if (!pts || !w1 || !b1 || !a_scale || !a_shift || !W2 || !y2 || !part_sum || !part_m2 || M <= 0) return PPT_EINVAL;
    if (M % 32 || (((uintptr_t)W2 | (uintptr_t)y2) & 15)) return PPT_EUNSUPPORTED;
    const int64_t tiles = M / 32;
    hipStream_t s = ppt_stream(stream);
#define MPN_LAUNCH(C, TJ_)                                                                                                 \
    hipLaunchKernelGGL((mpn1_kernel<bf16_t, C, TJ_, 1, false, true>), dim3(mpn1_grid(tiles, 4, s)), dim3(256), 0, s, pts, (int)tiles, w1, b1, \
                       a_scale, a_shift, (const bf16_t *)W2, bias2, (bf16_t *)y2, nullptr, part_sum, part_m2)
    if (C1 == 32 && N == 32) MPN_LAUNCH(32, 1);
    else if (C1 == 64 && N == 64) MPN_LAUNCH(64, 2);
    else if (C1 == 64 && N == 96) MPN_LAUNCH(64, 3);
    else if (C1 == 64 && N == 128) MPN_LAUNCH(64, 4);
    else if (C1 == 128 && N == 128) MPN_LAUNCH(128, 4);
    else return PPT_EUNSUPPORTED;
#undef MPN_LAUNCH
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
