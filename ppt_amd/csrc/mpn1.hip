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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int MPN_C1 = 128, MPN_N = 256, MPN_PITCH = 144;      // LDS row pitch of the 32 x 64 bf16 transpose tile (128 + 16 B)

__global__ __launch_bounds__(256) void mpn1_kernel(const float *__restrict__ pts, int n_tiles, const float *__restrict__ w1,
                                                    const float *__restrict__ b1, const float *__restrict__ a_scale,
                                                    const float *__restrict__ a_shift, const bf16_t *__restrict__ W2,
                                                    const float *__restrict__ bias2, bf16_t *__restrict__ y2,
                                                    bf16_t *__restrict__ gmax)
{
    __shared__ float4 tab[MPN_C1];
    __shared__ __align__(16) unsigned char tr_all[4][32 * MPN_PITCH];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int c = threadIdx.x; c < MPN_C1; c += 256) {                  // as gemm.hip's PPT_A_CONV1 table
        const float s = a_scale[c], h = a_shift[c];
        tab[c] = make_float4(s * w1[c * 3 + 0], s * w1[c * 3 + 1], s * w1[c * 3 + 2], fmaf(s, b1[c], h));
    }
    const int col = lane & 31, h = lane >> 5;
    bf16x8_t bfrag[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            bfrag[j][s] = *reinterpret_cast<const bf16x8_t *>(W2 + (size_t)(64 * w + 32 * j + col) * MPN_C1 + 16 * s + 8 * h);
    const float bias[2] = {bias2[64 * w + col], bias2[64 * w + 32 + col]};
    unsigned char *tr = tr_all[w];
    __syncthreads();

    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const float *pp = pts + ((size_t)t * 32 + col) * 3;
        const float x = pp[0], y = pp[1], z = pp[2];
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint32_t pk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v0 = tab[16 * s + 8 * h + 2 * i], v1 = tab[16 * s + 8 * h + 2 * i + 1];
                const float f0 = fmaxf(fmaf(v0.z, z, fmaf(v0.y, y, fmaf(v0.x, x, v0.w))), 0.0f);
                const float f1 = fmaxf(fmaf(v1.z, z, fmaf(v1.y, y, fmaf(v1.x, x, v1.w))), 0.0f);
                pk[i] = pack_bf16x2(f0, f1);
            }
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[0][s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[1][s], acc[1], 0, 0, 0);
        }
        // C layout: column (lane & 31), rows (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[j][e] += bias[j];
                mx = fmaxf(mx, acc[j][e]);
            }
            mx = xor32_max(mx);
            if (h == 0) gmax[(size_t)t * MPN_N + 64 * w + 32 * j + col] = f32_to_bf16(mx);
            // neighbour lanes trade one value per register pair, so that a lane owns two adjacent columns of one row:
            // even lanes keep row(e0), odd lanes row(e1) -- 4-byte LDS writes instead of 2-byte ones
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e0 = 2 * q, e1 = 2 * q + 1;
                const float send = (lane & 1) ? acc[j][e0] : acc[j][e1];
                const float recv = __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(send)));     // quad_perm [1,0,3,2]
                const uint32_t packed = (lane & 1) ? pack_bf16x2(recv, acc[j][e1]) : pack_bf16x2(acc[j][e0], recv);
                const int e = (lane & 1) ? e1 : e0;
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                *reinterpret_cast<uint32_t *>(tr + row * MPN_PITCH + (32 * j + (col & ~1)) * 2) = packed;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + (lane >> 3), ch = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4 *>(tr + row * MPN_PITCH + ch * 16);
            *reinterpret_cast<uint4 *>(y2 + ((size_t)t * 32 + row) * MPN_N + 64 * w + ch * 8) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int ppt_mini_pointnet_conv12_bf16(const float *pts, int64_t M, const float *w1, const float *b1, const float *a_scale,
                                             const float *a_shift, int C1, const void *W2, const float *bias2, int N, void *y2,
                                             void *gmax, void *stream)
{
    if (!pts || !w1 || !b1 || !a_scale || !a_shift || !W2 || !bias2 || !y2 || !gmax || M <= 0) return PPT_EINVAL;
    if (C1 != MPN_C1 || N != MPN_N || M % 32) return PPT_EUNSUPPORTED;
    if (((uintptr_t)W2 | (uintptr_t)y2) & 15) return PPT_EINVAL;
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int64_t tiles = M / 32;
    const int grid = (int)(tiles < (int64_t)cus * 3 ? tiles : (int64_t)cus * 3);
    hipLaunchKernelGGL(mpn1_kernel, dim3(grid), dim3(256), 0, ppt_stream(stream), pts, (int)tiles, w1, b1, a_scale, a_shift,
                       (const bf16_t *)W2, bias2, (bf16_t *)y2, (bf16_t *)gmax);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
