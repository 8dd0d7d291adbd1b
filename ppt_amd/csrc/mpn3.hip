// mpn3.hip -- the middle conv of the PointBERT mini-PointNet (Encoder.second_conv[0] on cat(global, local), dvae.py:194-195,
// 211-212), in the split form engine.mini_pointnet uses:  y3[m, :] = W3b . y2[m, :] + gterm[m / 32, :]  (gterm = the global
// half of the conv per group, bias included), y2 [M,256] bf16, W3b [512,256] bf16, y3 [M,512] bf16, plus the BatchNorm
// partials of y3 per 32-row chunk.  HBM-bound on its 537 MB of output + 268 MB of input; ppt_gemm's 128 x 128 tile loop
// needs 311 us for it.  Same scheme as mpn4.hip: 8 waves, wave w keeps columns 64 w .. 64 w + 63 of W3b in 128 VGPRs for
// the whole kernel; one group of 32 points per step: its 16 KB go to LDS once (double-buffered, one barrier per group),
// every wave reads its A fragments from there, adds the group term, forms the chunk statistics in registers and sends its
// 32 x 64 bf16 block out through a wave-private LDS transpose as 128-byte row pieces.  Same k order as the generic path:
// y3 is bit-identical.
// Round 4: y == NULL -- the statistics pass alone (STORE = false): the BatchNorm partials of y3 without y3 itself, for the
// training step whose second pass is the fused conv3 + BN + ReLU + conv4 + max kernel (csrc/mpn34.hip).
#include "ppt_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int M3_K = 256, M3_N = 512, M3_KS = M3_K / 16, M3_PITCH = 2 * M3_K + 16, M3_BUF = 32 * M3_PITCH;
constexpr int M3_TP = 64 * 2 + 16, M3_TR = 32 * M3_TP;             // a wave's 32 x 64 bf16 transpose tile

template <typename F, bool STATS, bool STORE>
__global__ __launch_bounds__(512, 2) void mpn3_kernel(const bf16_t *__restrict__ A, int n_tiles, const bf16_t *__restrict__ W,
                                                       const float *__restrict__ gterm, bf16_t *__restrict__ y,
                                                       float *__restrict__ part_sum, float *__restrict__ part_m2)
{
    extern __shared__ __align__(16) unsigned char smem[];          // 2 A buffers, then 8 transpose tiles
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, h = lane >> 5;
    uint4 bfrag[2][M3_KS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < M3_KS; ++s)
            bfrag[j][s] = *reinterpret_cast<const uint4 *>(W + (size_t)(64 * w + 32 * j + col) * M3_K + 16 * s + 8 * h);
    unsigned char *tr = smem + 2 * M3_BUF + w * M3_TR;
    const int cc = threadIdx.x & 31, rb = threadIdx.x >> 5;         // loader: 16-byte chunk cc of rows rb and rb + 16
    uint4 v0, v1;                                                    // (named values, not an array: an array that lives across
                                                                     //  the wavefront fences below is kept in scratch memory)
#define M3_LOAD(tile)                                                                                                   \
    v0 = *reinterpret_cast<const uint4 *>(A + ((size_t)(tile) * 32 + rb) * M3_K + 8 * cc);                              \
    v1 = *reinterpret_cast<const uint4 *>(A + ((size_t)(tile) * 32 + rb + 16) * M3_K + 8 * cc);
#define M3_STAGE(buf)                                                                                                   \
    *reinterpret_cast<uint4 *>(smem + (buf) * M3_BUF + rb * M3_PITCH + cc * 16) = v0;                                   \
    *reinterpret_cast<uint4 *>(smem + (buf) * M3_BUF + (rb + 16) * M3_PITCH + cc * 16) = v1;
    int t = blockIdx.x;
    if (t >= n_tiles) return;
    M3_LOAD(t);
    M3_STAGE(0);
    __syncthreads();
    for (int it = 0; t < n_tiles; t += gridDim.x, ++it) {
        const int cur = it & 1;
        const int tn = min(t + (int)gridDim.x, n_tiles - 1);
        M3_LOAD(tn);
        const unsigned char *at = smem + cur * M3_BUF + col * M3_PITCH + 16 * h;
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < M3_KS; ++s) {
            const uint4 a = *reinterpret_cast<const uint4 *>(at + 32 * s);
            acc[0] = h16<F>::mfma32(a, bfrag[0][s], acc[0]);
            acc[1] = h16<F>::mfma32(a, bfrag[1][s], acc[1]);
        }
        // C layout: column (lane & 31), rows (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 64 * w + 32 * j + col;
            const float gt = gterm[(size_t)t * M3_N + n];
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[j][e] += gt;
                sm += acc[j][e];
            }
            if constexpr (STATS) {
                sm = xor32_sum(sm);
                const float mean = sm * (1.0f / 32.0f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = acc[j][e] - mean; q = fmaf(d, d, q); }
                q = xor32_sum(q);
                if (h == 0) {
                    part_sum[(size_t)t * M3_N + n] = sm;
                    part_m2[(size_t)t * M3_N + n] = q;
                }
            }
            if constexpr (STORE)
#pragma unroll
            for (int q2 = 0; q2 < 8; ++q2) {                        // neighbour lanes trade values: 4-byte LDS writes (mpn1.hip)
                const int e0 = 2 * q2, e1 = 2 * q2 + 1;
                const float send = (lane & 1) ? acc[j][e0] : acc[j][e1];
                const float recv = __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(send)));
                const uint32_t packed = (lane & 1) ? h16<F>::pack2(recv, acc[j][e1]) : h16<F>::pack2(acc[j][e0], recv);
                const int e = (lane & 1) ? e1 : e0;
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                *reinterpret_cast<uint32_t *>(tr + row * M3_TP + (32 * j + (col & ~1)) * 2) = packed;
            }
        }
        if constexpr (STORE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const int row = 8 * q2 + (lane >> 3), ch = lane & 7;
                const uint4 o = *reinterpret_cast<const uint4 *>(tr + row * M3_TP + ch * 16);
                ppt_store16_stream(y + ((size_t)t * 32 + row) * M3_N + 64 * w + ch * 8, o);
            }
        }
        M3_STAGE(cur ^ 1);                                          // last read in iteration it - 1, before its barrier
        __syncthreads();
    }
#undef M3_LOAD
#undef M3_STAGE
}

}  // namespace

extern "C" int ppt_mini_pointnet_conv3_half(const void *A, int64_t M, int K, const void *W, const float *gterm, int N, void *y,
                                            float *part_sum, float *part_m2, int dtype, void *stream)
{
    if (dtype != PPT_BF16 && dtype != PPT_F16) return PPT_EINVAL;
    if (!A || !W || !gterm || M <= 0 || ((part_sum == nullptr) != (part_m2 == nullptr)) || (!y && !part_sum)) return PPT_EINVAL;
    if (K != M3_K || N != M3_N || M % 32) return PPT_EUNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)y) & 15) return PPT_EINVAL;
    constexpr int lds = 2 * M3_BUF + 8 * M3_TR;
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)mpn3_kernel<bf16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)mpn3_kernel<bf16_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)mpn3_kernel<f16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)mpn3_kernel<f16_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)mpn3_kernel<bf16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)mpn3_kernel<f16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return 0;
    }();
    (void)attrs_once;
    const int cus = ppt_cu_count(ppt_stream(stream));            // (of the stream's device, not process-global state)
    const int64_t tiles = M / 32;
    // ONE persistent workgroup per CU (alone the kernel is HBM-bound and as fast as with two: C2 tower 2.765 vs 2.776 ms), fewer
    // when the caller leaves room for the other stream (ppt_set_persistent_occupancy)
    int64_t want = (int64_t)cus * ppt_get_persistent_occupancy() / 100;
    want = want < 8 ? 8 : want;
    const int grid = (int)(tiles < want ? tiles : want);
#define PPT_M3(FF, ST, SO) hipLaunchKernelGGL((mpn3_kernel<FF, ST, SO>), dim3(grid), dim3(512), lds, ppt_stream(stream), (const bf16_t *)A, (int)tiles, \
                                              (const bf16_t *)W, gterm, (bf16_t *)y, part_sum, part_m2)
    if (dtype == PPT_F16) { if (!y) PPT_M3(f16_t, true, false); else if (part_sum) PPT_M3(f16_t, true, true); else PPT_M3(f16_t, false, true); }
    else { if (!y) PPT_M3(bf16_t, true, false); else if (part_sum) PPT_M3(bf16_t, true, true); else PPT_M3(bf16_t, false, true); }
#undef PPT_M3
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_mini_pointnet_conv3_bf16(const void *A, int64_t M, int K, const void *W, const float *gterm, int N, void *y,
                                            float *part_sum, float *part_m2, void *stream)
{
    return ppt_mini_pointnet_conv3_half(A, M, K, W, gterm, N, y, part_sum, part_m2, PPT_BF16, stream);
}
