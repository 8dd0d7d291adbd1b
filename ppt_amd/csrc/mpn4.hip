// mpn4.hip -- the last conv of the PointBERT mini-PointNet with its group max (Encoder.second_conv[1:] + max, dvae.py:194-199,
// 213-214):  tok[g, :] = max over the 32 points of group g of  W4 . relu(scale * y3 + shift) + bias,  y3 [M,512] bf16 (the raw
// conv3 output whose folded BatchNorm + ReLU is applied while it is read), W4 [256,512] bf16; nothing but tok is written.
// ppt_gemm runs this on 128 x 128 tiles through its register-staged A-prologue loop: 305 us for M = 524 288 (16 384 groups),
// against 134 us that reading y3 once costs at 4 TB/s.  Here the B operand never moves: a workgroup is 8 waves, wave w keeps
// columns 32 w .. 32 w + 31 of W4 -- 32 k-steps x 16 bytes = 128 VGPRs -- for the whole kernel.  One group of 32 points is one
// MFMA row tile: the 512 threads load its 32 KB (each thread always the same 16-byte column chunk, so its 8 (scale, shift)
// pairs live in registers), apply the affine + ReLU once, and park the bf16 tile in LDS (row pitch 1040 B: the 16 lanes of a
// ds_read_b128 phase hit 64 distinct banks); every wave then reads its A fragments from there (32 reads, 32 MFMA) and takes
// the max over the rows out of its accumulator.  Two LDS buffers, one barrier per group; the next group's global loads are
// issued before the MFMA loop.  Same affine expression and k order as the generic path: bit-identical maxima.
#include "ppt_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int M4_K = 512, M4_N = 256, M4_KS = M4_K / 16, M4_PITCH = 2 * M4_K + 16, M4_BUF = 32 * M4_PITCH;

template <typename F>
__global__ __launch_bounds__(512, 2) void mpn4_kernel(const bf16_t *__restrict__ A, int n_tiles, const float *__restrict__ a_scale,
                                                       const float *__restrict__ a_shift, const bf16_t *__restrict__ W,
                                                       const float *__restrict__ bias, bf16_t *__restrict__ tok)
{
    extern __shared__ __align__(16) unsigned char smem[];              // 2 x 32 rows x 1040 B
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, h = lane >> 5;
    uint4 bfrag[M4_KS];
#pragma unroll
    for (int s = 0; s < M4_KS; ++s)
        bfrag[s] = *reinterpret_cast<const uint4 *>(W + (size_t)(32 * w + col) * M4_K + 16 * s + 8 * h);
    const float bias_v = bias ? bias[32 * w + col] : 0.f;
    // loader: thread -> 16-byte chunk cc of rows rb, rb + 8, rb + 16, rb + 24
    const int cc = threadIdx.x & 63, rb = threadIdx.x >> 6;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = a_scale[8 * cc + e]; sh[e] = a_shift[8 * cc + e]; }
    uint4 v[4];
#define M4_LOAD(tile)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                       \
        v[i] = *reinterpret_cast<const uint4 *>(A + ((size_t)(tile) * 32 + rb + 8 * i) * M4_K + 8 * cc);
#define M4_STAGE(buf)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                     \
        const uint32_t wv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};                                                        \
        uint32_t pk[4];                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                 \
            const float lo = fmaxf(fmaf(h16<F>::lo(wv[e]), sc[2 * e], sh[2 * e]), 0.0f);                     \
            const float hi = fmaxf(fmaf(h16<F>::hi(wv[e]), sc[2 * e + 1], sh[2 * e + 1]), 0.0f);     \
            pk[e] = h16<F>::pack2(lo, hi);                                                                                \
        }                                                                                                               \
        *reinterpret_cast<uint4 *>(smem + (buf) * M4_BUF + (rb + 8 * i) * M4_PITCH + cc * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]); \
    }
    int t = blockIdx.x;
    if (t >= n_tiles) return;
    M4_LOAD(t);
    M4_STAGE(0);
    __syncthreads();
    for (int it = 0; t < n_tiles; t += gridDim.x, ++it) {
        const int cur = it & 1;
        const int tn = min(t + (int)gridDim.x, n_tiles - 1);           // unconditional (a branch parks v[] in scratch)
        M4_LOAD(tn);
        const unsigned char *at = smem + cur * M4_BUF + col * M4_PITCH + 16 * h;
        f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < M4_KS; ++s) {
            const uint4 a = *reinterpret_cast<const uint4 *>(at + 32 * s);
            acc = h16<F>::mfma32(a, bfrag[s], acc);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, acc[e] + bias_v);
        mx = xor32_max(mx);
        if (h == 0) tok[(size_t)t * M4_N + 32 * w + col] = h16<F>::from_f32(mx);
        M4_STAGE(cur ^ 1);                                            // last read in iteration it - 1, before its barrier
        __syncthreads();
    }
#undef M4_LOAD
#undef M4_STAGE
}

}  // namespace

extern "C" int ppt_mini_pointnet_conv4_half(const void *A, int64_t M, int K, const float *a_scale, const float *a_shift, const void *W,
                                            const float *bias, int N, void *tok, int dtype, void *stream)
{
    if (dtype != PPT_BF16 && dtype != PPT_F16) return PPT_EINVAL;
    if (!A || !a_scale || !a_shift || !W || !tok || M <= 0) return PPT_EINVAL;
    if (K != M4_K || N != M4_N || M % 32) return PPT_EUNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)W) & 15) return PPT_EINVAL;
    constexpr int lds = 2 * M4_BUF;
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)mpn4_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)mpn4_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
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
    if (dtype == PPT_F16)
        hipLaunchKernelGGL(mpn4_kernel<f16_t>, dim3(grid), dim3(512), lds, ppt_stream(stream), (const bf16_t *)A, (int)tiles, a_scale, a_shift,
                           (const bf16_t *)W, bias, (bf16_t *)tok);
    else
        hipLaunchKernelGGL(mpn4_kernel<bf16_t>, dim3(grid), dim3(512), lds, ppt_stream(stream), (const bf16_t *)A, (int)tiles, a_scale, a_shift,
                           (const bf16_t *)W, bias, (bf16_t *)tok);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_mini_pointnet_conv4_bf16(const void *A, int64_t M, int K, const float *a_scale, const float *a_shift, const void *W,
                                            const float *bias, int N, void *tok, void *stream)
{
    return ppt_mini_pointnet_conv4_half(A, M, K, a_scale, a_shift, W, bias, N, tok, PPT_BF16, stream);
}
