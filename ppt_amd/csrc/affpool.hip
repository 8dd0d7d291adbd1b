// affpool.hip -- the last conv of a PointNet2 set-abstraction branch (models/pointnet2/pointnet2_utils.py:186-206, 236-266):
//     a = relu(scale * y1 + shift)   (the previous layer's folded BatchNorm + ReLU, applied while y1 is read)
//     v = a @ W^T + bias             [M, N], never written
//     BatchNorm partials of v per 32-row chunk, max and min of v over every `pool` consecutive rows (the group max of :205 /
//     :262 commutes with the monotone BN + ReLU that follows; ppt_pool_finish folds and finishes)
// K <= 128, so ppt_gemm (PPT_A_AFFINE_RELU, 128 x 128 tiles) walks 2-8 K slabs per tile and is latency, not bandwidth:
// 365 us for M = 2.1 M rows whose only HBM traffic is the 403 MB of y1.  Same scheme as mpn1.hip: a wave keeps its B
// operand (32 TJ columns x K) in registers for the whole kernel, a lane loads its own A fragment (16 bytes per k-step of
// its row) straight from y1 and applies the affine in registers (constants from a 1 KB LDS table), one group of 32 rows
// per MFMA tile, no staging, no barrier; the epilogue is register-only.  Same expression and summation order as the generic
// prologue / MFMA loop: maxima and minima are bit-identical to that path.
#include "ppt_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int KS, int TJ, int NWN, int PR>
__global__ __launch_bounds__(256) void affpool_kernel(const bf16_t *__restrict__ A, int64_t lda, int n_units,
                                                       const float *__restrict__ a_scale, const float *__restrict__ a_shift,
                                                       const bf16_t *__restrict__ W, const float *__restrict__ bias_p,
                                                       float *__restrict__ pmax, float *__restrict__ pmin,
                                                       float *__restrict__ part_sum, float *__restrict__ part_m2)
{
    constexpr int K = 16 * KS, N = 32 * TJ * NWN, NWM = 4 / NWN, TPU = PR == 64 ? 2 : 1;      // tiles per unit of work
    __shared__ __align__(16) float2 tab[K];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = w % NWN, wm = w / NWN;
    for (int c = threadIdx.x; c < K; c += 256) tab[c] = make_float2(a_scale[c], a_shift[c]);
    const int col = lane & 31, h = lane >> 5;
    const int n_w = 32 * TJ * wn;
    bf16x8_t bfrag[TJ][KS];
    float bias[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            bfrag[j][s] = *reinterpret_cast<const bf16x8_t *>(W + (size_t)(n_w + 32 * j + col) * K + 16 * s + 8 * h);
        bias[j] = bias_p ? bias_p[n_w + 32 * j + col] : 0.f;
    }
    __syncthreads();

    for (int u = blockIdx.x * NWM + wm; u < n_units; u += gridDim.x * NWM) {
        float umx[TJ], umn[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) { umx[j] = -INFINITY; umn[j] = INFINITY; }
#pragma unroll
        for (int tt = 0; tt < TPU; ++tt) {
            const int64_t t = (int64_t)u * TPU + tt;
            const bf16_t *ap = A + (t * 32 + col) * lda + 8 * h;
            uint4 araw[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) araw[s] = *reinterpret_cast<const uint4 *>(ap + 16 * s);
            f32x16_t acc[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 *tp = reinterpret_cast<const float4 *>(tab + 16 * s + 8 * h);   // (sc, sh) of 8 consecutive k
                const uint32_t wv[4] = {araw[s].x, araw[s].y, araw[s].z, araw[s].w};
                uint32_t pk[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 c2 = tp[i];                                                 // sc[2i], sh[2i], sc[2i+1], sh[2i+1]
                    const float lo = fmaxf(fmaf(__uint_as_float(wv[i] << 16), c2.x, c2.y), 0.0f);
                    const float hi = fmaxf(fmaf(__uint_as_float(wv[i] & 0xFFFF0000u), c2.z, c2.w), 0.0f);
                    pk[i] = pack_bf16x2(lo, hi);
                }
                const bf16x8_t a = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[j][s], acc[j], 0, 0, 0);
            }
            // C layout: column (lane & 31), rows (e & 3) + 8 (e >> 2) + 4 h: e < 8 are rows 0-15, e >= 8 rows 16-31
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int n = n_w + 32 * j + col;
                float m0 = -INFINITY, m1 = -INFINITY, l0 = INFINITY, l1 = INFINITY, sm = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[j][e] += bias[j];
                    sm += acc[j][e];
                    if (e < 8) { m0 = fmaxf(m0, acc[j][e]); l0 = fminf(l0, acc[j][e]); }
                    else { m1 = fmaxf(m1, acc[j][e]); l1 = fminf(l1, acc[j][e]); }
                }
                sm = xor32_sum(sm);
                const float mean = sm * (1.0f / 32.0f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = acc[j][e] - mean; q = fmaf(d, d, q); }
                q = xor32_sum(q);
                if (h == 0) {
                    part_sum[t * N + n] = sm;
                    part_m2[t * N + n] = q;
                }
                if constexpr (PR == 16) {
                    m0 = xor32_max(m0); m1 = xor32_max(m1); l0 = xor32_min(l0); l1 = xor32_min(l1);
                    if (h == 0) {
                        pmax[(2 * t) * N + n] = m0; pmin[(2 * t) * N + n] = l0;
                        pmax[(2 * t + 1) * N + n] = m1; pmin[(2 * t + 1) * N + n] = l1;
                    }
                } else {
                    umx[j] = fmaxf(umx[j], fmaxf(m0, m1));
                    umn[j] = fminf(umn[j], fminf(l0, l1));
                }
            }
        }
        if constexpr (PR != 16) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const float mx = xor32_max(umx[j]), mn = xor32_min(umn[j]);
                if (h == 0) {
                    pmax[(int64_t)u * N + n_w + 32 * j + col] = mx;
                    pmin[(int64_t)u * N + n_w + 32 * j + col] = mn;
                }
            }
        }
    }
}

int affpool_grid(int64_t units, int nwm, hipStream_t st)
{
    const int cus = ppt_cu_count(st);            // (of the stream's device, not process-global state)
    const int64_t blocks = (units + nwm - 1) / nwm;
    return (int)(blocks < (int64_t)cus * 4 ? blocks : (int64_t)cus * 4);
}

}  // namespace

extern "C" int ppt_affine_conv_pool_bf16(const void *A, int64_t lda, int64_t M, int K, const float *a_scale, const float *a_shift,
                                         const void *W, const float *bias, int N, int pool_rows, float *pmax, float *pmin,
                                         float *part_sum, float *part_m2, void *stream)
{
    if (!A || !a_scale || !a_shift || !W || !pmax || !pmin || !part_sum || !part_m2 || M <= 0) return PPT_EINVAL;
    if ((lda % 8) || (((uintptr_t)A | (uintptr_t)W) & 15)) return PPT_EINVAL;
    if (M % (pool_rows == 64 ? 64 : 32)) return PPT_EUNSUPPORTED;
    hipStream_t s = ppt_stream(stream);
    const int64_t units = M / (pool_rows == 64 ? 64 : 32);
#define AFP_LAUNCH(KS, TJ, NWN, PR)                                                                                            \
    hipLaunchKernelGGL((affpool_kernel<KS, TJ, NWN, PR>), dim3(affpool_grid(units, 4 / NWN, s)), dim3(256), 0, s, (const bf16_t *)A, lda, \
                       (int)units, a_scale, a_shift, (const bf16_t *)W, bias, pmax, pmin, part_sum, part_m2)
    if (K == 32 && N == 64 && pool_rows == 16) AFP_LAUNCH(2, 2, 1, 16);
    else if (K == 64 && N == 128 && pool_rows == 32) AFP_LAUNCH(4, 2, 2, 32);
    else if (K == 96 && N == 128 && pool_rows == 64) AFP_LAUNCH(6, 2, 2, 64);
    else if (K == 128 && N == 256 && pool_rows == 64) AFP_LAUNCH(8, 2, 4, 64);
    else return PPT_EUNSUPPORTED;
#undef AFP_LAUNCH
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
