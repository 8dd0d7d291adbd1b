// text_lin_split.hip -- round 6: ONE linear of the CLIP text tower's attention half on split16 products (fp32 operands multiplied as
// hi + lo IEEE-half pairs, gemm_common.h), rows stationary and the weight streamed -- the first product of csrc/text_mlp_split.hip as a
// launch of its own.  On the prompt chain (817 rows with the shared prefix) these are in_proj (512 -> 1536), out_proj (512 -> 512,
// + bias + residual) and their two input-gradient products (512 -> 512; 1536 -> 512 as three K chunks whose partial products the
// LayerNorm backward adds up, as it did for the split-K tile GEMM): 48 of the chain's 96 split16 tile GEMMs of ~20 us each -- 64 x 64
// tiles whose K loop pays a global round trip, a split and a barrier per 32 k.
//
// A workgroup = a 32-row block x a slice of SLN output columns x one 512-wide K chunk:
//   * the block's rows of A (fp32, columns [512 c, 512 c + 512)) are multiplied by 2^a_pow2, saturated to half's range (counted),
//     split ONCE and kept as a hi and a lo image in LDS (68 KB);
//   * the slice of W (split once per weight version by ppt_text_lin_retile_split, x 2^b_pow2, fragment order, hi KiB then lo KiB)
//     streams through a register ring; a wave owns SLN / 8 columns; three MFMAs per fragment pair, fp32 accumulation;
//   * epilogue: x 2^-(a_pow2 + b_pow2) (+ bias) (+ residual) -> fp32 C[M, N], or the K chunk's partial product parts[c][M][N].
// SLN = 256 for N >= 1024 (in_proj: 26 x 6 = 156 workgroups), 128 below (N = 512: 26 x 4 (x 3 chunks) = 104 / 312 workgroups).
#include "ppt_common.h"
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int KC = 512;                                            // K chunk
constexpr int RB = 2, R = 16 * RB;
constexpr int AP = 2 * KC + 32;                                    // LDS pitch (bytes): = 32 mod 256
constexpr int A_BYTES = R * AP;                                    // ONE image (hi or lo)
constexpr int LDS_BYTES = 2 * A_BYTES;
constexpr int K1 = KC / 32;                                        // k-steps (16)
constexpr int D1 = 4;                                              // ring depth in k-steps

__device__ __forceinline__ void lds_barrier_l2()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void split4_l(const float (&x)[4], uint2 &H, uint2 &L)
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

// NH = column halves of 16 a wave owns: 2 (SLN = 256) or 1 (SLN = 128)
template <int NH>
__global__ __launch_bounds__(512, 2) void text_lin_split_kernel(const ppt_text_lin_params p)
{
    constexpr int SLN = 128 * NH;
    constexpr int WAVE_SLICE = K1 * NH * 2048;                     // bytes of one wave's fragments per (slice, K chunk): hi + lo
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *ai = smem;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4, lo16 = lane * 16;
    PPT_PRIO(p.wave_prio);
    const int nsl = p.N / SLN, nck = p.K / KC;
    const int sc = blockIdx.x % (nsl * nck), s = sc % nsl, c = sc / nsl;          // ids that agree modulo (slices x chunks) share a weight slice
    const int row0 = (blockIdx.x / (nsl * nck)) * R;
    const int nrow = min(R, p.M - row0);
    const float sa = pow2f(p.split_a_pow2), inv = pow2f(-(p.split_a_pow2 + p.split_b_pow2));
    uint32_t over = 0;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W), 0, (int)((size_t)p.N * p.K * 4), 0x00020000);
    int o1 = ((c * nsl + s) * 8 + w) * WAVE_SLICE;
    auto next1 = [&](uint4 (&f)[2 * NH]) {
#pragma unroll
        for (int i = 0; i < 2 * NH; ++i) f[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r1, lo16 + 1024 * i, o1, 0));
        o1 += 2048 * NH;
    };
    uint4 g1[D1][2 * NH];
#pragma unroll
    for (int i = 0; i < D1; ++i) next1(g1[i]);

    // ---- the block's rows of A (fp32, this K chunk) -> scaled, saturated, split -> the hi and lo images (rows past M: zeros)
    {
        const float *A = (const float *)p.A + (size_t)c * KC;
        constexpr int PIECES = R * (KC / 4);
        float4 v[PIECES / 512];
#pragma unroll
        for (int it = 0; it < PIECES / 512; ++it) {
            const int i = threadIdx.x + 512 * it, lr = i / (KC / 4), c4 = i % (KC / 4);
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lr < nrow) v[it] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + lr) * p.lda + 4 * c4);
        }
#pragma unroll
        for (int it = 0; it < PIECES / 512; ++it) {
            const int i = threadIdx.x + 512 * it, lr = i / (KC / 4), c4 = i % (KC / 4);
            const float x[4] = {split_saturate(v[it].x * sa, over), split_saturate(v[it].y * sa, over),
                                split_saturate(v[it].z * sa, over), split_saturate(v[it].w * sa, over)};
            uint2 H, L;
            split4_l(x, H, L);
            *reinterpret_cast<uint2 *>(ai + lr * AP + 8 * c4) = H;
            *reinterpret_cast<uint2 *>(ai + A_BYTES + lr * AP + 8 * c4) = L;
        }
    }
    // bias and residual of this lane's output columns: requested now
    const int ncol = SLN * s + 16 * NH * w + 4 * kg;                 // + 16 h
    float4 bv[NH], rv[RB][NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        bv[h] = p.bias ? *reinterpret_cast<const float4 *>(p.bias + ncol + 16 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int m = row0 + min(16 * rb + l15, nrow - 1);
            rv[rb][h] = p.residual ? *reinterpret_cast<const float4 *>(p.residual + (size_t)m * p.ld_res + ncol + 16 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    lds_barrier_l2();

    f32x4_t a1[RB][NH];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int h = 0; h < NH; ++h) a1[rb][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
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
            for (int h = 0; h < NH; ++h) {
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
    // ---- epilogue: a lane holds four consecutive output columns of a row
    float *C = p.C + (size_t)c * p.M * p.ldc;                        // (K chunks > 1: parts[c][M][N], ldc == N)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int lr = 16 * rb + l15;
        if (lr < nrow) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
                *reinterpret_cast<float4 *>(C + (size_t)(row0 + lr) * p.ldc + ncol + 16 * h) =
                    make_float4(a1[rb][h][0] * inv + bv[h].x + rv[rb][h].x, a1[rb][h][1] * inv + bv[h].y + rv[rb][h].y,
                                a1[rb][h][2] * inv + bv[h].z + rv[rb][h].z, a1[rb][h][3] * inv + bv[h].w + rv[rb][h].w);
        }
    }
    split_report(over, p.split_overflow);
}

// fragment order: Wt[c][s][w][ks < 16][h < NH][hi, lo][lane][8] <- W[SLN s + 16 NH w + 16 h + l15][512 c + 32 ks + 8 kg ..) * 2^b_pow2      W [N, K] f32
template <int NH>
__global__ __launch_bounds__(256) void text_lin_retile_split_kernel(const float *__restrict__ W, unsigned char *__restrict__ Wt, int N, int K, int b_pow2)
{
    constexpr int SLN = 128 * NH;
    const int nsl = N / SLN, nck = K / KC;
    const int i = blockIdx.x * 256 + threadIdx.x;                        // over nck * nsl * 8 * 16 * NH * 64 pieces
    if (i >= nck * nsl * 8 * K1 * NH * 64) return;
    const int lane = i & 63, f = (i >> 6) % (K1 * NH), w = (i / (64 * K1 * NH)) & 7, sc = i / (64 * K1 * NH * 8);
    const int s = sc % nsl, c = sc / nsl, ks = f / NH, h = f % NH;
    const int l15 = lane & 15, kg = lane >> 4;
    const float sb = pow2f(b_pow2);
    const float *src = W + (size_t)(SLN * s + 16 * NH * w + 16 * h + l15) * K + KC * c + 32 * ks + 8 * kg;
    uint32_t over = 0;       // (weights are fitted into half's range by the caller: saturated, not counted)
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    const float x0[4] = {split_saturate(a.x * sb, over), split_saturate(a.y * sb, over), split_saturate(a.z * sb, over), split_saturate(a.w * sb, over)};
    const float x1[4] = {split_saturate(b.x * sb, over), split_saturate(b.y * sb, over), split_saturate(b.z * sb, over), split_saturate(b.w * sb, over)};
    uint2 h0, l0, h1, l1;
    split4_l(x0, h0, l0);
    split4_l(x1, h1, l1);
    unsigned char *dst = Wt + (size_t)(sc * 8 + w) * (K1 * NH * 2048) + (size_t)f * 2048 + lane * 16;
    *reinterpret_cast<uint4 *>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4 *>(dst + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

__host__ int slice_halves(int N) { return N >= 1024 ? 2 : 1; }

}  // namespace

extern "C" int ppt_text_lin_retile_split(const float *W, void *Wt, int N, int K, int b_pow2, void *stream)
{
    if (!W || !Wt || (((uintptr_t)W | (uintptr_t)Wt) & 15) || abs(b_pow2) > 24) return PPT_EINVAL;
    if (N <= 0 || K <= 0 || K % KC || N % 256) return PPT_EUNSUPPORTED;
    const int pieces = N * (K / 8);
    if (slice_halves(N) == 2)
        hipLaunchKernelGGL(text_lin_retile_split_kernel<2>, dim3((pieces + 255) / 256), dim3(256), 0, ppt_stream(stream), W, (unsigned char *)Wt, N, K, b_pow2);
    else
        hipLaunchKernelGGL(text_lin_retile_split_kernel<1>, dim3((pieces + 255) / 256), dim3(256), 0, ppt_stream(stream), W, (unsigned char *)Wt, N, K, b_pow2);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_text_lin_split(const ppt_text_lin_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_text_lin_params p = *pp;
    if (!p.A || !p.W || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0) return PPT_EINVAL;
    if (p.K % KC || p.N % 256) return PPT_EUNSUPPORTED;
    if (p.lda < p.K || (p.lda % 4) || p.ldc < p.N || (p.ldc % 4) || abs(p.split_a_pow2) > 24 || abs(p.split_b_pow2) > 24) return PPT_EINVAL;
    if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.C | (uintptr_t)p.bias | (uintptr_t)p.residual) & 15) return PPT_EINVAL;
    if (p.residual && (p.ld_res < p.N || (p.ld_res % 4))) return PPT_EINVAL;
    if (p.K > KC && (p.bias || p.residual || p.ldc != p.N)) return PPT_EINVAL;          // K chunks leave as plain partial products
    if (p.wave_prio == 0) p.wave_prio = ppt_get_wave_priority();
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)text_lin_split_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)text_lin_split_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return 0;
    }();
    (void)attrs_once;
    const int nh = slice_halves(p.N);
    const int grid = (p.N / (128 * nh)) * (p.K / KC) * ((p.M + R - 1) / R);
    hipStream_t st = ppt_stream(stream);
    if (nh == 2) hipLaunchKernelGGL((text_lin_split_kernel<2>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    else hipLaunchKernelGGL((text_lin_split_kernel<1>), dim3(grid), dim3(512), LDS_BYTES, st, p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
