// lnlin.hip -- round 6: LayerNorm + a K = 384 linear of a frozen PointBERT block as ONE launch with the WEIGHT streamed and the ROWS
// stationary (point_encoder.py:46-55, 76: qkv = Linear(384, 1152, bias=False)(norm1(x + pos))).
//
// csrc/rowgemm.hip does this the other way round -- a workgroup keeps 384 columns of W in registers (8 waves x 144 VGPRs, a whole CU) and
// walks 32-row tiles of x through LDS: 33-38 us at 16 416 rows, MFMA-busy 0.13 (SQ counters): every tile is a barrier-bracketed
// load -> LayerNorm -> 72 MFMAs per wave -> C tile round, and nothing else fits on the CU.  What round 6 measured about the other way:
// a CU pulls 48 B/clk of fragment-ordered weights out of L2 with eight waves (tools/wstream_bench.hip), i.e. the 295 KB of a 384-column
// slice of W in ~6 000 cycles -- the time of the slice's MFMAs over 64 rows.  So here a workgroup takes 64 ROWS and one 384-column slice:
//     image[64, 384] = LayerNorm(x rows) (16 threads per row, two-pass statistics over DPP adds; 16-bit, LDS)
//     C[64, 384]     = image . W[slice]^T   12 k-steps, a wave owns 48 columns (4 row blocks x 3 column blocks: 144 MFMAs), W through a
//                      register ring fed from one running scalar offset (csrc/mlp_fused3.hip)
//     C leaves through the LDS the image occupied (16-byte row pieces; a lane's own 8 bytes would be 32-byte pieces)
// 51 KB of LDS and <= 128 VGPRs: TWO workgroups share a CU, and nothing ties them together -- one's row requests, LayerNorm and
// stores run under the other's MFMAs.  257 row chunks x 3 slices = 771 workgroups; ids that agree modulo 8 share an XCD, so the three
// slices of a chunk (consecutive ids / 8) read the same x rows out of one L2.
// (Measured and removed: 80-row chunks that walk all three slices themselves -- one read and one LayerNorm of the rows, 206 workgroups =
// one round at 16 416 rows, 1 008 KB taken in per 80 rows instead of 3 x 393 KB per 64 -- as one 8-wave workgroup per CU: 36.5-37.3 us
// against 28.0 here (79.7 / 51.8 at 32 832 rows), ring depth 4 or 6 alike.  With eight waves a CU took in 885 KB in ~26 us beside its MFMAs, with
// sixteen in two independent workgroups 786 KB in ~13 us; the bytes saved do not buy back the waves.)
#include "ppt_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int D = 384, NC = 384;                                   // K, columns per workgroup
constexpr int RB = 4, R = 16 * RB;                                 // row blocks / rows per workgroup
constexpr int AP = 2 * D + 32;                                     // image pitch (bytes): = 32 mod 256 -> conflict-free b128 fragment reads
constexpr int CP = 2 * NC + 16;                                    // C tile pitch (bytes)
constexpr int LDS_BYTES = R * AP;                                  // (the C tile, R x CP, reuses it)
static_assert(R * CP <= LDS_BYTES, "the C tile must fit where the image was");
constexpr int KS = D / 32;                                         // k-steps (12)
constexpr int DR = 3;                                              // ring depth in k-steps (divides 12; 4 -> 128 VGPRs with 4 spilled)
constexpr int WAVE_SLICE = KS * 3 * 1024;                          // bytes of one wave's fragments per slice (36 KiB)

__device__ __forceinline__ float row16_sum_l(float v)
{
    v += __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x4E, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x141, 0xf>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<0x140, 0xf>(__float_as_uint(v)));
    return v;
}

__device__ __forceinline__ void lds_barrier_l()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Diagnostic build only (tools/lnlin_stamp.py compiles this file with -DPPT_LNLIN_STAMP into its own library): lane 0 of every wave stores
// s_memtime at its phase boundaries into the buffer passed as `bias`: [workgroup][wave][8].
#ifdef PPT_LNLIN_STAMP
#define LNLIN_STAMP(slot) do { if (lane == 0) reinterpret_cast<unsigned long long *>(const_cast<float *>(p.bias))[((size_t)blockIdx.x * 8 + w) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define LNLIN_BIAS ((const float *)nullptr)
#else
#define LNLIN_STAMP(slot) do { } while (0)
#define LNLIN_BIAS (p.bias)
#endif

template <typename F>
__global__ __launch_bounds__(512, 4) void lnlin_kernel(const ppt_lnlin_params p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4, lo16 = lane * 16;
    // ids that agree modulo 8 share an XCD: the slices of a chunk are consecutive in (id / 8)
    const int b_lo = blockIdx.x & 7, b_q = blockIdx.x >> 3;
    const int slice = b_q % p.slices, chunk = (b_q / p.slices) * 8 + b_lo;
    const int row0 = chunk * R;
    if (row0 >= p.M) return;
    const int nrow = min(R, p.M - row0);
    LNLIN_STAMP(0);

    // ---- the chunk's rows first (they are what the first phase waits for; in-order return would put them behind the ring's 9 KB per
    // wave), BOTH passes at once -- a pass requested behind the other's arithmetic cost a second trip to L2 (~2 us of the ~11 a
    // workgroup lives): 16 threads per row, 32 rows per pass (rows past M: the last row again, zeroed below)
    const int lr0 = threadIdx.x >> 4, j = threadIdx.x & 15;
    float4 xf[R / 32][D / 64];
#pragma unroll
    for (int pass = 0; pass < R / 32; ++pass) {
        const float *src = p.x + (size_t)(row0 + min(32 * pass + lr0, nrow - 1)) * D;
#pragma unroll
        for (int i = 0; i < D / 64; ++i) xf[pass][i] = *reinterpret_cast<const float4 *>(src + 4 * (j + 16 * i));
    }

    // ---- the weight ring: W in fragment order [slice][wave][ks < 12][nb < 3][lane][8] (ppt_lnlin_retile), first k-steps requested at once
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.W), 0, p.slices * 8 * WAVE_SLICE, 0x00020000);
    int ow = (slice * 8 + w) * WAVE_SLICE;
    auto next = [&](uint4 (&f)[3]) {
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) f[nb] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, lo16 + 1024 * nb, ow, 0));
        ow += 3072;
    };
    uint4 g[DR][3];
#pragma unroll
    for (int i = 0; i < DR; ++i) next(g[i]);
    LNLIN_STAMP(1);

    // ---- LayerNorm -> the image (two-pass statistics over DPP adds)
    {
#pragma unroll
        for (int pass = 0; pass < R / 32; ++pass) {
            const int lr = 32 * pass + lr0;
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < D / 64; ++i) sm += (xf[pass][i].x + xf[pass][i].y) + (xf[pass][i].z + xf[pass][i].w);
            const float mean = row16_sum_l(sm) * (1.0f / (float)D);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < D / 64; ++i) {
                const float d0 = xf[pass][i].x - mean, d1 = xf[pass][i].y - mean, d2 = xf[pass][i].z - mean, d3 = xf[pass][i].w - mean;
                q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
            }
            const float rstd = 1.0f / sqrtf(row16_sum_l(q) * (1.0f / (float)D) + p.ln_eps);
            unsigned char *dst = smem + lr * AP;
#pragma unroll
            for (int i = 0; i < D / 64; ++i) {
                const int c = 4 * (j + 16 * i);
                const float4 gm = *reinterpret_cast<const float4 *>(p.ln_w + c), bt = *reinterpret_cast<const float4 *>(p.ln_b + c);
                uint2 o = make_uint2(0u, 0u);
                if (lr < nrow)
                    o = make_uint2(h16<F>::pack2((xf[pass][i].x - mean) * rstd * gm.x + bt.x, (xf[pass][i].y - mean) * rstd * gm.y + bt.y),
                                   h16<F>::pack2((xf[pass][i].z - mean) * rstd * gm.z + bt.z, (xf[pass][i].w - mean) * rstd * gm.w + bt.w));
                *reinterpret_cast<uint2 *>(dst + 2 * c) = o;
            }
        }
    }
    LNLIN_STAMP(2);
    lds_barrier_l();
    LNLIN_STAMP(3);

    // ---- C^T[n][m] = W[n][k] image[m][k]: a lane holds four consecutive columns of a row
    f32x4_t acc[RB][3];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[rb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        const unsigned char *ha = smem + l15 * AP + 16 * kg;
        uint4 fa[2][RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fa[0][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * AP);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[(ks + 1) & 1][rb] = *reinterpret_cast<const uint4 *>(ha + rb * 16 * AP + 64 * (ks + 1));
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) acc[rb][nb] = h16<F>::mfma16(g[ks % DR][nb], fa[ks & 1][rb], acc[rb][nb]);
            if (ks + DR < KS) next(g[ks % DR]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    LNLIN_STAMP(4);
    lds_barrier_l();                                                     // every wave is done reading the image
    LNLIN_STAMP(5);

    // ---- (+ bias) -> the C tile in LDS -> 16-byte row pieces
    {
        const int ncol = slice * NC + 48 * w;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const float4 bv = LNLIN_BIAS ? *reinterpret_cast<const float4 *>(LNLIN_BIAS + ncol + 16 * nb + 4 * kg) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                *reinterpret_cast<uint2 *>(smem + (16 * rb + l15) * CP + (48 * w + 16 * nb + 4 * kg) * 2) =
                    make_uint2(h16<F>::pack2(acc[rb][nb][0] + bv.x, acc[rb][nb][1] + bv.y), h16<F>::pack2(acc[rb][nb][2] + bv.z, acc[rb][nb][3] + bv.w));
        }
    }
    lds_barrier_l();
    LNLIN_STAMP(6);
    {
        constexpr int CPR = NC / 8;                                       // 16-byte pieces per row (48)
        F *C = (F *)p.C;
#pragma unroll
        for (int i = 0; i < R * CPR / 512; ++i) {
            const int c = threadIdx.x + 512 * i, row = c / CPR, ch = c - row * CPR;
            if (row < nrow)
                *reinterpret_cast<uint4 *>(C + (size_t)(row0 + row) * p.N + slice * NC + 8 * ch) =
                    *reinterpret_cast<const uint4 *>(smem + row * CP + 16 * ch);
        }
    }
    LNLIN_STAMP(7);
}

// fragment order: Wt[slice][w][ks < 12][nb < 3][lane][8] = W[384 slice + 48 w + 16 nb + l15][32 ks + 8 kg ..)       W [N, 384] row-major
__global__ __launch_bounds__(256) void lnlin_retile_kernel(const bf16_t *__restrict__ W, bf16_t *__restrict__ Wt, int slices)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                        // over slices * 8 * 36 * 64 pieces
    if (i >= slices * 8 * 36 * 64) return;
    const int lane = i & 63, f = (i >> 6) % 36, w = (i / (64 * 36)) & 7, s = i / (64 * 36 * 8);
    const int l15 = lane & 15, kg = lane >> 4, ks = f / 3, nb = f % 3;
    *reinterpret_cast<uint4 *>(Wt + (size_t)i * 8) = *reinterpret_cast<const uint4 *>(W + (size_t)(NC * s + 48 * w + 16 * nb + l15) * D + 32 * ks + 8 * kg);
}

}  // namespace

extern "C" int ppt_lnlin_retile(const void *W, void *W_tiled, int N, void *stream)
{
    if (!W || !W_tiled || N <= 0 || N % NC || (((uintptr_t)W | (uintptr_t)W_tiled) & 15)) return PPT_EINVAL;
    const int slices = N / NC;
    hipLaunchKernelGGL(lnlin_retile_kernel, dim3((slices * 8 * 36 * 64 + 255) / 256), dim3(256), 0, ppt_stream(stream), (const bf16_t *)W,
                       (bf16_t *)W_tiled, slices);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_lnlin(const ppt_lnlin_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_lnlin_params p = *pp;
    if (!p.x || !p.W || !p.C || !p.ln_w || !p.ln_b || p.M <= 0 || p.N <= 0) return PPT_EINVAL;
    if (p.K != D || p.N % NC) return PPT_EUNSUPPORTED;
    if (p.dtype != PPT_BF16 && p.dtype != PPT_F16) return PPT_EINVAL;
    if (((uintptr_t)p.x | (uintptr_t)p.W | (uintptr_t)p.C | (uintptr_t)p.ln_w | (uintptr_t)p.ln_b | (uintptr_t)p.bias) & 15) return PPT_EINVAL;
    p.slices = p.N / NC;
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)lnlin_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)lnlin_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return 0;
    }();
    (void)attrs_once;
    const int chunks = (p.M + R - 1) / R;
    const int grid = ((chunks + 7) / 8) * 8 * p.slices;                  // (chunk ids past the last return at once)
    if (p.dtype == PPT_F16) hipLaunchKernelGGL(lnlin_kernel<f16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), p);
    else hipLaunchKernelGGL(lnlin_kernel<bf16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
