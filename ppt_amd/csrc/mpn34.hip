// mpn34.hip -- the second half of the PointBERT mini-PointNet as ONE kernel (Encoder.second_conv + max, dvae.py:194-199, 211-214):
//     tok[g, :] = max over the 32 points m of group g of  W4 . relu( bn2( W3b . y2[m, :] + gterm[g, :] ) ) + b4
// y2 [M,256] 16-bit (the conv2 output, csrc/mpn1.hip), tok [M/32, 256] 16-bit.  The 512-wide conv3 output -- 537 MB written by
// mpn3.hip and read back by mpn4.hip per C2 step: 1.1 GB of the step's 6.5 GB of HBM traffic -- never exists (SURVEY §8(f) N1:
// "whole mini-PointNet fused").
//
// The folded BatchNorm is folded once more, into the conv (host side, engine.mini_pointnet): with scale s and shift t per channel
//     relu(s (W3b y + gterm) + t) = relu( (s o W3b) y + (s gterm + t) ) = relu( W3s y + gs[g] )
// so the kernel sees a plain conv with a per-GROUP bias gs [M/32, 512] f32, and that bias rides in the matrix pipe: a 17th k-step
// whose activation operand is the row's group indicator and whose weight operand is gs split into a 16-bit head and tail
// (hi + lo: the sum carries ~22 bits), formed per chunk from four floats per lane.
//
// A workgroup is 8 waves and owns CHUNKS of 128 rows (four groups), handled as two SUB-chunks of 64 rows in phase 1:
//   phase 1  y3 = relu(W3s y2 + gs):  wave w keeps rows 64 w .. 64 w + 63 of W3s in 128 VGPRs for the whole kernel (as mpn3.hip);
//            a sub-chunk's 32 KB of y2 arrive in LDS by LDS-DMA (row pitch 512 B, 16-byte chunks XOR-swizzled on the SOURCE side:
//            conflict-free ds_read_b128 fragments without padding); the product is formed transposed (weight first), so a lane
//            holds four consecutive channels of a row: ReLU, pack, 8-byte LDS writes into the y3 slab [128 x 512] (pitch 1 024 B,
//            chunks XOR-swizzled by the row: no padding -- the slab is 128 KB of the CU's 160).
//   phase 2  tok = max(W4 y3 + b4):  wave w owns output channels 32 w .. 32 w + 31; W4 STREAMS from L2 in fragment order
//            (ppt_mpn34_retile: one load instruction of a wave = 1 KB of consecutive bytes) through an 8-deep register ring, each
//            fragment feeding FOUR MFMAs (the chunk's four groups); A fragments from the y3 slab; max over the rows out of the
//            accumulator (mpn4.hip).
// Why 128 rows: what bounds the kernel is that W4 stream -- a CU takes in ~23 B/clk from L2 whatever the ring depth (in-kernel
// stamps, tools/mpn34_stamp.py: 256 KB in 11 100 cycles at depths 4 ... 16; the matrix pipe needs 4 096 for 64 rows) -- so the
// weight is streamed once per 128 rows, and the 128-row y3 slab takes all the LDS there is beside ONE 32 KB y2 buffer.  The second
// sub-chunk's y2 therefore lands in the slab's upper half, which its own epilogue overwrites afterwards (one barrier in between).
// Four barriers per chunk; y2 requests are issued right behind the barrier that frees their destination (the DMA needs no registers).
// Per chunk and CU: 2 176 MFMA 32x32x16 (17 408 matrix-pipe cycles per SIMD), 256 KB of W4 from L2, 64 KB of y2 from HBM.
#include "ppt_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int K1 = 256, N1 = 512, N2 = 256;                        // y2 width, y3 width, token width
constexpr int R = 128, RS = 64;                                    // rows per chunk (four groups) / per phase-1 sub-chunk
constexpr int KS1 = K1 / 16, KS2 = N1 / 16;                        // k-steps of the two products
constexpr int PA = 2 * K1;                                         // A slab pitch (bytes): unpadded, swizzled
constexpr int PY = 2 * N1;                                         // y3 slab pitch: 1 024 B, 16-byte chunks swizzled by (row & 15)
constexpr int A_BYTES = RS * PA, Y_BYTES = R * PY;
constexpr int LDS_BYTES = A_BYTES + Y_BYTES;
#ifndef PPT_M34_DR
#define PPT_M34_DR 8
#endif
constexpr int DR = PPT_M34_DR;                                     // depth of the register ring the W4 fragments stream through

typedef short ppt_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_pk(uint32_t v)
{
    const ppt_i16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ppt_i16x2, v), z));
}

template <typename F> __device__ __forceinline__ uint32_t one_one();      // {1.0, 1.0} in the operand format
template <> __device__ __forceinline__ uint32_t one_one<f16_t>() { return 0x3C003C00u; }
template <> __device__ __forceinline__ uint32_t one_one<bf16_t>() { return 0x3F803F80u; }

// Diagnostic build only (tools/mpn34_stamp.py compiles this file with -DPPT_M34_STAMP): s_memtime stamps of the phases of each
// workgroup's SECOND chunk go to a buffer passed in place of `bias`: [workgroup][wave][8].
#ifdef PPT_M34_STAMP
#define M34_STAMP(slot) do { if (lane == 0 && chunk == (int)blockIdx.x + (int)gridDim.x) stamps[((size_t)blockIdx.x * 8 + w) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define M34_STAMP(slot) do { } while (0)
#endif

template <typename F>
__global__ __launch_bounds__(512, 2) void mpn34_kernel(const bf16_t *__restrict__ A, int n_tiles, const bf16_t *__restrict__ W3s,
                                                        const float *__restrict__ gs, const bf16_t *__restrict__ W4t,
                                                        const float *__restrict__ bias, bf16_t *__restrict__ tok)
{
#ifdef PPT_M34_STAMP
    unsigned long long *stamps = (unsigned long long *)bias;
    bias = nullptr;
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *as = smem, *ys = smem + A_BYTES;
    unsigned char *as_b = ys + (size_t)RS * PY;                      // the second sub-chunk's y2 lands in the slab's upper half
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int n_chunks = (n_tiles + 3) >> 2;
    int chunk = blockIdx.x;
    if (chunk >= n_chunks) return;

    // ---- stationary operand: W3s rows 64 w + 32 jt + ml (phase 1)
    uint4 b3[2][KS1];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int s = 0; s < KS1; ++s)
            b3[jt][s] = *reinterpret_cast<const uint4 *>(W3s + (size_t)(64 * w + 32 * jt + ml) * K1 + 16 * s + 8 * h);
    // (the streamed fragments have the same addresses in every chunk: the pointer is laundered once per chunk -- see the loop -- or
    // LICM hoists all 32 loads out of it, into 128 registers the kernel does not have)
    const bf16_t *w4p = W4t;
    auto w4frag = [&](int s) { return *reinterpret_cast<const uint4 *>(w4p + ((size_t)(w * KS2 + s) * 64 + lane) * 8); };
    const float bias_v = bias ? bias[32 * w + ml] : 0.f;

    // ---- a sub-chunk's y2 rows -> LDS by LDS-DMA: instruction i of this wave fills rows 2 (4 w + i), 2 (4 w + i) + 1; lane ->
    // 16-byte position pos of its row, holding source chunk pos ^ (row & 15)
    auto request_a = [&](int ch, int sub, unsigned char *dst) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                                 // (recomputed per call: hoisted, the four 64-bit lane offsets spill)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 2 * (4 * w + i) + (ln >> 5), pos = ln & 31;
            const int64_t grow = (int64_t)ch * R + sub * RS + row;
            // rows past the end (a group count that is not a multiple of four): re-read the last valid row -- never stored
            const int64_t srow = grow < (int64_t)n_tiles * 32 ? grow : (int64_t)n_tiles * 32 - 1;
            lds_dma16(A + srow * K1 + 8 * (pos ^ (row & 15)), dst + (size_t)(2 * (4 * w + i)) * PA);
        }
    };
    // ---- phase 1 of one sub-chunk: MFMA loop over the y2 image at `ab`; returns with acc = W3s y2 + gs (pre-activation)
    f32x16_t acc[2][2];
    auto phase1 = [&](const unsigned char *ab, int g_first) {
        // group biases for this lane's two weight rows, requested in front of the loop that hides their latency
        const int g0 = min(g_first, n_tiles - 1), g1 = min(g_first + 1, n_tiles - 1);
        float gv[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            gv[jt][0] = gs[(size_t)g0 * N1 + 64 * w + 32 * jt + ml];
            gv[jt][1] = gs[(size_t)g1 * N1 + 64 * w + 32 * jt + ml];
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[rb][jt][e] = 0.f;
        const unsigned char *a0p = ab + (size_t)ml * PA, *a1p = ab + (size_t)(32 + ml) * PA;
        // (rows ml and 32 + ml share their low four bits.  The swizzle term is made opaque per phase: the 16 / 32 / 16 XOR-ed
        // offsets of the three fragment walks are loop-invariant, and hoisted out of the chunk loop they cost 64 VGPRs for the
        // whole kernel -- the weight registers then spill INTO the MFMA loops)
        int sw = ml & 15;
        asm volatile("" : "+v"(sw));
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const int off = 16 * ((2 * s + h) ^ sw);
            const uint4 a0 = *reinterpret_cast<const uint4 *>(a0p + off);
            const uint4 a1 = *reinterpret_cast<const uint4 *>(a1p + off);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                acc[0][jt] = h16<F>::mfma32(b3[jt][s], a0, acc[0][jt]);
                acc[1][jt] = h16<F>::mfma32(b3[jt][s], a1, acc[1][jt]);
            }
        }
        // the 17th k-step: weight operand = [hi(g0), hi(g1), lo(g0), lo(g1), 0 ...] (h = 0 half only), activation operand =
        // the row's group indicator [g == 0, g == 1, g == 0, g == 1, 0 ...]
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t oo = one_one<F>();
        const uint4 ind0 = h ? z : make_uint4(oo & 0x0000ffffu, oo & 0x0000ffffu, 0u, 0u);
        const uint4 ind1 = h ? z : make_uint4(oo & 0xffff0000u, oo & 0xffff0000u, 0u, 0u);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const float hi0 = h16<F>::to_f32(h16<F>::from_f32(gv[jt][0])), hi1 = h16<F>::to_f32(h16<F>::from_f32(gv[jt][1]));
            const uint4 wx = h ? z : make_uint4(h16<F>::pack2(hi0, hi1), h16<F>::pack2(gv[jt][0] - hi0, gv[jt][1] - hi1), 0u, 0u);
            acc[0][jt] = h16<F>::mfma32(wx, ind0, acc[0][jt]);
            acc[1][jt] = h16<F>::mfma32(wx, ind1, acc[1][jt]);
        }
    };
    // ---- ReLU, pack, 8-byte writes of acc into rows r0 .. r0 + 63 of the y3 slab (16-byte chunk c of row m at c ^ (m & 15))
    auto store_y3 = [&](int r0) {
        int sw = ml & 15;
        asm volatile("" : "+v"(sw));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                unsigned char *yp = ys + (size_t)(r0 + 32 * rb + ml) * PY + 8 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // ReLU on the PACKED pair: both 16-bit formats are sign-magnitude, so max(x, 0) is a signed 16-bit integer
                    // max with 0 (v_pk_max_i16: one instruction per two values; -0 and negative NaNs become +0)
                    const uint32_t p0 = relu_pk(h16<F>::pack2(acc[rb][jt][4 * q + 0], acc[rb][jt][4 * q + 1]));
                    const uint32_t p1 = relu_pk(h16<F>::pack2(acc[rb][jt][4 * q + 2], acc[rb][jt][4 * q + 3]));
                    const int c = 8 * w + 4 * jt + q;                // channels 64 w + 32 jt + 8 q + 4 h .. + 3: chunk c, half h
                    *reinterpret_cast<uint2 *>(yp + 16 * (c ^ sw)) = make_uint2(p0, p1);
                }
            }
    };
    request_a(chunk, 0, as);

    for (; chunk < n_chunks; chunk += gridDim.x) {
        M34_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's share of sub-chunk a has landed ...
        __syncthreads();                                             // ... everybody's has; phase 2 of the previous chunk is over
        M34_STAMP(1);
        request_a(chunk, 1, as_b);                                   // sub-chunk b -> the slab's (idle) upper half, under phase 1 (a)
        phase1(as, 4 * chunk);
        store_y3(0);
        M34_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                             // sub-chunk b is in LDS
        M34_STAMP(3);
        phase1(as_b, 4 * chunk + 2);
        // the first DR streamed W4 fragments are requested HERE, ahead of the epilogue and the barriers: their L2 round trip is over
        // when phase 2 starts (the accumulators of phase 1 are still live: 128 + 64 + 4 DR registers)
        asm volatile("" : "+s"(w4p));
        uint4 ring[DR];
#pragma unroll
        for (int s = 0; s < DR; ++s) ring[s] = w4frag(s);
        __syncthreads();                                             // every wave has read its last y2 fragment of the upper half ...
        store_y3(RS);                                                // ... which the rows 64 .. 127 of y3 now overwrite
        M34_STAMP(4);
        __syncthreads();                                             // the y3 slab is complete; the A buffer is free
        M34_STAMP(5);
        if (chunk + (int)gridDim.x < n_chunks) request_a(chunk + (int)gridDim.x, 0, as);

        // ---- phase 2: acc2[rb] [m = 32 rb + (e & 3) + 8 (e >> 2) + 4 h][c = 32 w + ml], rb = the chunk's four groups
        f32x16_t acc2[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[rb][e] = 0.f;
        {
            const unsigned char *yb = ys + (size_t)ml * PY;
            int sw = ml & 15;
            asm volatile("" : "+v"(sw));
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const int off = 16 * ((2 * s + h) ^ sw);
                const uint4 bw = ring[s % DR];
                const uint4 a0 = *reinterpret_cast<const uint4 *>(yb + off), a1 = *reinterpret_cast<const uint4 *>(yb + 32 * PY + off);
                acc2[0] = h16<F>::mfma32(a0, bw, acc2[0]);
                acc2[1] = h16<F>::mfma32(a1, bw, acc2[1]);
                const uint4 a2 = *reinterpret_cast<const uint4 *>(yb + 64 * PY + off), a3 = *reinterpret_cast<const uint4 *>(yb + 96 * PY + off);
                acc2[2] = h16<F>::mfma32(a2, bw, acc2[2]);
                acc2[3] = h16<F>::mfma32(a3, bw, acc2[3]);
                if (s + DR < KS2) ring[s % DR] = w4frag(s + DR);     // (slot free: refilled for k-step s + DR)
            }
        }
        M34_STAMP(6);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            // max first, bias after (max(x) + b == max(x + b) in any rounding: x -> x + b is monotone)
            float mx = fmaxf(fmaxf(acc2[rb][0], acc2[rb][1]), acc2[rb][2]);
#pragma unroll
            for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, acc2[rb][e]), acc2[rb][e + 1]);
            mx = xor32_max(fmaxf(mx, acc2[rb][15])) + bias_v;
            const int g = 4 * chunk + rb;
            if (h == 0 && g < n_tiles) tok[(size_t)g * N2 + 32 * w + ml] = h16<F>::from_f32(mx);
        }
        M34_STAMP(7);
    }
}

// W4 [256, 512] row-major -> fragment order of phase 2: W4t[w][s][lane] = W4[32 w + (lane & 31)][16 s + 8 (lane >> 5) .. + 8)
__global__ __launch_bounds__(256) void mpn34_retile_kernel(const bf16_t *__restrict__ W4, bf16_t *__restrict__ W4t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                    // over 8 * 32 * 64 pieces of 16 bytes
    if (i >= 8 * KS2 * 64) return;
    const int lane = i & 63, s = (i >> 6) % KS2, w = i / (64 * KS2);
    *reinterpret_cast<uint4 *>(W4t + (size_t)i * 8) =
        *reinterpret_cast<const uint4 *>(W4 + (size_t)(32 * w + (lane & 31)) * N1 + 16 * s + 8 * (lane >> 5));
}

}  // namespace

extern "C" int ppt_mpn34_retile(const void *W4, void *W4_tiled, void *stream)
{
    if (!W4 || !W4_tiled || (((uintptr_t)W4 | (uintptr_t)W4_tiled) & 15)) return PPT_EINVAL;
    hipLaunchKernelGGL(mpn34_retile_kernel, dim3((8 * KS2 * 64 + 255) / 256), dim3(256), 0, ppt_stream(stream), (const bf16_t *)W4, (bf16_t *)W4_tiled);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_mini_pointnet_conv34_half(const void *y2, int64_t M, const void *W3s, const float *gs, const void *W4_tiled,
                                             const float *bias4, void *tok, int dtype, void *stream)
{
    if (dtype != PPT_BF16 && dtype != PPT_F16) return PPT_EINVAL;
    if (!y2 || !W3s || !gs || !W4_tiled || !tok || M <= 0) return PPT_EINVAL;
    if (M % 32 || M / 32 > 0x3fffffff) return PPT_EUNSUPPORTED;
    if (((uintptr_t)y2 | (uintptr_t)W3s | (uintptr_t)W4_tiled) & 15) return PPT_EINVAL;
    static const int attrs_once = [] {
        (void)hipFuncSetAttribute((const void *)mpn34_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)mpn34_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return 0;
    }();
    (void)attrs_once;
    const int cus = ppt_cu_count(ppt_stream(stream));            // (of the stream's device, not process-global state)
    const int tiles = (int)(M / 32), chunks = (tiles + 3) / 4;
    // one persistent workgroup per CU, fewer when the caller leaves room for the other stream (ppt_set_persistent_occupancy)
    int64_t want = (int64_t)cus * ppt_get_persistent_occupancy() / 100;
    want = want < 8 ? 8 : want;
    const int grid = (int)(chunks < want ? chunks : want);
    if (dtype == PPT_F16)
        hipLaunchKernelGGL(mpn34_kernel<f16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), (const bf16_t *)y2, tiles, (const bf16_t *)W3s, gs,
                           (const bf16_t *)W4_tiled, bias4, (bf16_t *)tok);
    else
        hipLaunchKernelGGL(mpn34_kernel<bf16_t>, dim3(grid), dim3(512), LDS_BYTES, ppt_stream(stream), (const bf16_t *)y2, tiles, (const bf16_t *)W3s, gs,
                           (const bf16_t *)W4_tiled, bias4, (bf16_t *)tok);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
