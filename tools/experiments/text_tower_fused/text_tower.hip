// text_tower.hip -- the CLIP text tower under PromptLearner as ONE persistent kernel per direction (bf16 mode).
//
// Replaces, for the prompt chain of a training step (encode_text, ULIP_models.py:203-222; Transformer / ResidualAttentionBlock
// :35-67; frozen weights, input gradient only), the ~100 small launches of the unfused pipeline (engine.text_tower_forward /
// _backward: LayerNorm, in_proj, attention, out_proj, LayerNorm, c_fc + QuickGELU, c_proj per layer, and the dX twins).
//
// Decomposition.  Prompts are independent sequences; with the class name in the middle / at the end they share their first P
// positions (start token + leading context tokens), whose activations are the same in every prompt at every layer (causal
// mask).  A workgroup owns NP whole prompts PLUS ITS OWN COPY of the P shared rows:
//     rows of workgroup g:  [0, P) the shared prefix (positions 0 .. P-1), then prompt g*NP + n at rows P + n (L - P) ..., n < NP
// (ModelNet40: P = 17, L = 37, NP = 2 -> 57 rows in a 64-row tile, 20 workgroups).  With its private prefix copy a workgroup
// needs NOTHING from any other workgroup through all 12 layers, forward or backward: no grid barrier, no flag, no cross-XCD
// hand-off, placement-independent by construction.  Backward: the loss is a sum over prompts and backpropagation is linear in
// the upstream gradient, so workgroup g back-propagates the loss terms of ITS prompts through its own prefix copy and hands
// back a PARTIAL gradient for the prefix rows; the partials are summed once, at the end (ppt_prompt_rows_bwd lists every
// workgroup's prefix rows for the tokens they hold).  The price is redundant prefix work (20 x 57 = 1 140 rows instead of 817),
// irrelevant next to what actually bounds the kernel:
//
// Roofline.  Per layer a workgroup multiplies its 64 rows with 3.1 M weights (0.4 GFLOP: ~40 us of one CU's MFMA peak at 100 %)
// and must pull those 6.3 MB of bf16 weights through ONE CU's L2 -> register path (~28 B/clk ~ 67 GB/s: ~94 us).  The kernel is
// bound by that per-CU weight stream.  So the weights are pre-tiled once (they are frozen) into the exact order each wave
// consumes them -- [wave][layer][unit][k-step][column tile][lane][8 bf16], one wave-instruction = 1 KiB of consecutive bytes --
// and every wave runs a 16-deep register ring of such pieces that never drains across phases; the activations (A operands)
// sit in LDS images (pitch = 32 B mod 256: conflict-free 16-byte fragment reads).  Every GEMM of the layer is the same
// "unit": 64 rows x 512 columns (64 per wave) x K = 512 -- in_proj = 3 units, out_proj 1, c_fc 4 slabs, c_proj 4 slabs.
//
// Why this is the right trade for the step: the chain's ~100 launches occupied 100-400 workgroups each and were stretched
// 2.3x by the point tower running beside them; this kernel holds 20 CUs for ~1.2 ms and leaves the other 236 to
// the tower.
#include "ppt_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

constexpr int WD = 512, HID = 2048, NH = 8, MT = 64;          // width, MLP hidden, heads (of 64), rows per workgroup tile
constexpr int HP = 2 * WD + 32;                               // LDS row pitch of a [64 x 512] bf16 image (1056 = 32 mod 256)
constexpr int IMG = MT * HP;                                  // 67 584 bytes
constexpr int DEPTH = 8;                                      // weight ring: 1 KiB pieces in flight per wave
constexpr int PIECES = 64;                                    // pieces per unit per wave (16 k-steps x 4 column tiles)
static_assert(32 % DEPTH == 0, "a half unit (32 pieces) must keep the ring position");
constexpr int VIMG = 64 * 128;                                // wave-private V image (64 keys x 64 dims bf16)
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int v_off(int key, int dbyte) { return key * 128 + (dbyte ^ (((key >> 1) & 1) << 6)); }

// ---- addressing: buffer resources (base in 4 SGPRs) + 32-bit per-lane byte offsets + a wave-uniform SGPR offset.  With 64-bit
// per-lane pointers hipcc hoisted dozens of loop-invariant addresses out of the layer loop and spilled ~400 registers; with
// descriptors the per-lane part of every access is one 32-bit register.  An offset >= NUM_RECORDS is out of range: the load
// returns zeros and the store is dropped, which is how rows past the workgroup's last valid row are masked (no branches).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int NUM_RECORDS = 0x40000000;                       // 1 GiB window per descriptor (the SGPR offset is not range-checked)
constexpr unsigned OOB = 0x7ffffff0u;
#ifndef PPT_TT_AUX
#define PPT_TT_AUX 2
#endif
constexpr int AUX_NT = PPT_TT_AUX;                            // re-reads of this workgroup's own earlier stores: served by L2
__device__ __forceinline__ rsrc_t mk_rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, NUM_RECORDS, 0x00020000); }
template <int AUX> __device__ __forceinline__ uint4 bld16(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
template <int AUX> __device__ __forceinline__ float4 bldf4(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
__device__ __forceinline__ void bst16(rsrc_t r, unsigned voff, unsigned soff, uint4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bstf4(rsrc_t r, unsigned voff, unsigned soff, float4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bst8(rsrc_t r, unsigned voff, unsigned soff, uint2 v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bst4(rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, (int)soff, 0);
}

// LDS-only synchronisation: wait for this wave's LDS traffic, then the barrier -- NOT __syncthreads(), whose vmcnt(0) would
// drain the weight ring at every phase boundary
#ifdef PPT_TT_FULLBAR
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
// the three points per layer where other waves' GLOBAL stores are read back (qkv, x_mid, x_out): stores complete, then barrier
// wave-private LDS hand-over (a wave's own writes read back by itself): the LDS executes one wave's instructions in order, so no
// wait is needed -- only the COMPILER must not move the reads above the writes (the accesses go through differently typed pointers)
__device__ __forceinline__ void wave_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void vm_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8_t pack8(const f32x16_t &x, int s)
{
    const uint4 u = make_uint4(pack_bf16x2(x[8 * s + 0], x[8 * s + 1]), pack_bf16x2(x[8 * s + 2], x[8 * s + 3]),
                               pack_bf16x2(x[8 * s + 4], x[8 * s + 5]), pack_bf16x2(x[8 * s + 6], x[8 * s + 7]));
    return __builtin_bit_cast(bf16x8_t, u);
}

template <int NT> struct AccT { f32x4_t v[4][NT]; };      // [row block of 16][column tile of 16]: D[n = 16 t + 4 kg + i][m = 16 rb + l15]
typedef AccT<4> Acc;

template <int NT> __device__ __forceinline__ void acc_zero(AccT<NT> &a)
{
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int t = 0; t < NT; ++t) a.v[rb][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}
// a per-phase copy of a lane constant that hipcc cannot see through: everything derived from it is computed where it is used
// instead of being hoisted out of the layer loop and kept (or spilled) for the whole kernel
#define OPAQUE(x) asm volatile("" : "+v"(x))
// Wait states behind the LAST matrix instruction of an accumulation chain, fenced against the scheduler.  Found the hard way
// (tools/text_fused_eval_check.py: run-to-run differences in element 0 of four lanes of ONE 16 x 16 tile, any layer): where
// hipcc (ROCm 7.2) lets the final v_mfma_f32_16x16x32_bf16 of a chain write a NEW destination (D != C), it may reuse the old
// accumulator register two instructions later -- here a v_mov of an epilogue index into C's first register behind one more MFMA,
// a buffer_load and `s_nop 0` -- while the matrix pipe has not yet read SrcC for its last pass (columns 12-15: exactly the lanes
// that came out wrong, and only when the pipe was busy with the SIMD's other wave).  16 wait states cover the four passes.
#ifndef PPT_TT_TAIL_ASM
#define PPT_TT_TAIL_ASM "s_nop 7\n\ts_nop 7"
#endif
#define MFMA_TAIL() do { __builtin_amdgcn_sched_barrier(0); asm volatile(PPT_TT_TAIL_ASM); __builtin_amdgcn_sched_barrier(0); } while (0)
// diagnostic stamps (p.dbg != NULL only): shader clock of workgroup 0 / wave 0 at the phase boundaries of layer LSTAMP
#define STAMP(i) do { if (p.dbg && blockIdx.x == 0 && w == 0 && l == LSTAMP && lane == 0) p.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
constexpr int LSTAMP = 1;

// The wave's weight stream: piece i of the wave at byte offset (wave base + 1024 i) of the fragment-ordered buffer; a ring of
// DEPTH pieces in registers, every consumed slot refilled with the piece DEPTH ahead.
struct WStream {
    rsrc_t r;
    unsigned lane16;          // per-lane byte offset inside a piece
    unsigned soff;            // byte offset of the next piece to request (wave-uniform)
};
__device__ __forceinline__ bf16x8_t ws_next(WStream &ws)
{
    const bf16x8_t v = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(ws.r, (int)ws.lane16, (int)ws.soff, 0));
    ws.soff += 1024;
    return v;
}

// One PASS of a unit over NT of the wave's four 16-column tiles: acc[64 rows x 16 NT columns] += A[64 x 512] (LDS image) . W^T,
// the 16 NT weight pieces of the pass coming out of the ring in order ([k-step][tile]).  A unit is one pass with NT = 4 (64
// pieces), or two passes with NT = 2 (tiles 0-1, then 2-3) where a second 64-register accumulator would not fit.  16 NT is a
// multiple of DEPTH either way, so every pass starts at ring position 0.
template <int NT, int NA>
__device__ __forceinline__ void gemm_pass(AccT<NA> &acc, const unsigned char *img, bf16x8_t (&ring)[DEPTH], WStream &ws, const int l15,
                                          const int kg)
{
    const unsigned char *a0 = img + l15 * HP + 16 * kg;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        bf16x8_t fa[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) fa[rb] = *reinterpret_cast<const bf16x8_t *>(a0 + rb * 16 * HP + 64 * s);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8_t b = ring[(NT * s + t) % DEPTH];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                acc.v[rb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, fa[rb], acc.v[rb][t], 0, 0, 0);
            ring[(NT * s + t) % DEPTH] = ws_next(ws);
        }
    }
    MFMA_TAIL();
}
__device__ __forceinline__ void gemm_unit(Acc &acc, const unsigned char *img, bf16x8_t (&ring)[DEPTH], WStream &ws, const int l15,
                                          const int kg)
{
    gemm_pass<4, 4>(acc, img, ring, ws, l15, kg);
}

// LayerNorm of the workgroup's rows (fp32, row pitch WD; buffer r at SGPR offset soff) -> bf16 image in LDS; wave w takes rows
// w, w + 8, ...; two-pass mean / variance as norm.hip; rows >= nrow become zeros.  Statistics go to (rst, st_soff) + 4 row
// (mean) and + 4 (rows_total + row) (rstd) when want_stats.
template <int AUX>
__device__ __forceinline__ void ln_rows(rsrc_t r, unsigned soff, const float *__restrict__ gw, const float *__restrict__ gb,
                                        unsigned char *img, bool want_stats, rsrc_t rst, unsigned st_soff, unsigned rstd_delta,
                                        int nrow, int w, int lane)
{
    const int c = lane * 8;
    const float4 g0 = *reinterpret_cast<const float4 *>(gw + c), g1 = *reinterpret_cast<const float4 *>(gw + c + 4);
    const float4 b0 = *reinterpret_cast<const float4 *>(gb + c), b1 = *reinterpret_cast<const float4 *>(gb + c + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 v0[4], v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = w + 8 * (4 * half + i);
            const unsigned off = row < nrow ? (unsigned)(row * (WD * 4) + c * 4) : OOB;
            v0[i] = bldf4<AUX>(r, off, soff);
            v1[i] = bldf4<AUX>(r, off + 16, soff);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = w + 8 * (4 * half + i);
            const float v[8] = {v0[i].x, v0[i].y, v0[i].z, v0[i].w, v1[i].x, v1[i].y, v1[i].z, v1[i].w};
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
            const float mean = wave_reduce_sum(s) * (1.0f / (float)WD);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; q = fmaf(d, d, q); }
            const float rstd = 1.0f / sqrtf(wave_reduce_sum(q) * (1.0f / (float)WD) + 1e-5f);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = row < nrow ? (v[j] - mean) * rstd * g[j] + be[j] : 0.f;
            *reinterpret_cast<uint4 *>(img + row * HP + 16 * lane) =
                make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
            if (want_stats) {
                const unsigned so = (lane == 0 && row < nrow) ? (unsigned)(row * 4) : OOB;
                bst4(rst, so, st_soff, mean);
                bst4(rst, so, st_soff + rstd_delta, rstd);
            }
        }
    }
}

// key j is visible to query i (rows of the workgroup's layout): causal, and j is a shared-prefix row or of i's own prompt
__device__ __forceinline__ bool visible(int i, int j, int P, int own)
{
    if (j > i) return false;
    if (j < P) return true;
    return (j - P) / own == (i - P) / own;
}

// ------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void text_tower_fwd_kernel(const ppt_text_tower_params p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *img1 = smem, *img2 = smem + IMG;           // img1: h -> V images -> u slab;  img2: attention out -> h2
    // the layer's biases and LayerNorm parameters in LDS (refreshed per layer): an epilogue that fetched them from memory would
    // wait for its own YOUNGEST load, i.e. vmcnt(0) -- the whole weight ring drained at every epilogue
    float *prm = reinterpret_cast<float *>(smem + 2 * IMG);
    float *pb_in = prm, *pb_out = prm + 3 * WD, *pb_fc = prm + 4 * WD, *pb_proj = prm + 4 * WD + HID;
    float *pl1w = prm + 5 * WD + HID, *pl1b = pl1w + WD, *pl2w = pl1b + WD, *pl2b = pl2w + WD;
    PPT_PRIO(p.prio);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int P = p.P, own = p.L - p.P;
    const int RW = P + p.NP * own;
    const int ng = min(p.NP, p.C - (int)blockIdx.x * p.NP);
    const int nrow = P + ng * own;                            // valid rows of this workgroup (<= 64)
    const unsigned row0 = blockIdx.x * RW;
    const bool save = p.pre != nullptr;

    const rsrc_t rX0 = mk_rsrc(p.x0), rX = mk_rsrc(p.x), rXM = mk_rsrc(p.xmid), rQKV = mk_rsrc(p.qkv);
    const rsrc_t rA = mk_rsrc(p.a), rLSE = mk_rsrc(p.lse), rPRE = mk_rsrc(p.pre), rST = mk_rsrc(p.stats);

    // the wave's weight stream
    WStream ws;
    ws.r = mk_rsrc(p.wfrag);
    ws.lane16 = lane * 16;
    ws.soff = (unsigned)w * (unsigned)p.layers * (12u * PIECES * 1024u);
    bf16x8_t ring[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) ring[i] = ws_next(ws);

    // causal / ownership mask of the attention, as bits over this lane's 32 score elements per 32-query tile
    unsigned amask[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        unsigned m = 0;
        const int qi = 32 * qt + (lane & 31);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kj = 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (visible(qi, kj, P, own)) m |= 1u << (16 * sub + e);
            }
        amask[qt] = m;
    }
    const float c = p.scale * LOG2E;

#pragma unroll 1
    for (int l = 0; l < p.layers; ++l) {
        // wave-uniform byte offsets of this layer's rows in each tensor
        const unsigned xin_soff = l == 0 ? row0 * (WD * 4) : (unsigned)((l - 1) * p.x_stride * 4) + row0 * (WD * 4);
        const unsigned xout_soff = (unsigned)(l * p.x_stride * 4) + row0 * (WD * 4);
        const unsigned xm_soff = (unsigned)(l * p.xm_stride * 4) + row0 * (WD * 4);
        const unsigned qkv_soff = (unsigned)(l * p.qkv_stride * 2) + row0 * (3 * WD * 2);
        const unsigned st_soff = (unsigned)(l * p.stats_stride * 4) + row0 * 4;
        const rsrc_t rXin = l == 0 ? rX0 : rX;

        STAMP(0);
        // ---- this layer's parameters -> LDS (the previous layer's last readers passed the barrier that ended it)
        {
            auto ldp = [&](const float *src, int n, float *dst) {
                for (int i = threadIdx.x; i < n / 4; i += 512) reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
            };
            ldp(p.b_in + (size_t)l * 3 * WD, 3 * WD, pb_in);
            ldp(p.b_out + (size_t)l * WD, WD, pb_out);
            ldp(p.b_fc + (size_t)l * HID, HID, pb_fc);
            ldp(p.b_proj + (size_t)l * WD, WD, pb_proj);
            ldp(p.ln1_w + (size_t)l * WD, WD, pl1w);
            ldp(p.ln1_b + (size_t)l * WD, WD, pl1b);
            ldp(p.ln2_w + (size_t)l * WD, WD, pl2w);
            ldp(p.ln2_b + (size_t)l * WD, WD, pl2b);
        }
        lds_barrier();

        STAMP(1);
        // ---- LN1 -> h (img1)
        if (l == 0) ln_rows<0>(rXin, xin_soff, pl1w, pl1b, img1, p.stats != nullptr, rST, st_soff, p.rows * 4, nrow, w, lane);
        else ln_rows<AUX_NT>(rXin, xin_soff, pl1w, pl1b, img1, p.stats != nullptr, rST, st_soff, p.rows * 4, nrow, w, lane);
        lds_barrier();

        STAMP(2);
        // ---- in_proj: three units (q, k, v); wave w computes head w's 64 columns of each
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            Acc acc;
            acc_zero(acc);
            gemm_unit(acc, img1, ring, ws, l15, kg);
            int l15e = l15, kge = kg;
            OPAQUE(l15e); OPAQUE(kge);
            const float *bias = pb_in + u * WD + 64 * w;
            const unsigned cb = (u * WD + 64 * w + 4 * kge) * 2;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kge);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const f32x4_t a = acc.v[rb][t];
                    const unsigned off = 16 * rb + l15e < nrow ? (unsigned)((16 * rb + l15e) * (3 * WD * 2)) + cb + 32 * t : OOB;
                    bst8(rQKV, off, qkv_soff, make_uint2(pack_bf16x2(a[0] + bv.x, a[1] + bv.y), pack_bf16x2(a[2] + bv.z, a[3] + bv.w)));
                }
            }
        }
        STAMP(3);
        vm_barrier();                                          // qkv complete in memory; h (img1) dead
        STAMP(4);

        // ---- attention of head w over the workgroup's rows (scores never leave registers).  The weight ring is given up for
        // the phase -- its DEPTH pieces are requested again behind it -- so that its registers are free here: the attention's
        // fragments and the ring together spilled, and every scratch reload is a vmcnt(0)
#ifndef PPT_TT_NOGIVEUP
        ws.soff -= DEPTH * 1024;
#endif
        {
            int r31 = lane & 31, hh = lane >> 5, lane_ = lane;
            OPAQUE(r31); OPAQUE(hh); OPAQUE(lane_);
            const int tg = lane_ >> 4, tq = (lane_ >> 2) & 3, tp = lane_ & 3;
            const int tr_key = 4 * (tg >> 1) + tq;
            const int tr_dbyte = (16 * (tg & 1) + 4 * tp) * 2;
            unsigned char *vimg = img1 + w * VIMG;
            bf16x8_t kf[2][4];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int row = 32 * sub + r31;
                    const unsigned off = row < nrow ? (unsigned)(row * (3 * WD * 2) + (WD + 64 * w + 16 * kk + 8 * hh) * 2) : OOB;
                    kf[sub][kk] = __builtin_bit_cast(bf16x8_t, bld16<AUX_NT>(rQKV, off, qkv_soff));
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cidx = lane_ + 64 * i, key = cidx >> 3, ch = cidx & 7;
                const unsigned off = key < nrow ? (unsigned)(key * (3 * WD * 2) + (2 * WD + 64 * w + ch * 8) * 2) : OOB;
                *reinterpret_cast<uint4 *>(vimg + v_off(key, ch * 16)) = bld16<AUX_NT>(rQKV, off, qkv_soff);
            }
            wave_fence();
            const unsigned a_soff = (unsigned)(l * p.a_stride * 2) + row0 * (WD * 2);
            const unsigned lse_soff = (unsigned)(l * p.lse_stride * 4) + row0 * (NH * 4);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qrow = 32 * qt + r31;
                bf16x8_t qf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const unsigned off = qrow < nrow ? (unsigned)(qrow * (3 * WD * 2) + (64 * w + 16 * kk + 8 * hh) * 2) : OOB;
                    qf[kk] = __builtin_bit_cast(bf16x8_t, bld16<AUX_NT>(rQKV, off, qkv_soff));
                }
                f32x16_t sc[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sc[sub][e] = 0.f;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    if (sub <= qt) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) sc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][kk], qf[kk], sc[sub], 0, 0, 0);
                    }
                MFMA_TAIL();
                unsigned am = amask[qt];
                asm volatile("" : "+v"(am));          // (opaque: keeps hipcc from hoisting 64 loop-invariant compares into SGPR pairs)
                float mx = -INFINITY;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((am >> (16 * sub + e)) & 1u) mx = fmaxf(mx, sc[sub][e]);
                mx = xor32_max(mx);
                const float mn = mx * c;
                float psum = 0.f;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float pv = ((am >> (16 * sub + e)) & 1u) ? __builtin_amdgcn_exp2f(fmaf(sc[sub][e], c, -mn)) : 0.f;
                        sc[sub][e] = pv;
                        psum += pv;
                    }
                const float lt = xor32_sum(psum);
                f32x16_t ot[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ot[i][e] = 0.f;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    if (sub <= qt) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const bf16x8_t pf = pack8(sc[sub], s);
#pragma unroll
                            for (int dtile = 0; dtile < 2; ++dtile) {
                                const int key0 = 32 * sub + 16 * s + tr_key;
                                struct { s4_t a, b; } vf;
                                vf.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                    (__attribute__((address_space(3))) s4_t *)(vimg + v_off(key0, tr_dbyte + 64 * dtile)));
                                vf.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                    (__attribute__((address_space(3))) s4_t *)(vimg + v_off(key0 + 8, tr_dbyte + 64 * dtile)));
                                ot[dtile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf), pf, ot[dtile], 0, 0, 0);
                            }
                        }
                    }
                MFMA_TAIL();
                const float inv = 1.0f / lt;
                const bool qok = qrow < nrow;
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const uint2 u2 = make_uint2(pack_bf16x2(ot[dtile][4 * gq + 0] * inv, ot[dtile][4 * gq + 1] * inv),
                                                    pack_bf16x2(ot[dtile][4 * gq + 2] * inv, ot[dtile][4 * gq + 3] * inv));
                        const int d = 32 * dtile + 8 * gq + 4 * hh;
                        *reinterpret_cast<uint2 *>(img2 + qrow * HP + (64 * w + d) * 2) = u2;
                        if (save) bst8(rA, qok ? (unsigned)(qrow * (WD * 2) + (64 * w + d) * 2) : OOB, a_soff, u2);
                    }
                if (save) bst4(rLSE, (qok && hh == 0) ? (unsigned)(qrow * (NH * 4) + w * 4) : OOB, lse_soff, (mn + __log2f(lt)) * 0.6931471805599453f);
            }
        }
#ifndef PPT_TT_NOGIVEUP
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ring[i] = ws_next(ws);
#endif
        STAMP(5);
        lds_barrier();                                         // attention output image complete; V images dead
        STAMP(6);

        // ---- out_proj + residual -> x_mid
        {
            Acc acc;
            acc_zero(acc);
            gemm_unit(acc, img2, ring, ws, l15, kg);
            int l15e = l15, kge = kg;
            OPAQUE(l15e); OPAQUE(kge);
            const float *bias = pb_out + 64 * w;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const unsigned ro = 16 * rb + l15e < nrow ? (unsigned)((16 * rb + l15e) * (WD * 4) + (64 * w + 4 * kge) * 4) : OOB;
                float4 rv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) rv[t] = l == 0 ? bldf4<0>(rXin, ro + 64 * t, xin_soff) : bldf4<AUX_NT>(rXin, ro + 64 * t, xin_soff);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kge);
                    const f32x4_t a = acc.v[rb][t];
                    bstf4(rXM, ro + 64 * t, xm_soff, make_float4(a[0] + bv.x + rv[t].x, a[1] + bv.y + rv[t].y, a[2] + bv.z + rv[t].z, a[3] + bv.w + rv[t].w));
                }
            }
        }
        STAMP(7);
        vm_barrier();                                          // x_mid complete in memory; attention image dead
        STAMP(8);

        // ---- LN2 -> h2 (img2)
        ln_rows<AUX_NT>(rXM, xm_soff, pl2w, pl2b, img2, p.stats != nullptr, rST, st_soff + 2 * p.rows * 4, p.rows * 4, nrow, w, lane);
        lds_barrier();

        STAMP(9);
        // ---- MLP: four hidden slabs of 512; c_fc + QuickGELU -> u (img1), c_proj accumulates over the slabs
        Acc accp;
        acc_zero(accp);
        const unsigned pre_soff = (unsigned)(l * p.pre_stride * 2) + row0 * (HID * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // c_fc of the slab in two passes of two column tiles (a second 64-register accumulator beside accp does not fit)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                AccT<2> acc;
                acc_zero(acc);
                gemm_pass<2, 2>(acc, img2, ring, ws, l15, kg);
                int l15e = l15, kge = kg;
                OPAQUE(l15e); OPAQUE(kge);
                const float *bias = pb_fc + j * WD + 64 * w + 32 * ps;
                const unsigned cb = (j * WD + 64 * w + 32 * ps + 4 * kge) * 2;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kge);
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        const int m = 16 * rb + l15e;
                        const f32x4_t a = acc.v[rb][t];
                        float v[4] = {a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w};
                        if (save)
                            bst8(rPRE, m < nrow ? (unsigned)(m * (HID * 2)) + cb + 32 * t : OOB, pre_soff,
                                 make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] / (1.0f + __expf(-1.702f * v[i]));      // QuickGELU (ULIP_models.py:30-32)
                        *reinterpret_cast<uint2 *>(img1 + m * HP + (64 * w + 32 * ps + 16 * t + 4 * kge) * 2) =
                            make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    }
                }
            }
            STAMP(10 + 3 * j);
            lds_barrier();                                     // slab complete
            STAMP(11 + 3 * j);
            gemm_unit(accp, img1, ring, ws, l15, kg);
            STAMP(12 + 3 * j);
            lds_barrier();                                     // slab consumed
        }
        {
            int l15e = l15, kge = kg;
            OPAQUE(l15e); OPAQUE(kge);
            const float *bias = pb_proj + 64 * w;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const unsigned ro = 16 * rb + l15e < nrow ? (unsigned)((16 * rb + l15e) * (WD * 4) + (64 * w + 4 * kge) * 4) : OOB;
                float4 rv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) rv[t] = bldf4<AUX_NT>(rXM, ro + 64 * t, xm_soff);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kge);
                    const f32x4_t a = accp.v[rb][t];
                    bstf4(rX, ro + 64 * t, xout_soff, make_float4(a[0] + bv.x + rv[t].x, a[1] + bv.y + rv[t].y, a[2] + bv.z + rv[t].z, a[3] + bv.w + rv[t].w));
                }
            }
        }
        STAMP(22);
        vm_barrier();                                          // x_out complete in memory
        STAMP(23);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// backward (input gradient only: the tower is frozen, ULIP_models.py:487-507)
// ------------------------------------------------------------------------------------------------------------------------
constexpr int FP = 4 * WD + 16;                               // row pitch of the fp32 [64 x 512] LDS buffer (both images as one)
constexpr int AUXLDS = 2 * IMG;                               // per-wave [64] lse2 + [64] delta behind the images

__device__ __forceinline__ float bld4(rsrc_t r, unsigned voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char *img, int row0, int dbyte)
{
    struct { s4_t a, b; } f;
    f.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + v_off(row0, dbyte)));
    f.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + v_off(row0 + 8, dbyte)));
    return __builtin_bit_cast(bf16x8_t, f);
}
__device__ __forceinline__ bf16x8_t row_frag(const unsigned char *img, int row, int chunk)
{
    return *reinterpret_cast<const bf16x8_t *>(img + v_off(row, chunk * 16));
}
__device__ __forceinline__ float dot8_bf16(uint4 a, uint4 b)
{
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc = fmaf(__uint_as_float(aw[i] << 16), __uint_as_float(bw[i] << 16), acc);
        acc = fmaf(__uint_as_float(aw[i] & 0xffff0000u), __uint_as_float(bw[i] & 0xffff0000u), acc);
    }
    return acc;
}

// accumulator (fp32 [64 rows x this wave's 64 columns]) -> the fp32 LDS buffer, 16 bytes per (row block, tile)
__device__ __forceinline__ void acc_to_lds(const Acc &a, unsigned char *buf, int w, int l15, int kg)
{
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<f32x4_t *>(buf + (16 * rb + l15) * FP + (64 * w + 16 * t + 4 * kg) * 4) = a.v[rb][t];
}

// LayerNorm backward of the workgroup's rows (input gradient only), wave w rows w, w + 8, ...:
//   g[row] <- g[row] + rstd (dy w - mean(dy w) - xhat mean(dy w xhat)),   dy from the fp32 LDS buffer, x / stats from memory;
// the new g rows go to memory (fp32) and are RETURNED as bf16 (8 columns per lane, row i of the wave in out[i]) so that the
// caller can write the next GEMM's A image after the barrier that ends the reads of the LDS buffer.
__device__ __forceinline__ void ln_bwd_rows(const unsigned char *buf, rsrc_t rx, unsigned x_soff, const float *__restrict__ gw, rsrc_t rst,
                                            unsigned mean_soff, unsigned rstd_delta, rsrc_t rg, unsigned g_soff, uint4 (&out)[8], int nrow,
                                            int w, int lane)
{
    const int c = lane * 8;
    const float4 w0 = *reinterpret_cast<const float4 *>(gw + c), w1 = *reinterpret_cast<const float4 *>(gw + c + 4);
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 x0[4], x1[4], a0[4], a1[4];
        float mu[4], rs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = w + 8 * (4 * half + i);
            const unsigned off = row < nrow ? (unsigned)(row * (WD * 4) + c * 4) : OOB;
            x0[i] = bldf4<0>(rx, off, x_soff); x1[i] = bldf4<0>(rx, off + 16, x_soff);
            a0[i] = bldf4<AUX_NT>(rg, off, g_soff); a1[i] = bldf4<AUX_NT>(rg, off + 16, g_soff);
            const unsigned so = row < nrow ? (unsigned)(row * 4) : OOB;
            mu[i] = bld4(rst, so, mean_soff); rs[i] = bld4(rst, so, mean_soff + rstd_delta);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = w + 8 * (4 * half + i);
            const float4 d0 = *reinterpret_cast<const float4 *>(buf + row * FP + c * 4), d1 = *reinterpret_cast<const float4 *>(buf + row * FP + c * 4 + 16);
            const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            const float xv[8] = {x0[i].x, x0[i].y, x0[i].z, x0[i].w, x1[i].x, x1[i].y, x1[i].z, x1[i].w};
            const float av[8] = {a0[i].x, a0[i].y, a0[i].z, a0[i].w, a1[i].x, a1[i].y, a1[i].z, a1[i].w};
            float gv[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[j] = (xv[j] - mu[i]) * rs[i];
                gv[j] = dv[j] * wv[j];
                s1 += gv[j]; s2 += gv[j] * xh[j];
            }
            s1 = wave_reduce_sum(s1) * (1.0f / (float)WD);
            s2 = wave_reduce_sum(s2) * (1.0f / (float)WD);
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = row < nrow ? av[j] + rs[i] * (gv[j] - s1 - xh[j] * s2) : 0.f;
            const unsigned off = row < nrow ? (unsigned)(row * (WD * 4) + c * 4) : OOB;
            bstf4(rg, off, g_soff, make_float4(t[0], t[1], t[2], t[3]));
            bstf4(rg, off + 16, g_soff, make_float4(t[4], t[5], t[6], t[7]));
            out[4 * half + i] = make_uint4(pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3]), pack_bf16x2(t[4], t[5]), pack_bf16x2(t[6], t[7]));
        }
    }
}

__global__ __launch_bounds__(512, 2) void text_tower_bwd_kernel(const ppt_text_tower_params p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *img1 = smem, *img2 = smem + IMG;
    PPT_PRIO(p.prio);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int P = p.P, own = p.L - p.P;
    const int RW = P + p.NP * own;
    const int ng = min(p.NP, p.C - (int)blockIdx.x * p.NP);
    const int nrow = P + ng * own;
    const unsigned row0 = blockIdx.x * RW;
    float *l2s = reinterpret_cast<float *>(smem + AUXLDS) + w * 128, *dls = l2s + 64;      // wave-private lse * log2(e), delta
    float *pg1 = reinterpret_cast<float *>(smem + AUXLDS) + 8 * 128, *pg2 = pg1 + WD;      // the layer's ln_1 / ln_2 weights

    const rsrc_t rX0 = mk_rsrc(p.x0), rX = mk_rsrc(p.x), rXM = mk_rsrc(p.xmid), rQKV = mk_rsrc(p.qkv);
    const rsrc_t rA = mk_rsrc(p.a), rLSE = mk_rsrc(p.lse), rPRE = mk_rsrc(p.pre), rST = mk_rsrc(p.stats);
    const rsrc_t rG = mk_rsrc(p.g), rDQ = mk_rsrc(p.dqkv);
    const unsigned g_soff = row0 * (WD * 4), dq_soff = row0 * (3 * WD * 2);

    WStream ws;
    ws.r = mk_rsrc(p.wfrag_bwd);
    ws.lane16 = lane * 16;
    ws.soff = (unsigned)w * (unsigned)p.layers * (12u * PIECES * 1024u);
    bf16x8_t ring[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) ring[i] = ws_next(ws);

    // visibility masks: amask[qt] bit (16 kt + e): key = 32 kt + (e & 3) + 8 (e >> 2) + 4 hh seen by query 32 qt + r31 (S^T layout);
    //                   bmask bit (16 pair + e): query = 32 qt + (e & 3) + 8 (e >> 2) + 4 hh sees key 32 kt + r31 (S layout),
    //                   pair = 0: (qt 0, kt 0), 1: (1, 0), 2: (1, 1)
    unsigned amask[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        unsigned m = 0;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (visible(32 * qt + (lane & 31), 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), P, own)) m |= 1u << (16 * sub + e);
        amask[qt] = m;
    }
    unsigned long long bmask = 0;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int qt = pr == 0 ? 0 : 1, kt = pr == 2 ? 1 : 0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (visible(32 * qt + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), 32 * kt + (lane & 31), P, own)) bmask |= 1ull << (16 * pr + e);
    }
    const float c = p.scale * LOG2E;

    // ---- the incoming gradient's bf16 image (img1): rows of g, zeros past nrow
    {
        const int cc = lane * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = w + 8 * i;
            const unsigned off = row < nrow ? (unsigned)(row * (WD * 4) + cc * 4) : OOB;
            const float4 a = bldf4<0>(rG, off, g_soff), b = bldf4<0>(rG, off + 16, g_soff);
            *reinterpret_cast<uint4 *>(img1 + row * HP + 16 * lane) =
                make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
        }
    }
    lds_barrier();

#pragma unroll 1
    for (int l = p.layers - 1; l >= 0; --l) {
        const unsigned xin_soff = l == 0 ? row0 * (WD * 4) : (unsigned)((l - 1) * p.x_stride * 4) + row0 * (WD * 4);
        const unsigned xm_soff = (unsigned)(l * p.xm_stride * 4) + row0 * (WD * 4);
        const unsigned qkv_soff = (unsigned)(l * p.qkv_stride * 2) + row0 * (3 * WD * 2);
        const unsigned a_soff = (unsigned)(l * p.a_stride * 2) + row0 * (WD * 2);
        const unsigned lse_soff = (unsigned)(l * p.lse_stride * 4) + row0 * (NH * 4);
        const unsigned pre_soff = (unsigned)(l * p.pre_stride * 2) + row0 * (HID * 2);
        const unsigned st_soff = (unsigned)(l * p.stats_stride * 4) + row0 * 4;
        const rsrc_t rXin = l == 0 ? rX0 : rX;
        if (threadIdx.x < 256) {          // this layer's LayerNorm weights -> LDS (first read two barriers further down)
            const float *src = threadIdx.x < 128 ? p.ln1_w + (size_t)l * WD : p.ln2_w + (size_t)l * WD;
            reinterpret_cast<float4 *>(threadIdx.x < 128 ? pg1 : pg2)[threadIdx.x & 127] = reinterpret_cast<const float4 *>(src)[threadIdx.x & 127];
        }

        STAMP(0);
        // ---- MLP backward: d_pre = (g W_proj) * QuickGELU'(pre) per hidden slab -> img2; d_h2 += d_pre W_fc
        Acc acch;
        acc_zero(acch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (g W_proj) of the slab in two passes of two column tiles: a second 64-register accumulator beside acch does not fit
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                AccT<2> acc;
                acc_zero(acc);
                int l15e = l15, kge = kg;
                OPAQUE(l15e); OPAQUE(kge);
                // the slab's saved pre-activations, requested BEFORE the pass (older than its ring refills: no drain when used)
                uint2 pv[4][2];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        pv[rb][t] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(
                            rPRE, (int)(16 * rb + l15e < nrow ? (unsigned)((16 * rb + l15e) * (HID * 2) + (j * WD + 64 * w + 32 * ps + 16 * t + 4 * kge) * 2) : OOB),
                            (int)pre_soff, 0));
                gemm_pass<2, 2>(acc, img1, ring, ws, l15, kg);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4_t a = acc.v[rb][t];
                        const float x[4] = {__uint_as_float(pv[rb][t].x << 16), __uint_as_float(pv[rb][t].x & 0xffff0000u),
                                            __uint_as_float(pv[rb][t].y << 16), __uint_as_float(pv[rb][t].y & 0xffff0000u)};
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float sg = 1.0f / (1.0f + __expf(-1.702f * x[i]));
                            v[i] = a[i] * (sg * (1.0f + 1.702f * x[i] * (1.0f - sg)));
                        }
                        *reinterpret_cast<uint2 *>(img2 + (16 * rb + l15e) * HP + (64 * w + 32 * ps + 16 * t + 4 * kge) * 2) =
                            make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    }
            }
            lds_barrier();
            gemm_unit(acch, img2, ring, ws, l15, kg);
            lds_barrier();
        }
        STAMP(1);
        // ---- LayerNorm-2 backward: g <- g + LN2'(d_h2); the new g's bf16 image -> img1
        {
            acc_to_lds(acch, smem, w, l15, kg);
            lds_barrier();
            uint4 nb[8];
            ln_bwd_rows(smem, rXM, xm_soff, pg2, rST, st_soff + 2 * p.rows * 4, p.rows * 4, rG, g_soff, nb, nrow, w, lane);
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4 *>(img1 + (w + 8 * i) * HP + 16 * lane) = nb[i];
        }
        lds_barrier();

        STAMP(2);
        // ---- out_proj backward: d_a = g W_out; wave w's 64 columns are head w's dO -> its private dO image (img2 + 8 KiB w)
        unsigned char *imD = img2 + w * VIMG, *imQ = img1 + w * VIMG;
        {
            Acc acc;
            acc_zero(acc);
            gemm_unit(acc, img1, ring, ws, l15, kg);
            int l15e = l15, kge = kg;
            OPAQUE(l15e); OPAQUE(kge);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4_t a = acc.v[rb][t];
                    *reinterpret_cast<uint2 *>(imD + v_off(16 * rb + l15e, (16 * t + 4 * kge) * 2)) =
                        make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
                }
        }
        STAMP(3);
        lds_barrier();                                         // every wave is done with img1
        STAMP(4);

        // ---- attention backward of head w (wave-private): dV, dK per key tile, then dQ per query tile -> DQKV scratch.
        // (the weight ring is given up for the phase and requested again behind it: see the forward kernel)
        ws.soff -= DEPTH * 1024;
        {
            int r31 = lane & 31, hh = lane >> 5, lane_ = lane;
            OPAQUE(r31); OPAQUE(hh); OPAQUE(lane_);
            const int tg = lane_ >> 4, tq = (lane_ >> 2) & 3, tp = lane_ & 3;
            const int tr_row = 4 * (tg >> 1) + tq;
            const int tr_dbyte = (16 * (tg & 1) + 4 * tp) * 2;
            const unsigned qcol = (64 * w) * 2, kcol = (WD + 64 * w) * 2, vcol = (2 * WD + 64 * w) * 2;
            // Q rows -> imQ; lse2 / delta of the 64 queries
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cidx = lane_ + 64 * i, row = cidx >> 3, ch = cidx & 7;
                const unsigned off = row < nrow ? (unsigned)(row * (3 * WD * 2)) + qcol + ch * 16 : OOB;
                *reinterpret_cast<uint4 *>(imQ + v_off(row, ch * 16)) = bld16<0>(rQKV, off, qkv_soff);
            }
            wave_fence();
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int q = 32 * qt + r31;
                float d = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const uint4 o = bld16<0>(rA, q < nrow ? (unsigned)(q * (WD * 2) + (64 * w + 16 * kk + 8 * hh) * 2) : OOB, a_soff);
                    const uint4 gq = __builtin_bit_cast(uint4, row_frag(imD, q, 2 * kk + hh));
                    d += dot8_bf16(gq, o);
                }
                d = xor32_sum(d);
                const float ls = q < nrow ? bld4(rLSE, (unsigned)(q * (NH * 4) + w * 4), lse_soff) * LOG2E : INFINITY;
                if (hh == 0) { l2s[q] = ls; dls[q] = q < nrow ? d : 0.f; }
            }
            wave_fence();
            // dK / dV: keys on the lane (S = Q K^T with the queries in the accumulator rows)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = 32 * kt + r31;
                bf16x8_t kf[4], vf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const unsigned off = key < nrow ? (unsigned)(key * (3 * WD * 2) + (16 * kk + 8 * hh) * 2) : OOB;
                    kf[kk] = __builtin_bit_cast(bf16x8_t, bld16<0>(rQKV, off + kcol, qkv_soff));
                    vf[kk] = __builtin_bit_cast(bf16x8_t, bld16<0>(rQKV, off + vcol, qkv_soff));
                }
                f32x16_t dvt[2], dkt[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { dvt[i][e] = 0.f; dkt[i][e] = 0.f; }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    if (qt >= kt) {
                        const int pr = qt + kt;                                   // (0,0) -> 0, (1,0) -> 1, (1,1) -> 2
                        f32x16_t sa, dp;
#pragma unroll
                        for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(imQ, 32 * qt + r31, 2 * kk + hh), kf[kk], sa, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(imD, 32 * qt + r31, 2 * kk + hh), vf[kk], dp, 0, 0, 0);
                        }
                        MFMA_TAIL();
                        unsigned bm = (unsigned)(bmask >> (16 * pr)) & 0xffffu;
                        asm volatile("" : "+v"(bm));
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const float4 l4 = *reinterpret_cast<const float4 *>(l2s + 32 * qt + 8 * gq + 4 * hh);
                            const float4 d4 = *reinterpret_cast<const float4 *>(dls + 32 * qt + 8 * gq + 4 * hh);
                            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const int e = 4 * gq + jj;
                                const float pv = ((bm >> e) & 1u) ? __builtin_amdgcn_exp2f(fmaf(sa[e], c, -lv[jj])) : 0.f;
                                sa[e] = pv;
                                dp[e] = pv * (dp[e] - dv[jj]) * p.scale;
                            }
                        }
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const bf16x8_t pf = pack8(sa, s), df = pack8(dp, s);
#pragma unroll
                            for (int dtile = 0; dtile < 2; ++dtile) {
                                dvt[dtile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(imD, 32 * qt + 16 * s + tr_row, tr_dbyte + 64 * dtile), pf, dvt[dtile], 0, 0, 0);
                                dkt[dtile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(imQ, 32 * qt + 16 * s + tr_row, tr_dbyte + 64 * dtile), df, dkt[dtile], 0, 0, 0);
                            }
                        }
                    }
                MFMA_TAIL();
                const unsigned ro = key < nrow ? (unsigned)(key * (3 * WD * 2)) : OOB;
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const unsigned dcol = (32 * dtile + 8 * gq + 4 * hh) * 2;
                        bst8(rDQ, ro + kcol + dcol, dq_soff, make_uint2(pack_bf16x2(dkt[dtile][4 * gq], dkt[dtile][4 * gq + 1]), pack_bf16x2(dkt[dtile][4 * gq + 2], dkt[dtile][4 * gq + 3])));
                        bst8(rDQ, ro + vcol + dcol, dq_soff, make_uint2(pack_bf16x2(dvt[dtile][4 * gq], dvt[dtile][4 * gq + 1]), pack_bf16x2(dvt[dtile][4 * gq + 2], dvt[dtile][4 * gq + 3])));
                    }
            }
            // dQ: queries on the lane (S^T = K Q^T); the K rows replace the Q rows in imQ
            bf16x8_t qfb[2][4];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) qfb[qt][kk] = row_frag(imQ, 32 * qt + r31, 2 * kk + hh);
            wave_fence();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cidx = lane_ + 64 * i, row = cidx >> 3, ch = cidx & 7;
                const unsigned off = row < nrow ? (unsigned)(row * (3 * WD * 2)) + kcol + ch * 16 : OOB;
                *reinterpret_cast<uint4 *>(imQ + v_off(row, ch * 16)) = bld16<0>(rQKV, off, qkv_soff);
            }
            wave_fence();
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int q = 32 * qt + r31;
                const float l2 = l2s[q], dl = dls[q];
                bf16x8_t gfb[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) gfb[kk] = row_frag(imD, q, 2 * kk + hh);
                f32x16_t dqt[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dqt[i][e] = 0.f;
                unsigned am = amask[qt];
                asm volatile("" : "+v"(am));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    if (kt <= qt) {
                        f32x16_t sa, dp;
#pragma unroll
                        for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const int key = 32 * kt + r31;
                            const bf16x8_t va = __builtin_bit_cast(bf16x8_t, bld16<0>(rQKV, key < nrow ? (unsigned)(key * (3 * WD * 2)) + vcol + (16 * kk + 8 * hh) * 2 : OOB, qkv_soff));
                            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(imQ, key, 2 * kk + hh), qfb[qt][kk], sa, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, gfb[kk], dp, 0, 0, 0);
                        }
                        MFMA_TAIL();
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float pv = ((am >> (16 * kt + e)) & 1u) ? __builtin_amdgcn_exp2f(fmaf(sa[e], c, -l2)) : 0.f;
                            dp[e] = pv * (dp[e] - dl) * p.scale;
                        }
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const bf16x8_t df = pack8(dp, s);
#pragma unroll
                            for (int dtile = 0; dtile < 2; ++dtile)
                                dqt[dtile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(imQ, 32 * kt + 16 * s + tr_row, tr_dbyte + 64 * dtile), df, dqt[dtile], 0, 0, 0);
                        }
                    }
                MFMA_TAIL();
                const unsigned ro = q < nrow ? (unsigned)(q * (3 * WD * 2)) + qcol : OOB;
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        bst8(rDQ, ro + (32 * dtile + 8 * gq + 4 * hh) * 2, dq_soff,
                             make_uint2(pack_bf16x2(dqt[dtile][4 * gq], dqt[dtile][4 * gq + 1]), pack_bf16x2(dqt[dtile][4 * gq + 2], dqt[dtile][4 * gq + 3])));
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ring[i] = ws_next(ws);
        STAMP(5);
        vm_barrier();                                          // d_qkv complete in memory; the private images are dead
        STAMP(6);

        // ---- in_proj backward: d_h = dq W_q + dk W_k + dv W_v (three units); the parts are staged from DQKV into the images
        Acc accd;
        acc_zero(accd);
        auto stage = [&](int part, unsigned char *img) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cidx = threadIdx.x + 512 * i, row = cidx >> 6, ch = cidx & 63;
                const unsigned off = row < nrow ? (unsigned)(row * (3 * WD * 2) + part * (WD * 2) + ch * 16) : OOB;
                *reinterpret_cast<uint4 *>(img + row * HP + ch * 16) = bld16<AUX_NT>(rDQ, off, dq_soff);
            }
        };
        stage(0, img1);
        stage(1, img2);
        lds_barrier();
        gemm_unit(accd, img1, ring, ws, l15, kg);
        gemm_unit(accd, img2, ring, ws, l15, kg);
        lds_barrier();
        stage(2, img1);
        lds_barrier();
        gemm_unit(accd, img1, ring, ws, l15, kg);
        lds_barrier();

        STAMP(7);
        // ---- LayerNorm-1 backward: g <- g + LN1'(d_h); the bf16 image of the new g -> img1 for the next layer down
        {
            acc_to_lds(accd, smem, w, l15, kg);
            lds_barrier();
            uint4 nb[8];
            ln_bwd_rows(smem, rXin, xin_soff, pg1, rST, st_soff, p.rows * 4, rG, g_soff, nb, nrow, w, lane);
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4 *>(img1 + (w + 8 * i) * HP + 16 * lane) = nb[i];
        }
        lds_barrier();
        STAMP(8);
    }
}

}  // namespace

constexpr int FWD_LDS = 2 * 67584 + (9 * 512 + 2048) * 4;      // two images + the layer's biases and LayerNorm parameters

extern "C" int ppt_text_tower_fwd_bf16(const ppt_text_tower_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_text_tower_params p = *pp;
    if (!p.x0 || !p.wfrag || !p.x || !p.xmid || !p.qkv || !p.ln1_w || !p.ln1_b || !p.ln2_w || !p.ln2_b || !p.b_in || !p.b_out ||
        !p.b_fc || !p.b_proj)
        return PPT_EINVAL;
    if (p.C <= 0 || p.L <= 0 || p.P < 0 || p.P >= p.L || p.NP <= 0 || p.layers <= 0) return PPT_EINVAL;
    if (p.P + p.NP * (p.L - p.P) > MT) return PPT_EUNSUPPORTED;
    if (((uintptr_t)p.x0 | (uintptr_t)p.wfrag | (uintptr_t)p.x | (uintptr_t)p.xmid | (uintptr_t)p.qkv | (uintptr_t)p.a | (uintptr_t)p.pre) & 15)
        return PPT_EINVAL;
    p.prio = ppt_get_wave_priority();
    static const int once = [] {
        return (int)hipFuncSetAttribute((const void *)text_tower_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS);
    }();
    (void)once;
    const int groups = (p.C + p.NP - 1) / p.NP;
    hipLaunchKernelGGL(text_tower_fwd_kernel, dim3(groups), dim3(512), FWD_LDS, ppt_stream(stream), p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_text_tower_bwd_bf16(const ppt_text_tower_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_text_tower_params p = *pp;
    if (!p.x0 || !p.wfrag_bwd || !p.x || !p.xmid || !p.qkv || !p.a || !p.lse || !p.pre || !p.stats || !p.g || !p.dqkv || !p.ln1_w || !p.ln2_w)
        return PPT_EINVAL;
    if (p.C <= 0 || p.L <= 0 || p.P < 0 || p.P >= p.L || p.NP <= 0 || p.layers <= 0) return PPT_EINVAL;
    if (p.P + p.NP * (p.L - p.P) > MT) return PPT_EUNSUPPORTED;
    if (((uintptr_t)p.x0 | (uintptr_t)p.wfrag_bwd | (uintptr_t)p.x | (uintptr_t)p.xmid | (uintptr_t)p.qkv | (uintptr_t)p.a | (uintptr_t)p.pre |
         (uintptr_t)p.g | (uintptr_t)p.dqkv) & 15)
        return PPT_EINVAL;
    p.prio = ppt_get_wave_priority();
    constexpr int LDS = 2 * IMG + 8 * 128 * 4 + 2 * WD * 4;
    static const int once = [] {
        return (int)hipFuncSetAttribute((const void *)text_tower_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }();
    (void)once;
    const int groups = (p.C + p.NP - 1) / p.NP;
    hipLaunchKernelGGL(text_tower_bwd_kernel, dim3(groups), dim3(512), LDS, ppt_stream(stream), p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
