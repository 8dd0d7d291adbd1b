// attention_mfma.hip -- bf16 flash-attention forward on the gfx950 matrix cores (head dim 64).
//
// Replaces point_encoder.py:46-55 (T = 513, non-causal) and nn.MultiheadAttention at
// ULIP_models.py:38,49-51 (L = 77, causal) for the bf16 performance mode.
//
//   workgroup = 4 waves = 128 query rows of one (batch, head); wave = one 32-row q tile;
//   K/V stream through LDS in 64-key tiles (register-staged double buffer, one barrier per tile).
//   QK^T is computed SWAPPED, S^T = K . Q^T (v_mfma_f32_32x32x16_bf16: A = K tile from LDS via
//   conflict-free ds_read_b128, B = Q^T held in registers for the whole kernel), so a lane owns one
//   query column: the online-softmax max / sum are register loops plus ONE v_permlane32_swap, and
//   the exponentiated S^T accumulator is, after a bf16 pack, directly the B operand of the second
//   product O^T = V^T . P (no LDS round trip for P).  V^T fragments come from the row-major V tile
//   with the hardware transpose read ds_read_b64_tr_b16; the V image XORs address bit 6 with bit 1
//   of the key so that the four key rows of a half-wave hit four different 64-byte bank quarters.
//   Scores never leave registers; softmax scale and log2(e) are folded into one v_exp_f32 argument.
#include "ppt_common.h"
#include <stdlib.h>
#include "attn_rowmap.h"

extern "C" int ppt_attention_fwd_quad_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H, float scale,
                                           int causal, int P, int fmt, hipStream_t s);

namespace {

typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int HD = 64, KVT = 64, QB = 128, TILE = KVT * 128;   // bytes per K or V tile (64 keys x 128 B)

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int key, int dbyte) { return key * 128 + (dbyte ^ (((key >> 1) & 1) << 6)); }

// (ppt_common.h: v_permlane32_swap with wait states on both sides)
__device__ __forceinline__ float lane_xor32_max(float v) { return xor32_max(v); }
__device__ __forceinline__ float lane_xor32_sum(float v) { return xor32_sum(v); }

// Workgroup -> (query block, batch x head): the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with an
// L2 of its own.  With the query blocks of one (batch, head) on consecutive workgroup ids its K / V rows were fetched from HBM
// by up to five XCDs (PMC: 139 MB read per launch at T = 513, B = 32 for 38 MB of qkv).  When the number of (batch, head)
// pairs is a multiple of 8 the ids are dealt so that all query blocks of a pair land on ONE XCD, next to each other in time.
// (Tried: T = 4 x 128 + 1 as four blocks of FIVE waves, the fifth wave of the last block owning the class-token row, instead of
// a fifth block that streams every K / V tile for one row: 36.3 -> 49.0 us.  Five-wave workgroups fit three to a CU instead of
// five, and this kernel lives on occupancy.  The ViT shape has a kernel of its own: attn_fwd_resident below.)
template <bool CAUSAL, typename F>
__global__ __launch_bounds__(256, 2) void attn_fwd_mfma(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ out,
                                                     float *__restrict__ lse, int Tfull, int H, float c /* scale*log2(e) */,
                                                     int P, int C, int prio, int xcd_map)
{
    __shared__ __align__(16) unsigned char smem[4 * TILE];    // K0 K1 V0 V1
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    int bh = blockIdx.y, qblk = blockIdx.x;
    if (xcd_map && (gridDim.y & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, slot = lin >> 3;
        bh = (slot / (int)gridDim.x) * 8 + (lin & 7);
        qblk = slot % (int)gridDim.x;
    }
    const int b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD;
    const int T = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);     // (attn_rowmap.h: b is a virtual sequence when P > 0)
    const bf16_t *qb = qkv + head * HD;
    const bf16_t *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const int q0 = qblk * QB + w * 32;
    const int qrow = q0 + r;

    uint4 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qrow < T) v = *reinterpret_cast<const uint4 *>(qb + am_row(Tfull, P, b, qrow) * rs + 16 * kk + 8 * h);
        qf[kk] = (v);
    }

    // T = 64 n + 1 (the ViT's class token: 513): the last key would cost a ninth 64-key tile for ONE key.  It is peeled:
    // the running (max, sum, O) state is INITIALISED from it -- m = c q.k_last, l = 1, O = v_last -- and the loop covers
    // the remaining 64 n keys in whole, unmasked tiles.  (Same softmax, another summation order.)
    const bool peel = !CAUSAL && T > KVT && (T % KVT) == 1;
    const int Tk = peel ? T - 1 : T;
    uint4 sk[2], sv[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int key = kt * KVT + (cidx >> 3), ch = cidx & 7;
            sk[i] = sv[i] = make_uint4(0, 0, 0, 0);
            if (key < Tk) {
                const int64_t kr = am_row(Tfull, P, b, key) * rs;
                sk[i] = *reinterpret_cast<const uint4 *>(kb + kr + ch * 8);
                sv[i] = *reinterpret_cast<const uint4 *>(vb + kr + ch * 8);
            }
        }
    };
    auto write_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int row = cidx >> 3, ch = cidx & 7;
            *reinterpret_cast<uint4 *>(smem + buf * TILE + k_off(row, ch)) = sk[i];
            *reinterpret_cast<uint4 *>(smem + (2 + buf) * TILE + v_off(row, ch * 16)) = sv[i];
        }
    };

    const int q_hi = min(T, (int)(qblk + 1) * QB) - 1;                // last query row of this workgroup
    const int nkt = CAUSAL ? min((T + KVT - 1) / KVT, q_hi / KVT + 1) : (Tk + KVT - 1) / KVT;

    f32x16_t ot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[i][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    if (peel) {
        const bf16_t *kl = kb + am_row(Tfull, P, b, T - 1) * rs, *vl = vb + am_row(Tfull, P, b, T - 1) * rs;
        float dot = 0.f;                                  // this lane's 32 of the 64 dimensions; the other half-wave has the rest
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint4 kv = *reinterpret_cast<const uint4 *>(kl + 16 * kk + 8 * h);
            const uint4 qv = (qf[kk]);
            const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dot = fmaf(h16<F>::lo(kw[e]), h16<F>::lo(qw[e]), dot);
                dot = fmaf(h16<F>::hi(kw[e]), h16<F>::hi(qw[e]), dot);
            }
        }
        m = lane_xor32_sum(dot) * c;
        l = h == 0 ? 1.0f : 0.0f;                         // (the two half-waves' sums are added at the end)
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const uint2 vv = *reinterpret_cast<const uint2 *>(vl + 32 * dtile + 8 * gq + 4 * h);
                ot[dtile][4 * gq + 0] = h16<F>::lo(vv.x); ot[dtile][4 * gq + 1] = h16<F>::hi(vv.x);
                ot[dtile][4 * gq + 2] = h16<F>::lo(vv.y); ot[dtile][4 * gq + 3] = h16<F>::hi(vv.y);
            }
    }

    // per-lane constant part of the transposed V reads: lane = 16g + 4q + p supplies row q, columns 4p..4p+3
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_key = 4 * (g >> 1) + tq;                 // + 32*sub + 16*s (+8 for the second read)
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;     // + 64*dt

    load_tile(0);
    write_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const bool active = q0 < T && (!CAUSAL || kt * KVT <= q0 + 31);      // wave-uniform
        if (active) {
            const unsigned char *Kc = smem + cur * TILE;
            const unsigned char *Vc = smem + (2 + cur) * TILE;
            f32x16_t st[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) st[i][e] = 0.f;
            // all eight K fragments first (independent ds_read_b128, one wait), then the MFMAs: loaded one by one, each
            // MFMA waited for its own LDS round trip
            uint4 kf[4][2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    kf[kk][sub] = __builtin_bit_cast(
                        uint4, *reinterpret_cast<const uint4 *>(Kc + k_off(32 * sub + r, 2 * kk + h)));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    st[sub] = h16<F>::mfma32(kf[kk][sub], qf[kk], st[sub]);
            const bool need_mask = (kt * KVT + KVT > Tk) || (CAUSAL && kt * KVT + KVT - 1 > q0);
            if (need_mask) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * KVT + 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (key >= Tk || (CAUSAL && key > qrow)) st[sub][e] = -INFINITY;
                    }
            }
            float mx = st[0][0];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[sub][e]);
            mx = lane_xor32_max(mx);
            const float mn = fmaxf(m, mx * c);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[sub][e], c, -mn));
                    st[sub][e] = pv;
                    psum += pv;
                }
            l = fmaf(l, alpha, psum);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) ot[i][e] *= alpha;
            // P (still in the S^T accumulator layout) -> bf16 B fragments of the 16-key k-steps
            uint4 pf[2][2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 u = make_uint4(h16<F>::pack2(st[sub][8 * s + 0], st[sub][8 * s + 1]),
                                               h16<F>::pack2(st[sub][8 * s + 2], st[sub][8 * s + 3]),
                                               h16<F>::pack2(st[sub][8 * s + 4], st[sub][8 * s + 5]),
                                               h16<F>::pack2(st[sub][8 * s + 6], st[sub][8 * s + 7]));
                    pf[sub][s] = (u);
                }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int dtile = 0; dtile < 2; ++dtile) {
                        const int key0 = 32 * sub + 16 * s + tr_key;
                        struct { s4_t a, b; } vf;
                        vf.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s4_t *)(Vc + v_off(key0, tr_dbyte + 64 * dtile)));
                        vf.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s4_t *)(Vc + v_off(key0 + 8, tr_dbyte + 64 * dtile)));
                        ot[dtile] = h16<F>::mfma32(__builtin_bit_cast(uint4, vf), pf[sub][s],
                                                                             ot[dtile]);
                    }
        }
        if (kt + 1 < nkt) write_tile(cur ^ 1);
        __syncthreads();
    }

    const float lt = lane_xor32_sum(l);
    if (qrow < T && qrow >= q_lo) {
        const float inv = 1.0f / lt;
        bf16_t *ob = out + am_row(Tfull, P, b, qrow) * (H * HD) + head * HD;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const uint2 u = make_uint2(h16<F>::pack2(ot[dtile][4 * gq + 0] * inv, ot[dtile][4 * gq + 1] * inv),
                                           h16<F>::pack2(ot[dtile][4 * gq + 2] * inv, ot[dtile][4 * gq + 3] * inv));
                *reinterpret_cast<uint2 *>(ob + 32 * dtile + 8 * gq + 4 * h) = u;
            }
        if (lse && h == 0) lse[am_stat(Tfull, P, H, b, head, qrow)] = (m + __log2f(lt)) * 0.6931471805599453f;
    }
}


// =================================================================================================
// The ViT shape (point_encoder.py:46-55: T = 64 n + 1 <= 513 tokens, non-causal): K and V of one (batch, head) are
// 2 x 64 KB -- they FIT in the 160 KB of LDS of a CU.  One workgroup of 8 waves per (batch, head):
//   * the 64 n keys go to LDS ONCE, by LDS-DMA (global_load_lds, 1 KB per wave and instruction: wave w brings rows 8 w .. 8 w + 7
//     of every 64-key tile of K and of V), in tile order; a wave waits for its own pieces of tile kt with a counted vmcnt
//     and the workgroup meets at one barrier per tile -- the fill runs ahead of the arithmetic, nothing is staged through
//     registers, no tile is ever fetched twice (the streaming kernel above: every 128-query block re-streams all keys through
//     a register-staged double buffer -- 5 blocks x 8 tiles x 2 barriers of latency per (batch, head), 139 MB from HBM per
//     launch for 38 MB of qkv);
//   * wave w owns the 64 queries 64 w .. 64 w + 63 as TWO 32-column B operands, so every K fragment (ds_read_b128) and every
//     transposed V fragment (ds_read_b64_tr_b16) feeds two MFMAs: half the LDS fragment traffic per flop;
//   * the last key (the 64 n + 1st) initialises the running softmax state as in the streaming kernel;
//   * the last QUERY row -- a 513th row would cost a ninth wave a whole pass -- is done beside the matrix work on the vector
//     ALU: wave w takes key tile w (lane = key: 64-term dot product against the wave-uniform query, softmax pieces by DPP
//     reductions, then lane = output dimension: 64 readlane-broadcast weights x V[key][lane]); the 8 partial (max, sum, O[64])
//     meet in LDS and wave 0 folds them with the last key's term.
// Same products, exponentials and fp32 accumulation as the streaming kernel; the summation order over keys is the same for
// rows < 64 n (tile order, last key first) and differs for the last row.
// =================================================================================================
constexpr int RES_MAXK = 512, RES_SCR_STRIDE = 68;
// Diagnostic build only (tools/attn_stamp.py compiles this file with -DPPT_ATTN_STAMP into its own library): lane 0 of every
// wave stores s_memtime at its phase boundaries into the buffer passed as `lse`: [workgroup][wave][16].
#ifdef PPT_ATTN_STAMP
#define ATTN_STAMP(slot) do { if (lane == 0) reinterpret_cast<unsigned long long *>(lse)[((size_t)blockIdx.x * 8 + w) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATTN_STAMP(slot) do { } while (0)
#endif
constexpr int RES_LDS = 2 * RES_MAXK * 128 + 8 * RES_SCR_STRIDE * 4;

template <typename F, int NKT>
__global__ __launch_bounds__(512) void attn_fwd_resident(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ out, float *__restrict__ lse,
                                                         int T, int H, float c /* scale*log2(e) */, int prio)
{
    extern __shared__ __align__(16) unsigned char rsm[];
    unsigned char *Ks = rsm, *Vs = rsm + RES_MAXK * 128;
    float *scr = reinterpret_cast<float *>(rsm + 2 * RES_MAXK * 128);
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD;
    constexpr int nkt = NKT;
    const int Tk = NKT * KVT;                                          // keys in LDS (whole tiles); key Tk = T - 1 is peeled
    const bf16_t *qb = qkv + (int64_t)b * T * rs + head * HD;          // row 0 of this sequence, q of this head
    const bf16_t *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const int q0 = 64 * w;
    const bool active = q0 < Tk;                                       // wave-uniform
    ATTN_STAMP(0);

    // ---- plain loads first (they complete in order, ahead of the fill): Q fragments, the last key / value
    uint4 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[qt][kk] = make_uint4(0, 0, 0, 0);
            if (active) qf[qt][kk] = *reinterpret_cast<const uint4 *>(qb + (int64_t)(q0 + 32 * qt + r) * rs + 16 * kk + 8 * h);
        }
    const bf16_t *kl = kb + (int64_t)Tk * rs, *vl = vb + (int64_t)Tk * rs;
    uint4 klf[4];
    uint2 vlf[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) klf[kk] = *reinterpret_cast<const uint4 *>(kl + 16 * kk + 8 * h);
#pragma unroll
    for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) vlf[dtile][gq] = *reinterpret_cast<const uint2 *>(vl + 32 * dtile + 8 * gq + 4 * h);

    // ---- the fill: per 64-key tile one K piece and one V piece of this wave (rows 8 w .. 8 w + 7 of the tile), swizzled on the
    // source side.  Tiles 0 and 1 here, tile kt + 2 behind the barrier of tile kt: issued all at once, the 16 pieces held every
    // wave at the issue stage for 7 500 cycles (in-kernel stamps: a CU's memory pipeline takes ~1 KB per 26 cycles) before its
    // first MFMA.  (The two compiler fences keep the plain loads OLDER than every piece: hipcc otherwise sinks some of them
    // between the pieces.)
    asm volatile("" ::: "memory");
    const int lr = 8 * w + (lane >> 3);                                // row inside the tile this lane feeds
    const int kc = (lane & 7) ^ ((lr >> 1) & 7);                       // K image: chunk c of row r sits in slot c ^ ((r >> 1) & 7)
    const int vc = (lane & 7) ^ (((lr >> 1) & 1) << 2);                // V image: byte bit 6 ^= bit 1 of the key <-> chunk bit 2
    const bf16_t *ksrc = kb + (int64_t)lr * rs + kc * 8, *vsrc = vb + (int64_t)lr * rs + vc * 8;
    auto fill = [&](int kt) {
        lds_dma16(ksrc + (int64_t)kt * KVT * rs, Ks + (kt * KVT + 8 * w) * 128);
        lds_dma16(vsrc + (int64_t)kt * KVT * rs, Vs + (kt * KVT + 8 * w) * 128);
    };
    fill(0);
    if (nkt > 1) fill(1);
    asm volatile("" ::: "memory");
    ATTN_STAMP(1);
    // ---- running state, initialised from the last key (m = c q.k_last, l = 1, O = v_last)
    f32x16_t ot[2][2];
    float m[2], l[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float dot = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint32_t kw[4] = {klf[kk].x, klf[kk].y, klf[kk].z, klf[kk].w};
            const uint32_t qw[4] = {qf[qt][kk].x, qf[qt][kk].y, qf[qt][kk].z, qf[qt][kk].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dot = fmaf(h16<F>::lo(kw[e]), h16<F>::lo(qw[e]), dot);
                dot = fmaf(h16<F>::hi(kw[e]), h16<F>::hi(qw[e]), dot);
            }
        }
        m[qt] = lane_xor32_sum(dot) * c;
        l[qt] = h == 0 ? 1.0f : 0.0f;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                ot[qt][dtile][4 * gq + 0] = h16<F>::lo(vlf[dtile][gq].x); ot[qt][dtile][4 * gq + 1] = h16<F>::hi(vlf[dtile][gq].x);
                ot[qt][dtile][4 * gq + 2] = h16<F>::lo(vlf[dtile][gq].y); ot[qt][dtile][4 * gq + 3] = h16<F>::hi(vlf[dtile][gq].y);
            }
    }

    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_key = 4 * (g >> 1) + tq;
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;

    // One counted wait + one barrier per tile: in-order completion, so with this wave's two youngest pieces (tile kt + 1) still
    // flying its pieces of tile kt have landed; behind the barrier everybody's have, and tile kt + 2 is requested.  The fill is
    // paced by what a CU pulls out of L2 (~600 cycles per tile), a tile's arithmetic takes ~4 000 (the loop is bound by the
    // vector ALU: 2 waves x ~1 700 issue cycles of softmax per tile and SIMD), so only tile 0 is ever waited for.
    // (Tried: two barriers -- after half of the tiles, after all -- with the upper four waves sleeping half a tile behind each
    // so that one wave's softmax would run under the other's MFMAs: no change per tile.)
#pragma unroll
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) fill(kt + 2);
        ATTN_STAMP(3 + kt);
        if (!active) continue;
        const unsigned char *Kc = Ks + kt * TILE, *Vc = Vs + kt * TILE;
        f32x16_t st[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) st[qt][sub][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 kf[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) kf[sub] = *reinterpret_cast<const uint4 *>(Kc + k_off(32 * sub + r, 2 * kk + h));
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) st[qt][sub] = h16<F>::mfma32(kf[sub], qf[qt][kk], st[qt][sub]);
        }
        uint4 pf[2][2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = st[qt][0][0];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[qt][sub][e]);
            mx = lane_xor32_max(mx);
            const float mn = fmaxf(m[qt], mx * c);
            const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);
            m[qt] = mn;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[qt][sub][e], c, -mn));
                    st[qt][sub][e] = pv;
                    psum += pv;
                }
            l[qt] = fmaf(l[qt], alpha, psum);
            // (the running maximum moves in the first tiles and then rarely: skip the 32 rescaling multiplies when no lane's did)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ot[qt][dtile][e] *= alpha;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    pf[qt][sub][s2] = make_uint4(h16<F>::pack2(st[qt][sub][8 * s2 + 0], st[qt][sub][8 * s2 + 1]),
                                                 h16<F>::pack2(st[qt][sub][8 * s2 + 2], st[qt][sub][8 * s2 + 3]),
                                                 h16<F>::pack2(st[qt][sub][8 * s2 + 4], st[qt][sub][8 * s2 + 5]),
                                                 h16<F>::pack2(st[qt][sub][8 * s2 + 6], st[qt][sub][8 * s2 + 7]));
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile) {
                    const int key0 = 32 * sub + 16 * s2 + tr_key;
                    struct { s4_t a, b; } vf;
                    vf.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4_t *)(Vc + v_off(key0, tr_dbyte + 64 * dtile)));
                    vf.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4_t *)(Vc + v_off(key0 + 8, tr_dbyte + 64 * dtile)));
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        ot[qt][dtile] = h16<F>::mfma32(__builtin_bit_cast(uint4, vf), pf[qt][sub][s2], ot[qt][dtile]);
                }
    }

    ATTN_STAMP(11);
    // ---- rows < Tk out
#ifndef PPT_ATTN_STAMP
    if (active) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qrow = q0 + 32 * qt + r;
            const float lt = lane_xor32_sum(l[qt]);
            const float inv = 1.0f / lt;
            bf16_t *ob = out + ((int64_t)b * T + qrow) * (H * HD) + head * HD;
#pragma unroll
            for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const uint2 u = make_uint2(h16<F>::pack2(ot[qt][dtile][4 * gq + 0] * inv, ot[qt][dtile][4 * gq + 1] * inv),
                                               h16<F>::pack2(ot[qt][dtile][4 * gq + 2] * inv, ot[qt][dtile][4 * gq + 3] * inv));
                    *reinterpret_cast<uint2 *>(ob + 32 * dtile + 8 * gq + 4 * h) = u;
                }
            if (lse && h == 0) lse[((int64_t)b * H + head) * T + qrow] = (m[qt] + __log2f(lt)) * 0.6931471805599453f;
        }
    }
#else
    if (active && ot[0][0][0] == 12345.678f && ot[1][1][3] == 1.5f && l[0] + l[1] == 7.f) out[0] = 0;       // (keep the work alive)
#endif
    ATTN_STAMP(12);

    // ---- the last query row (position Tk): wave w < nkt takes key tile w
    const uint32_t *qc = reinterpret_cast<const uint32_t *>(qb + (int64_t)Tk * rs);      // 32 packed pairs, wave-uniform
    if (w < nkt) {
        const unsigned char *Kc = Ks + w * TILE, *Vc = Vs + w * TILE;
        float dot = 0.f;                                                // lane = key 64 w + lane
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const uint4 kv = *reinterpret_cast<const uint4 *>(Kc + k_off(lane, ch));
            const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t qv = qc[4 * ch + e];
                dot = fmaf(h16<F>::lo(kw[e]), h16<F>::lo(qv), dot);
                dot = fmaf(h16<F>::hi(kw[e]), h16<F>::hi(qv), dot);
            }
        }
        const float sc = dot * c;
        const float mw = wave_reduce_max(sc);
        const float pk = __builtin_amdgcn_exp2f(sc - mw);
        const float lw = wave_reduce_sum(pk);
        float acc = 0.f;                                                // lane = output dimension
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float pw = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(pk), k));
            const uint16_t vv = *reinterpret_cast<const uint16_t *>(Vc + v_off(k, 2 * lane));
            acc = fmaf(pw, h16<F>::to_f32(vv), acc);
        }
        if (lane == 0) { scr[w * RES_SCR_STRIDE] = mw; scr[w * RES_SCR_STRIDE + 1] = lw; }
        scr[w * RES_SCR_STRIDE + 4 + lane] = acc;
    }
    ATTN_STAMP(13);
    __syncthreads();
    if (w == 0) {
        const uint16_t *qh = reinterpret_cast<const uint16_t *>(qc);
        const float self = wave_reduce_sum(h16<F>::to_f32(qh[lane]) * h16<F>::to_f32(kl[lane])) * c;     // the last key against itself
        float M = self;
        for (int i = 0; i < nkt; ++i) M = fmaxf(M, scr[i * RES_SCR_STRIDE]);
        const float es = __builtin_amdgcn_exp2f(self - M);
        float L = es, O = es * h16<F>::to_f32(vl[lane]);
        for (int i = 0; i < nkt; ++i) {
            const float e = __builtin_amdgcn_exp2f(scr[i * RES_SCR_STRIDE] - M);
            L = fmaf(scr[i * RES_SCR_STRIDE + 1], e, L);
            O = fmaf(scr[i * RES_SCR_STRIDE + 4 + lane], e, O);
        }
        out[((int64_t)b * T + Tk) * (H * HD) + head * HD + lane] = h16<F>::from_f32(O / L);
#ifndef PPT_ATTN_STAMP
        if (lse && lane == 0) lse[((int64_t)b * H + head) * T + Tk] = (M + __log2f(L)) * 0.6931471805599453f;
#endif
    }
    ATTN_STAMP(14);
}


// =================================================================================================
// backward: dQ, dK, dV by recomputation from Q, K, V, dO, LSE and delta = rowsum(dO * O).
// Two kernels, no atomics (deterministic):
//   attn_bwd_dkv_mfma  wave owns 32 keys (K, V fragments in registers as B operands -> "key on the
//                      lane"), sweeps the query tiles: S = Q K^T and dP = dO V^T land with the query
//                      index in the accumulator ROWS, so P and dS are, after a bf16 pack, directly the
//                      B operands of dV^T += dO^T P and dK^T += Q^T dS (A operands: hardware-transposed
//                      reads of the row-major Q / dO tiles).
//   attn_bwd_dq_mfma   wave owns 32 queries (Q, dO fragments in registers), sweeps the key tiles:
//                      S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T.
// Tiles that are read both row-wise (ds_read_b128) and transposed (ds_read_b64_tr_b16) are kept as
// two LDS images, each with the swizzle that makes its read conflict-free.
// =================================================================================================
template <typename F>
__device__ __forceinline__ uint4 pack8(const f32x16_t &x, int s)
{
    const uint4 u = make_uint4(h16<F>::pack2(x[8 * s + 0], x[8 * s + 1]), h16<F>::pack2(x[8 * s + 2], x[8 * s + 3]),
                               h16<F>::pack2(x[8 * s + 4], x[8 * s + 5]), h16<F>::pack2(x[8 * s + 6], x[8 * s + 7]));
    return (u);
}

__device__ __forceinline__ uint4 tr_frag(const unsigned char *img, int row0, int dbyte)
{
    struct { s4_t a, b; } f;
    f.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + v_off(row0, dbyte)));
    f.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + v_off(row0 + 8, dbyte)));
    return __builtin_bit_cast(uint4, f);
}

// sum over the eight bf16 pairs of two 16-byte chunks (fp32)
template <typename F>
__device__ __forceinline__ float dot8_bf16(uint4 a, uint4 b)
{
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc = fmaf(h16<F>::lo(aw[i]), h16<F>::lo(bw[i]), acc);
        acc = fmaf(h16<F>::hi(aw[i]), h16<F>::hi(bw[i]), acc);
    }
    return acc;
}

constexpr int QT = 32;                       // query rows per staged tile in the dK/dV kernel
constexpr int QTILE = QT * 128;              // bytes per 32-row image

constexpr int DKV_SMEM = 2 * (4 * QTILE + 256);   // per stage: Q row image, Q tr image, dO row image, dO tr image (4 KiB each) + lse2[32] + delta[32]

// INLINE_DELTA: delta[q] = sum_d out[q, d] * dout[q, d] is computed while the q tile is staged (the eight threads that stage a
// row hold its eight 16-byte chunks) instead of being read from the array a separate kernel filled: one node less in the
// prompt chain per layer.
template <typename F, bool CAUSAL, bool INLINE_DELTA, bool WHOLE_PREFIX = false>
__device__ __forceinline__ void attn_bwd_dkv_body(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out, const bf16_t *__restrict__ dout,
                                                  const float *__restrict__ lse, const float *__restrict__ delta,
                                                  bf16_t *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C,
                                                  float *__restrict__ part, unsigned char *smem, const int bx, const int by)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = by, b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD, os = (int64_t)H * HD;
    const int T = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);
    // WHOLE_PREFIX: the prefix sequence's workgroup (b == C) walks ALL physical rows as queries -- every prompt's own rows attend
    // to every shared key, the shared rows causally among themselves (key <= row covers both: own rows sit at >= P) -- and
    // writes dK / dV of the shared keys itself, in one fixed order; the prompts' workgroups then skip their shared keys.  No
    // fp32 partials, no reduction kernel behind this one.
    const bool pfx = WHOLE_PREFIX && P > 0 && b == C;
    const int Tq = pfx ? P + C * (Tfull - P) : T;
    const bf16_t *qb = qkv + head * HD;
    const bf16_t *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const bf16_t *gb = dout + head * HD;
    const int k0 = bx * 128 + w * 32;
    const int key = k0 + r;
    const float c = scale * 1.4426950408889634f;

    uint4 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        uint4 a = make_uint4(0, 0, 0, 0), v = make_uint4(0, 0, 0, 0);
        if (key < T) {
            a = *reinterpret_cast<const uint4 *>(kb + am_row(Tfull, P, b, key) * rs + 16 * kk + 8 * h);
            v = *reinterpret_cast<const uint4 *>(vb + am_row(Tfull, P, b, key) * rs + 16 * kk + 8 * h);
        }
        kf[kk] = (a);
        vf[kk] = (v);
    }
    f32x16_t dvt[2], dkt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dvt[i][e] = 0.f; dkt[i][e] = 0.f; }

    const int nqt = (Tq + QT - 1) / QT;
    // queries before the block's first key see none of it; queries below q_lo belong to the prefix sequence, not to this one
    const int qt0 = max(CAUSAL ? (int)(bx * 128) / QT : 0, q_lo / QT);
    uint4 sq, sg, so = make_uint4(0, 0, 0, 0);
    float sl = 0.f, sd = 0.f, sdr = 0.f;
    const int srow = threadIdx.x >> 3, sch = threadIdx.x & 7;   // staging: one 16-B chunk of Q and of dO per thread
    auto load_tile = [&](int qt) {
        const int q = qt * QT + srow;
        sq = sg = so = make_uint4(0, 0, 0, 0);
        if (q < Tq) {
            const int64_t qr = pfx ? (int64_t)q : am_row(Tfull, P, b, q);
            sq = *reinterpret_cast<const uint4 *>(qb + qr * rs + sch * 8);
            sg = *reinterpret_cast<const uint4 *>(gb + qr * os + sch * 8);
            if constexpr (INLINE_DELTA) so = *reinterpret_cast<const uint4 *>(out + head * HD + qr * os + sch * 8);
        }
        if constexpr (INLINE_DELTA) {
            // the row's eight chunks sit in eight consecutive lanes: three xor steps leave the row sum in all of them
            float d = dot8_bf16<F>(sg, so);
            d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
            const bool own = q < Tq && q >= q_lo;
            sdr = own ? d : 0.f;
            if (threadIdx.x < QT) {
                const int q2 = qt * QT + threadIdx.x;
                const bool own2 = q2 < Tq && q2 >= q_lo;
                sl = own2 ? lse[pfx ? (int64_t)q2 * H + head : am_stat(Tfull, P, H, b, head, q2)] * 1.4426950408889634f : INFINITY;
            }
        } else if (threadIdx.x < QT) {
            const int q2 = qt * QT + threadIdx.x;
            const bool own = q2 < T && q2 >= q_lo;                  // +inf -> p = 0 for padded rows and for rows this sequence does not own
            sl = own ? lse[am_stat(Tfull, P, H, b, head, q2)] * 1.4426950408889634f : INFINITY;
            sd = own ? delta[am_stat(Tfull, P, H, b, head, q2)] : 0.f;
        }
    };
    auto write_tile = [&](int buf) {
        unsigned char *base = smem + buf * (4 * QTILE + 256);
        *reinterpret_cast<uint4 *>(base + 0 * QTILE + k_off(srow, sch)) = sq;
        *reinterpret_cast<uint4 *>(base + 1 * QTILE + v_off(srow, sch * 16)) = sq;
        *reinterpret_cast<uint4 *>(base + 2 * QTILE + k_off(srow, sch)) = sg;
        *reinterpret_cast<uint4 *>(base + 3 * QTILE + v_off(srow, sch * 16)) = sg;
        if (threadIdx.x < QT) reinterpret_cast<float *>(base + 4 * QTILE)[threadIdx.x] = sl;
        if constexpr (INLINE_DELTA) {
            if (sch == 0) reinterpret_cast<float *>(base + 4 * QTILE + 128)[srow] = sdr;
        } else if (threadIdx.x < QT) {
            reinterpret_cast<float *>(base + 4 * QTILE + 128)[threadIdx.x] = sd;
        }
    };
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * (g >> 1) + tq;
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;

    if (qt0 < nqt) {
        load_tile(qt0);
        write_tile(0);
    }
    __syncthreads();
    for (int qt = qt0; qt < nqt; ++qt) {
        const int cur = (qt - qt0) & 1;
        if (qt + 1 < nqt) load_tile(qt + 1);
        const bool active = k0 < T && (!CAUSAL || qt * QT + QT - 1 >= k0);       // wave-uniform
        if (active) {
            const unsigned char *base = smem + cur * (4 * QTILE + 256);
            const float *L2 = reinterpret_cast<const float *>(base + 4 * QTILE);
            const float *DL = reinterpret_cast<const float *>(base + 4 * QTILE + 128);
            f32x16_t sa, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint4 qa = (*reinterpret_cast<const uint4 *>(base + 0 * QTILE + k_off(r, 2 * kk + h)));
                const uint4 ga = (*reinterpret_cast<const uint4 *>(base + 2 * QTILE + k_off(r, 2 * kk + h)));
                sa = h16<F>::mfma32(qa, kf[kk], sa);
                dp = h16<F>::mfma32(ga, vf[kk], dp);
            }
            const bool diag = CAUSAL && qt * QT < k0 + 32;        // some (q, key) pairs of this tile are masked
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 l4 = *reinterpret_cast<const float4 *>(L2 + 8 * gq + 4 * h);
                const float4 d4 = *reinterpret_cast<const float4 *>(DL + 8 * gq + 4 * h);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * gq + j;
                    float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], c, -lv[j]));
                    if (diag && key > qt * QT + 8 * gq + 4 * h + j) pv = 0.f;
                    sa[e] = pv;
                    dp[e] = pv * (dp[e] - dv[j]) * scale;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 pf = pack8<F>(sa, s), df = pack8<F>(dp, s);
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile) {
                    const uint4 gt = tr_frag(base + 3 * QTILE, 16 * s + tr_row, tr_dbyte + 64 * dtile);   // dO^T
                    dvt[dtile] = h16<F>::mfma32(gt, pf, dvt[dtile]);
                    const uint4 qt_ = tr_frag(base + 1 * QTILE, 16 * s + tr_row, tr_dbyte + 64 * dtile);  // Q^T
                    dkt[dtile] = h16<F>::mfma32(qt_, df, dkt[dtile]);
                }
            }
        }
        if (qt + 1 < nqt) write_tile(cur ^ 1);
        __syncthreads();
    }
    if (WHOLE_PREFIX && P > 0 && key < P && !pfx) {
        // (a shared key seen from a prompt: the prefix workgroup owns it)
    } else if (!WHOLE_PREFIX && key < T && P > 0 && key < P) {
        // a SHARED key: this virtual sequence's contribution goes to its fp32 partial slot [b][key][K | V][H * HD]; the
        // slots are folded in a fixed order by attn_prefix_reduce (no atomics)
        float *pk = part + (((int64_t)b * P + key) * 2) * os + head * HD, *pv = pk + os;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d = 32 * dtile + 8 * gq + 4 * h;
                *reinterpret_cast<float4 *>(pk + d) = make_float4(dkt[dtile][4 * gq], dkt[dtile][4 * gq + 1], dkt[dtile][4 * gq + 2], dkt[dtile][4 * gq + 3]);
                *reinterpret_cast<float4 *>(pv + d) = make_float4(dvt[dtile][4 * gq], dvt[dtile][4 * gq + 1], dvt[dtile][4 * gq + 2], dvt[dtile][4 * gq + 3]);
            }
    } else if (key < T) {
        bf16_t *ok = dqkv + am_row(Tfull, P, b, key) * rs + head * HD + H * HD;
        bf16_t *ov = ok + H * HD;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d = 32 * dtile + 8 * gq + 4 * h;
                *reinterpret_cast<uint2 *>(ok + d) = make_uint2(h16<F>::pack2(dkt[dtile][4 * gq], dkt[dtile][4 * gq + 1]),
                                                                 h16<F>::pack2(dkt[dtile][4 * gq + 2], dkt[dtile][4 * gq + 3]));
                *reinterpret_cast<uint2 *>(ov + d) = make_uint2(h16<F>::pack2(dvt[dtile][4 * gq], dvt[dtile][4 * gq + 1]),
                                                                 h16<F>::pack2(dvt[dtile][4 * gq + 2], dvt[dtile][4 * gq + 3]));
            }
    }
}

template <bool CAUSAL, typename F>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ dout,
                                                         const float *__restrict__ lse, const float *__restrict__ delta,
                                                         bf16_t *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C,
                                                         float *__restrict__ part)
{
    __shared__ __align__(16) unsigned char smem[DKV_SMEM];
    attn_bwd_dkv_body<F, CAUSAL, false>(qkv, nullptr, dout, lse, delta, dqkv, Tfull, H, scale, P, C, part, smem, blockIdx.x, blockIdx.y);
}

constexpr int DQ_SMEM = 2 * 3 * TILE;            // per stage: K row image, K tr image, V row image

template <typename F, bool CAUSAL, bool INLINE_DELTA>
__device__ __forceinline__ void attn_bwd_dq_body(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out, const bf16_t *__restrict__ dout,
                                                 const float *__restrict__ lse, const float *__restrict__ delta,
                                                 bf16_t *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C,
                                                 unsigned char *smem, const int bx, const int by)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = by, b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD, os = (int64_t)H * HD;
    const int T = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);
    const bf16_t *qb = qkv + head * HD;
    const bf16_t *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const bf16_t *gb = dout + head * HD;
    const int q0 = bx * QB + w * 32;
    const int qrow = q0 + r;
    const float c = scale * 1.4426950408889634f;

    uint4 qf[4], gf[4];
    float dsum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        uint4 a = make_uint4(0, 0, 0, 0), v = make_uint4(0, 0, 0, 0);
        if (qrow < T) {
            a = *reinterpret_cast<const uint4 *>(qb + am_row(Tfull, P, b, qrow) * rs + 16 * kk + 8 * h);
            v = *reinterpret_cast<const uint4 *>(gb + am_row(Tfull, P, b, qrow) * os + 16 * kk + 8 * h);
            if constexpr (INLINE_DELTA)      // this lane's half of the row (the other half sits in lane ^ 32)
                dsum += dot8_bf16<F>(v, *reinterpret_cast<const uint4 *>(out + head * HD + am_row(Tfull, P, b, qrow) * os + 16 * kk + 8 * h));
        }
        qf[kk] = (a);
        gf[kk] = (v);
    }
    const bool own = qrow < T && qrow >= q_lo;
    const float l2 = own ? lse[am_stat(Tfull, P, H, b, head, qrow)] * 1.4426950408889634f : INFINITY;
    float dl;
    if constexpr (INLINE_DELTA) {
        const float ds = lane_xor32_sum(dsum);            // (both halves of the pair run it: same row, same `own`)
        dl = own ? ds : 0.f;
    } else dl = own ? delta[am_stat(Tfull, P, H, b, head, qrow)] : 0.f;

    uint4 sk[2], sv[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int kx = kt * KVT + (cidx >> 3), ch = cidx & 7;
            sk[i] = sv[i] = make_uint4(0, 0, 0, 0);
            if (kx < T) {
                const int64_t kr = am_row(Tfull, P, b, kx) * rs;
                sk[i] = *reinterpret_cast<const uint4 *>(kb + kr + ch * 8);
                sv[i] = *reinterpret_cast<const uint4 *>(vb + kr + ch * 8);
            }
        }
    };
    auto write_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int row = cidx >> 3, ch = cidx & 7;
            unsigned char *base = smem + buf * 3 * TILE;
            *reinterpret_cast<uint4 *>(base + 0 * TILE + k_off(row, ch)) = sk[i];
            *reinterpret_cast<uint4 *>(base + 1 * TILE + v_off(row, ch * 16)) = sk[i];
            *reinterpret_cast<uint4 *>(base + 2 * TILE + k_off(row, ch)) = sv[i];
        }
    };
    const int q_hi = min(T, (int)(bx + 1) * QB) - 1;
    const int nkt = CAUSAL ? min((T + KVT - 1) / KVT, q_hi / KVT + 1) : (T + KVT - 1) / KVT;
    f32x16_t dqt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqt[i][e] = 0.f;
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * (g >> 1) + tq;
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;

    load_tile(0);
    write_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const bool active = q0 < T && (!CAUSAL || kt * KVT <= q0 + 31);
        if (active) {
            const unsigned char *base = smem + cur * 3 * TILE;
            const bool need_mask = (kt * KVT + KVT > T) || (CAUSAL && kt * KVT + KVT - 1 > q0);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16_t sa, dp;
#pragma unroll
                for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const uint4 ka = (*reinterpret_cast<const uint4 *>(base + 0 * TILE + k_off(32 * sub + r, 2 * kk + h)));
                    const uint4 va = (*reinterpret_cast<const uint4 *>(base + 2 * TILE + k_off(32 * sub + r, 2 * kk + h)));
                    sa = h16<F>::mfma32(ka, qf[kk], sa);     // S^T  [key][q]
                    dp = h16<F>::mfma32(va, gf[kk], dp);     // dP^T [key][q]
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], c, -l2));
                    if (need_mask) {
                        const int kx = kt * KVT + 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (kx >= T || (CAUSAL && kx > qrow)) pv = 0.f;
                    }
                    dp[e] = pv * (dp[e] - dl) * scale;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 df = pack8<F>(dp, s);
#pragma unroll
                    for (int dtile = 0; dtile < 2; ++dtile) {
                        const uint4 kt_ = tr_frag(base + 1 * TILE, 32 * sub + 16 * s + tr_row, tr_dbyte + 64 * dtile);   // K^T
                        dqt[dtile] = h16<F>::mfma32(kt_, df, dqt[dtile]);
                    }
                }
            }
        }
        if (kt + 1 < nkt) write_tile(cur ^ 1);
        __syncthreads();
    }
    if (own) {
        bf16_t *oq = dqkv + am_row(Tfull, P, b, qrow) * rs + head * HD;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<uint2 *>(oq + 32 * dtile + 8 * gq + 4 * h) =
                    make_uint2(h16<F>::pack2(dqt[dtile][4 * gq], dqt[dtile][4 * gq + 1]),
                               h16<F>::pack2(dqt[dtile][4 * gq + 2], dqt[dtile][4 * gq + 3]));
    }
}

template <bool CAUSAL, typename F>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_mfma(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ dout,
                                                        const float *__restrict__ lse, const float *__restrict__ delta,
                                                        bf16_t *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C)
{
    __shared__ __align__(16) unsigned char smem[DQ_SMEM];
    attn_bwd_dq_body<F, CAUSAL, false>(qkv, nullptr, dout, lse, delta, dqkv, Tfull, H, scale, P, C, smem, blockIdx.x, blockIdx.y);
}

// One launch for the whole backward of a SHORT sequence (T <= 128: one key block and one query block per (sequence, head);
// the CLIP text tower, 37 positions): blockIdx.x == 0 computes dK / dV, blockIdx.x == 1 computes dQ, both form delta
// themselves -- attn_delta + dkv + dq (three nodes of the prompt chain per layer) become one.  (WHOLE_PREFIX, which would also
// absorb attn_prefix_reduce, is compiled out: the prefix workgroup's walk over all 817 rows is 26 dependent tile iterations
// of ~1.1 us each -- the next tile's loads are only one iteration ahead -- and made the launch ~29 us instead of ~3 + a 4.5 us
// reduction: prompt chain alone 1.97 -> 2.33 ms.)
#ifndef PPT_SHORT_OCC
#define PPT_SHORT_OCC 2
#endif
template <bool CAUSAL, typename F>
__global__ __launch_bounds__(256, PPT_SHORT_OCC) void attn_bwd_short_mfma(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out,
                                                           const bf16_t *__restrict__ dout, const float *__restrict__ lse,
                                                           bf16_t *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C,
                                                           float *__restrict__ part, int prio)
{
    PPT_PRIO(prio);
    __shared__ __align__(16) unsigned char smem[DQ_SMEM > DKV_SMEM ? DQ_SMEM : DKV_SMEM];
    // gridDim.x == 2: one workgroup per role; == 1: ONE workgroup runs both roles back to back (half the workgroups: with two
    // 208-VGPR workgroups per CU, 656 role workgroups of the 41 x 8 (sequence, head) pairs were two rounds on 512 slots)
#ifdef PPT_SHORT_ROLE   // (diagnostic build: one role only)
    if (PPT_SHORT_ROLE == 0) { attn_bwd_dkv_body<F, CAUSAL, true, false>(qkv, out, dout, lse, nullptr, dqkv, Tfull, H, scale, P, C, part, smem, 0, blockIdx.y); return; }
    if (PPT_SHORT_ROLE == 1) { attn_bwd_dq_body<F, CAUSAL, true>(qkv, out, dout, lse, nullptr, dqkv, Tfull, H, scale, P, C, smem, 0, blockIdx.y); return; }
    if (PPT_SHORT_ROLE == 2) return;
#endif
    if (gridDim.x == 1 || blockIdx.x == 0)
        attn_bwd_dkv_body<F, CAUSAL, true, false>(qkv, out, dout, lse, nullptr, dqkv, Tfull, H, scale, P, C, part, smem, 0, blockIdx.y);
    if (gridDim.x == 1) __syncthreads();
    if (gridDim.x == 1 || blockIdx.x == 1)
        attn_bwd_dq_body<F, CAUSAL, true>(qkv, out, dout, lse, nullptr, dqkv, Tfull, H, scale, P, C, smem, 0, blockIdx.y);
}


// The whole backward of a TINY sequence (T <= 64: the CLIP text tower under prompt learning, 37 positions of which 17 are the shared
// prefix) as ONE pass: attn_bwd_short_mfma above runs the two roles back to back, each with its own trip to global memory (dK / dV:
// K, V into registers, Q / dO tiles staged; dQ: Q, dO into registers, K / V tiles staged) and with two of its four waves idle
// (37 keys = two 32-key waves) -- 9.4 + 8.0 us alone, 13 us together, for 0.3 MFLOP per workgroup.  Here every matrix of the
// (sequence, head) goes to LDS ONCE (row image, and the transposed-read image where a product wants it), delta = rowsum(dO * O) is
// formed while the rows are staged, and then waves 0-1 run the dK / dV role (32 keys each) WHILE waves 2-3 run the dQ role (32
// queries each): one global round trip, four busy waves.  Same products in the same order as the two bodies above.
constexpr int TINY_SMEM = 7 * TILE + 512;
template <bool CAUSAL, typename F>
__global__ __launch_bounds__(256, 2) void attn_bwd_tiny_mfma(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out,
                                                          const bf16_t *__restrict__ dout, const float *__restrict__ lse,
                                                          bf16_t *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C,
                                                          float *__restrict__ part, int prio)
{
    PPT_PRIO(prio);
    __shared__ __align__(16) unsigned char smem[TINY_SMEM];
    unsigned char *Qr = smem, *Qt = smem + TILE, *Gr = smem + 2 * TILE, *Gt = smem + 3 * TILE, *Kr = smem + 4 * TILE, *Kt = smem + 5 * TILE,
                  *Vr = smem + 6 * TILE;
    float *L2 = reinterpret_cast<float *>(smem + 7 * TILE), *DL = L2 + 64;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD, os = (int64_t)H * HD;
    const int T = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);
    const float c = scale * 1.4426950408889634f;

    // ---- stage: thread = one 16-byte chunk of two rows (rows t >> 3 and 32 + (t >> 3)) of Q, K, V, dO (and O for delta)
    {
        uint4 q[2], k[2], v[2], g[2];
        float dl[2], l2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 32 * i + (threadIdx.x >> 3), ch = threadIdx.x & 7;
            q[i] = k[i] = v[i] = g[i] = make_uint4(0, 0, 0, 0);
            uint4 o = make_uint4(0, 0, 0, 0);
            l2[i] = INFINITY;
            const bool own = row < T && row >= q_lo;
            if (row < T) {
                const int64_t pr = am_row(Tfull, P, b, row);
                const bf16_t *qp = qkv + pr * rs + head * HD + ch * 8;
                q[i] = *reinterpret_cast<const uint4 *>(qp);
                k[i] = *reinterpret_cast<const uint4 *>(qp + H * HD);
                v[i] = *reinterpret_cast<const uint4 *>(qp + 2 * H * HD);
                g[i] = *reinterpret_cast<const uint4 *>(dout + pr * os + head * HD + ch * 8);
                o = *reinterpret_cast<const uint4 *>(out + pr * os + head * HD + ch * 8);
                if (own && ch == 0) l2[i] = lse[am_stat(Tfull, P, H, b, head, row)] * 1.4426950408889634f;
            }
            float d = dot8_bf16<F>(g[i], o);
            d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
            dl[i] = own ? d : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 32 * i + (threadIdx.x >> 3), ch = threadIdx.x & 7;
            *reinterpret_cast<uint4 *>(Qr + k_off(row, ch)) = q[i];
            *reinterpret_cast<uint4 *>(Qt + v_off(row, ch * 16)) = q[i];
            *reinterpret_cast<uint4 *>(Gr + k_off(row, ch)) = g[i];
            *reinterpret_cast<uint4 *>(Gt + v_off(row, ch * 16)) = g[i];
            *reinterpret_cast<uint4 *>(Kr + k_off(row, ch)) = k[i];
            *reinterpret_cast<uint4 *>(Kt + v_off(row, ch * 16)) = k[i];
            *reinterpret_cast<uint4 *>(Vr + k_off(row, ch)) = v[i];
            if (ch == 0) { L2[row] = l2[i]; DL[row] = dl[i]; }
        }
    }
    __syncthreads();
    const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * (g4 >> 1) + tq;
    const int tr_dbyte = (16 * (g4 & 1) + 4 * tp) * 2;

    if (w < 2) {
        // ---- dK / dV of keys 32 w .. 32 w + 31 (attn_bwd_dkv_body with K, V, Q, dO out of LDS)
        const int k0 = 32 * w, key = k0 + r;
        if (k0 >= T) return;
        uint4 kf[4], vf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[kk] = *reinterpret_cast<const uint4 *>(Kr + k_off(key, 2 * kk + h));
            vf[kk] = *reinterpret_cast<const uint4 *>(Vr + k_off(key, 2 * kk + h));
        }
        f32x16_t dvt[2], dkt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dvt[i][e] = 0.f; dkt[i][e] = 0.f; }
        const int nqt = (T + QT - 1) / QT;
        for (int qt = q_lo / QT; qt < nqt; ++qt) {
            if (CAUSAL && qt * QT + QT - 1 < k0) continue;
            f32x16_t sa, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint4 qa = *reinterpret_cast<const uint4 *>(Qr + k_off(QT * qt + r, 2 * kk + h));
                const uint4 ga = *reinterpret_cast<const uint4 *>(Gr + k_off(QT * qt + r, 2 * kk + h));
                sa = h16<F>::mfma32(qa, kf[kk], sa);
                dp = h16<F>::mfma32(ga, vf[kk], dp);
            }
            const bool diag = CAUSAL && qt * QT < k0 + 32;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 l4 = *reinterpret_cast<const float4 *>(L2 + QT * qt + 8 * gq + 4 * h);
                const float4 d4 = *reinterpret_cast<const float4 *>(DL + QT * qt + 8 * gq + 4 * h);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * gq + j;
                    float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], c, -lv[j]));
                    if (diag && key > qt * QT + 8 * gq + 4 * h + j) pv = 0.f;
                    sa[e] = pv;
                    dp[e] = pv * (dp[e] - dv[j]) * scale;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 pf = pack8<F>(sa, s), df = pack8<F>(dp, s);
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile) {
                    const uint4 gt = tr_frag(Gt, QT * qt + 16 * s + tr_row, tr_dbyte + 64 * dtile);   // dO^T
                    dvt[dtile] = h16<F>::mfma32(gt, pf, dvt[dtile]);
                    const uint4 qt_ = tr_frag(Qt, QT * qt + 16 * s + tr_row, tr_dbyte + 64 * dtile);  // Q^T
                    dkt[dtile] = h16<F>::mfma32(qt_, df, dkt[dtile]);
                }
            }
        }
        if (key < T && P > 0 && key < P) {
            // a SHARED key: the fp32 partial slot of this virtual sequence (folded in a fixed order by attn_prefix_reduce)
            float *pk = part + (((int64_t)b * P + key) * 2) * os + head * HD, *pv = pk + os;
#pragma unroll
            for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int d = 32 * dtile + 8 * gq + 4 * h;
                    *reinterpret_cast<float4 *>(pk + d) = make_float4(dkt[dtile][4 * gq], dkt[dtile][4 * gq + 1], dkt[dtile][4 * gq + 2], dkt[dtile][4 * gq + 3]);
                    *reinterpret_cast<float4 *>(pv + d) = make_float4(dvt[dtile][4 * gq], dvt[dtile][4 * gq + 1], dvt[dtile][4 * gq + 2], dvt[dtile][4 * gq + 3]);
                }
        } else if (key < T) {
            bf16_t *ok = dqkv + am_row(Tfull, P, b, key) * rs + head * HD + H * HD;
            bf16_t *ov = ok + H * HD;
#pragma unroll
            for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int d = 32 * dtile + 8 * gq + 4 * h;
                    *reinterpret_cast<uint2 *>(ok + d) = make_uint2(h16<F>::pack2(dkt[dtile][4 * gq], dkt[dtile][4 * gq + 1]),
                                                                     h16<F>::pack2(dkt[dtile][4 * gq + 2], dkt[dtile][4 * gq + 3]));
                    *reinterpret_cast<uint2 *>(ov + d) = make_uint2(h16<F>::pack2(dvt[dtile][4 * gq], dvt[dtile][4 * gq + 1]),
                                                                     h16<F>::pack2(dvt[dtile][4 * gq + 2], dvt[dtile][4 * gq + 3]));
                }
        }
    } else {
        // ---- dQ of queries 32 (w - 2) .. + 31 (attn_bwd_dq_body with Q, dO, K, V out of LDS; one 64-key tile)
        const int q0 = 32 * (w - 2), qrow = q0 + r;
        if (q0 >= T || q0 + 31 < q_lo) return;
        uint4 qf[4], gf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[kk] = *reinterpret_cast<const uint4 *>(Qr + k_off(qrow, 2 * kk + h));
            gf[kk] = *reinterpret_cast<const uint4 *>(Gr + k_off(qrow, 2 * kk + h));
        }
        const bool own = qrow < T && qrow >= q_lo;
        const float l2 = L2[qrow], dl = DL[qrow];
        f32x16_t dqt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) dqt[i][e] = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (32 * sub >= T || (CAUSAL && 32 * sub > q0 + 31)) continue;      // (no unmasked key in this half: it would add zeros)
            f32x16_t sa, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint4 ka = *reinterpret_cast<const uint4 *>(Kr + k_off(32 * sub + r, 2 * kk + h));
                const uint4 va = *reinterpret_cast<const uint4 *>(Vr + k_off(32 * sub + r, 2 * kk + h));
                sa = h16<F>::mfma32(ka, qf[kk], sa);     // S^T  [key][q]
                dp = h16<F>::mfma32(va, gf[kk], dp);     // dP^T [key][q]
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], c, -l2));
                const int kx = 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (kx >= T || (CAUSAL && kx > qrow)) pv = 0.f;
                dp[e] = pv * (dp[e] - dl) * scale;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 df = pack8<F>(dp, s);
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile) {
                    const uint4 kt_ = tr_frag(Kt, 32 * sub + 16 * s + tr_row, tr_dbyte + 64 * dtile);   // K^T
                    dqt[dtile] = h16<F>::mfma32(kt_, df, dqt[dtile]);
                }
            }
        }
        if (own) {
            bf16_t *oq = dqkv + am_row(Tfull, P, b, qrow) * rs + head * HD;
#pragma unroll
            for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<uint2 *>(oq + 32 * dtile + 8 * gq + 4 * h) =
                        make_uint2(h16<F>::pack2(dqt[dtile][4 * gq], dqt[dtile][4 * gq + 1]),
                                   h16<F>::pack2(dqt[dtile][4 * gq + 2], dqt[dtile][4 * gq + 3]));
        }
    }
}

}  // namespace

// fmt = PPT_BF16 or PPT_F16 (the 16-bit operand format of qkv / out / dout / dqkv) for the functions below
extern "C" int ppt_attention_fwd_mfma_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H, float scale,
                                           int causal, int P, int fmt, hipStream_t s)
{
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return ppt_attention_fwd_quad_bf16(qkv, out, lse, Bt, T, H, scale, causal, P, fmt, s);
    const float c = scale * 1.4426950408889634f;
    const int prio = ppt_get_wave_priority();
    // the ViT shape with enough (batch, head) pairs to fill the chip: K / V resident in LDS (attn_fwd_resident)
    static const bool resident_ok = getenv("PPT_ATTN_RESIDENT") == nullptr || atoi(getenv("PPT_ATTN_RESIDENT")) != 0;
    if (resident_ok && !causal && P == 0 && (T % KVT) == 1 && T - 1 >= 6 * KVT && T - 1 <= RES_MAXK && Bt * H >= 128) {
#define PPT_RES_ATTR(FF, NN) (void)hipFuncSetAttribute((const void *)attn_fwd_resident<FF, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS)
        static const bool attr = [] {
            PPT_RES_ATTR(bf16_t, 6); PPT_RES_ATTR(bf16_t, 7); PPT_RES_ATTR(bf16_t, 8);
            PPT_RES_ATTR(f16_t, 6); PPT_RES_ATTR(f16_t, 7); PPT_RES_ATTR(f16_t, 8);
            return true;
        }();
        (void)attr;
#undef PPT_RES_ATTR
#define PPT_RES_LAUNCH(FF, NN) hipLaunchKernelGGL((attn_fwd_resident<FF, NN>), dim3(Bt * H), dim3(512), RES_LDS, s, (const bf16_t *)qkv, (bf16_t *)out, lse, T, H, c, prio)
        const int n = (T - 1) / KVT;
        if (fmt == PPT_F16) { if (n == 6) PPT_RES_LAUNCH(f16_t, 6); else if (n == 7) PPT_RES_LAUNCH(f16_t, 7); else PPT_RES_LAUNCH(f16_t, 8); }
        else { if (n == 6) PPT_RES_LAUNCH(bf16_t, 6); else if (n == 7) PPT_RES_LAUNCH(bf16_t, 7); else PPT_RES_LAUNCH(bf16_t, 8); }
#undef PPT_RES_LAUNCH
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    dim3 grid((T + QB - 1) / QB, (Bt + (P > 0)) * H);
    const dim3 block(256);
    static const int xcd_map = getenv("PPT_ATTN_XCD_MAP") == nullptr || atoi(getenv("PPT_ATTN_XCD_MAP")) != 0;
    if (fmt == PPT_F16) {
        if (causal) hipLaunchKernelGGL((attn_fwd_mfma<true, f16_t>), grid, block, 0, s, (const bf16_t *)qkv, (bf16_t *)out, lse, T, H, c, P, Bt, prio, xcd_map);
        else hipLaunchKernelGGL((attn_fwd_mfma<false, f16_t>), grid, block, 0, s, (const bf16_t *)qkv, (bf16_t *)out, lse, T, H, c, P, Bt, prio, xcd_map);
    } else {
        if (causal) hipLaunchKernelGGL((attn_fwd_mfma<true, bf16_t>), grid, block, 0, s, (const bf16_t *)qkv, (bf16_t *)out, lse, T, H, c, P, Bt, prio, xcd_map);
        else hipLaunchKernelGGL((attn_fwd_mfma<false, bf16_t>), grid, block, 0, s, (const bf16_t *)qkv, (bf16_t *)out, lse, T, H, c, P, Bt, prio, xcd_map);
    }
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// dq / dk / dv of the bf16 path; `delta` must already hold rowsum(dO * O) (attention.hip: attn_delta).  P > 0 (prefix-shared
// layout, attn_rowmap.h): `part` [Bt + 1, P, 2, H * 64] f32 receives the per-sequence dK / dV of the shared rows; the caller
// folds it (attention.hip: attn_prefix_reduce).
extern "C" int ppt_attention_bwd_short_mfma_bf16(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int Bt, int T,
                                                 int H, float scale, int causal, int P, float *part, int fmt, hipStream_t s)
{
    if (T > 128) return PPT_EUNSUPPORTED;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 7) || ((uintptr_t)part & 15)) return PPT_EUNSUPPORTED;
    static const int both = [] { const char *e = getenv("PPT_ATTN_SHORT_BOTH"); return e ? atoi(e) : 1; }();
    static const int tiny = [] { const char *e = getenv("PPT_ATTN_TINY"); return e ? atoi(e) : 1; }();
    const int pairs = (Bt + (P > 0)) * H;
    if (tiny && T <= 64 && !((uintptr_t)dout & 15)) {
        const int prio = ppt_get_wave_priority();
#define PPT_LAUNCH_TINY(CA, TT) hipLaunchKernelGGL((attn_bwd_tiny_mfma<CA, TT>), dim3(1, pairs), dim3(256), 0, s, (const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, T, H, scale, P, Bt, part, prio)
        if (fmt == PPT_F16) { if (causal) PPT_LAUNCH_TINY(true, f16_t); else PPT_LAUNCH_TINY(false, f16_t); }
        else { if (causal) PPT_LAUNCH_TINY(true, bf16_t); else PPT_LAUNCH_TINY(false, bf16_t); }
#undef PPT_LAUNCH_TINY
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    dim3 grid(both && 2 * pairs > 512 ? 1 : 2, pairs);    // (two roles per workgroup once the role workgroups would not fit in one round)
    const int prio = ppt_get_wave_priority();
#define PPT_LAUNCH_SHORT(CA, TT) hipLaunchKernelGGL((attn_bwd_short_mfma<CA, TT>), grid, dim3(256), 0, s, (const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, T, H, scale, P, Bt, part, prio)
    if (fmt == PPT_F16) { if (causal) PPT_LAUNCH_SHORT(true, f16_t); else PPT_LAUNCH_SHORT(false, f16_t); }
    else { if (causal) PPT_LAUNCH_SHORT(true, bf16_t); else PPT_LAUNCH_SHORT(false, bf16_t); }
#undef PPT_LAUNCH_SHORT
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_attention_bwd_mfma_bf16(const void *qkv, const void *dout, const float *lse, const float *delta,
                                           void *dqkv, int Bt, int T, int H, float scale, int causal, int P, float *part,
                                           int fmt, hipStream_t s)
{
    if (((uintptr_t)qkv & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 7) || ((uintptr_t)part & 15)) return PPT_EUNSUPPORTED;
    dim3 grid((T + 127) / 128, (Bt + (P > 0)) * H);
#define PPT_LAUNCH_BWD(CA, TT) do { \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma<CA, TT>), grid, dim3(256), 0, s, (const bf16_t *)qkv, (const bf16_t *)dout, lse, delta, (bf16_t *)dqkv, T, H, scale, P, Bt, part); \
        hipLaunchKernelGGL((attn_bwd_dq_mfma<CA, TT>), grid, dim3(256), 0, s, (const bf16_t *)qkv, (const bf16_t *)dout, lse, delta, (bf16_t *)dqkv, T, H, scale, P, Bt); } while (0)
    if (fmt == PPT_F16) { if (causal) PPT_LAUNCH_BWD(true, f16_t); else PPT_LAUNCH_BWD(false, f16_t); }
    else { if (causal) PPT_LAUNCH_BWD(true, bf16_t); else PPT_LAUNCH_BWD(false, bf16_t); }
#undef PPT_LAUNCH_BWD
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
