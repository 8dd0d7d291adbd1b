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
constexpr int DEPTH = 16;                                     // weight ring: 1 KiB pieces in flight per wave
constexpr int PIECES = 64;                                    // pieces per unit per wave (16 k-steps x 4 column tiles)
constexpr int VIMG = 64 * 128;                                // wave-private V image (64 keys x 64 dims bf16)
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int v_off(int key, int dbyte) { return key * 128 + (dbyte ^ (((key >> 1) & 1) << 6)); }

// re-reads of bytes this workgroup stored earlier in the launch (qkv, x_mid, x): L1-bypassing loads, served by L2
__device__ __forceinline__ uint4 ld16_nt(const void *p)
{
    return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p)));
}
__device__ __forceinline__ float4 ldf4_nt(const float *p)
{
    return __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p)));
}

// LDS-only synchronisation: wait for this wave's LDS traffic, then the barrier -- NOT __syncthreads(), whose vmcnt(0) would
// drain the weight ring at every phase boundary
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the three points per layer where other waves' GLOBAL stores are read back (qkv, x_mid, x_out): stores complete, then barrier
__device__ __forceinline__ void vm_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8_t pack8(const f32x16_t &x, int s)
{
    const uint4 u = make_uint4(pack_bf16x2(x[8 * s + 0], x[8 * s + 1]), pack_bf16x2(x[8 * s + 2], x[8 * s + 3]),
                               pack_bf16x2(x[8 * s + 4], x[8 * s + 5]), pack_bf16x2(x[8 * s + 6], x[8 * s + 7]));
    return __builtin_bit_cast(bf16x8_t, u);
}

struct Acc { f32x4_t v[4][4]; };          // [row block of 16][column tile of 16]: D[n = 16 t + 4 kg + i][m = 16 rb + l15]

__device__ __forceinline__ void acc_zero(Acc &a)
{
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int t = 0; t < 4; ++t) a.v[rb][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// One unit: acc[64 rows x 64 columns of this wave] += A[64 x 512] (LDS image) . W_unit^T, the wave's 64 weight pieces coming
// out of the ring in order; every consumed slot is refilled with the piece DEPTH ahead in the wave's linear stream.
__device__ __forceinline__ void gemm_unit(Acc &acc, const unsigned char *img, bf16x8_t (&ring)[DEPTH], const bf16x8_t *&wnext,
                                          const int l15, const int kg)
{
    const unsigned char *a0 = img + l15 * HP + 16 * kg;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        bf16x8_t fa[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) fa[rb] = *reinterpret_cast<const bf16x8_t *>(a0 + rb * 16 * HP + 64 * s);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t b = ring[(4 * s + t) % DEPTH];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc.v[rb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, fa[rb], acc.v[rb][t], 0, 0, 0);
            ring[(4 * s + t) % DEPTH] = *wnext;
            wnext += 64;
        }
    }
}

// LayerNorm of the workgroup's rows (global fp32, row pitch WD) -> bf16 image in LDS; wave w takes rows w, w + 8, ...;
// two-pass mean / variance as norm.hip; rows >= nrow become zeros.  Statistics are saved when st_mean != nullptr.
template <bool NT>
__device__ __forceinline__ void ln_rows(const float *__restrict__ x, const float *__restrict__ gw, const float *__restrict__ gb,
                                        unsigned char *img, float *st_mean, float *st_rstd, int nrow, int w, int lane)
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
            const int r = w + 8 * (4 * half + i);
            v0[i] = v1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrow) {
                const float *xr = x + (size_t)r * WD + c;
                if (NT) { v0[i] = ldf4_nt(xr); v1[i] = ldf4_nt(xr + 4); }
                else { v0[i] = *reinterpret_cast<const float4 *>(xr); v1[i] = *reinterpret_cast<const float4 *>(xr + 4); }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = w + 8 * (4 * half + i);
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
            for (int j = 0; j < 8; ++j) o[j] = r < nrow ? (v[j] - mean) * rstd * g[j] + be[j] : 0.f;
            *reinterpret_cast<uint4 *>(img + r * HP + 16 * lane) =
                make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
            if (st_mean && lane == 0 && r < nrow) { st_mean[r] = mean; st_rstd[r] = rstd; }
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
    PPT_PRIO(p.prio);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int r31 = lane & 31, hh = lane >> 5;
    const int P = p.P, own = p.L - p.P;
    const int RW = P + p.NP * own;
    const int ng = min(p.NP, p.C - (int)blockIdx.x * p.NP);
    const int nrow = P + ng * own;                            // valid rows of this workgroup (<= 64)
    const size_t row0 = (size_t)blockIdx.x * RW;

    // the wave's weight stream
    const bf16x8_t *wnext = reinterpret_cast<const bf16x8_t *>(p.wfrag) + (size_t)w * p.layers * 12 * PIECES * 64 + lane;
    bf16x8_t ring[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) { ring[i] = *wnext; wnext += 64; }

    // causal / ownership mask of the attention, as bits over this lane's 32 score elements per 32-query tile
    unsigned amask[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        unsigned m = 0;
        const int qi = 32 * qt + r31;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kj = 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (visible(qi, kj, P, own)) m |= 1u << (16 * sub + e);
            }
        amask[qt] = m;
    }
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_key = 4 * (tg >> 1) + tq;
    const int tr_dbyte = (16 * (tg & 1) + 4 * tp) * 2;
    const float c = p.scale * LOG2E;

    for (int l = 0; l < p.layers; ++l) {
        const float *x_in = l == 0 ? p.x0 + row0 * WD : p.x + (size_t)(l - 1) * p.x_stride + row0 * WD;
        float *x_mid = p.xmid + (size_t)l * p.xm_stride + row0 * WD;
        float *x_out = p.x + (size_t)l * p.x_stride + row0 * WD;
        bf16_t *qkv = (bf16_t *)p.qkv + (size_t)l * p.qkv_stride + row0 * 3 * WD;
        float *st = p.stats ? p.stats + (size_t)l * p.stats_stride + row0 : nullptr;     // mean1 | rstd1 | mean2 | rstd2, each [rows]

        // ---- LN1 -> h (img1)
        if (l == 0) ln_rows<false>(x_in, p.ln1_w, p.ln1_b, img1, st, st ? st + p.rows : nullptr, nrow, w, lane);
        else ln_rows<true>(x_in, p.ln1_w + l * WD, p.ln1_b + l * WD, img1, st, st ? st + p.rows : nullptr, nrow, w, lane);
        lds_barrier();

        // ---- in_proj: three units (q, k, v); wave w computes head w's 64 columns of each
        for (int u = 0; u < 3; ++u) {
            Acc acc;
            acc_zero(acc);
            gemm_unit(acc, img1, ring, wnext, l15, kg);
            const float *bias = p.b_in + (size_t)l * 3 * WD + u * WD + 64 * w;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kg);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const int m = 16 * rb + l15;
                    if (m < nrow) {
                        const f32x4_t a = acc.v[rb][t];
                        *reinterpret_cast<uint2 *>(qkv + (size_t)m * 3 * WD + u * WD + 64 * w + 16 * t + 4 * kg) =
                            make_uint2(pack_bf16x2(a[0] + bv.x, a[1] + bv.y), pack_bf16x2(a[2] + bv.z, a[3] + bv.w));
                    }
                }
            }
        }
        vm_barrier();                                          // qkv complete in memory; h (img1) dead

        // ---- attention of head w over the workgroup's rows (scores never leave registers)
        {
            const bf16_t *qb = qkv + 64 * w, *kb = qb + WD, *vb = qb + 2 * WD;
            unsigned char *vimg = img1 + w * VIMG;
            bf16x8_t kf[2][4];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int row = 32 * sub + r31;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (row < nrow) v = ld16_nt(kb + (size_t)row * 3 * WD + 16 * kk + 8 * hh);
                    kf[sub][kk] = __builtin_bit_cast(bf16x8_t, v);
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cidx = lane + 64 * i, key = cidx >> 3, ch = cidx & 7;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (key < nrow) v = ld16_nt(vb + (size_t)key * 3 * WD + ch * 8);
                *reinterpret_cast<uint4 *>(vimg + v_off(key, ch * 16)) = v;
            }
            bf16_t *a_out = p.a ? (bf16_t *)p.a + (size_t)l * p.a_stride + row0 * WD + 64 * w : nullptr;
            float *lse = p.lse ? p.lse + (size_t)l * p.lse_stride + row0 * NH + w : nullptr;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qrow = 32 * qt + r31;
                bf16x8_t qf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (qrow < nrow) v = ld16_nt(qb + (size_t)qrow * 3 * WD + 16 * kk + 8 * hh);
                    qf[kk] = __builtin_bit_cast(bf16x8_t, v);
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
                float mx = -INFINITY;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((amask[qt] >> (16 * sub + e)) & 1u) mx = fmaxf(mx, sc[sub][e]);
                mx = xor32_max(mx);
                const float mn = mx * c;
                float psum = 0.f;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float pv = ((amask[qt] >> (16 * sub + e)) & 1u) ? __builtin_amdgcn_exp2f(fmaf(sc[sub][e], c, -mn)) : 0.f;
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
                const float inv = 1.0f / lt;
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const uint2 u2 = make_uint2(pack_bf16x2(ot[dtile][4 * gq + 0] * inv, ot[dtile][4 * gq + 1] * inv),
                                                    pack_bf16x2(ot[dtile][4 * gq + 2] * inv, ot[dtile][4 * gq + 3] * inv));
                        const int d = 32 * dtile + 8 * gq + 4 * hh;
                        *reinterpret_cast<uint2 *>(img2 + qrow * HP + (64 * w + d) * 2) = u2;
                        if (a_out && qrow < nrow) *reinterpret_cast<uint2 *>(a_out + (size_t)qrow * WD + d) = u2;
                    }
                if (lse && hh == 0 && qrow < nrow) lse[(size_t)qrow * NH] = (mn + __log2f(lt)) * 0.6931471805599453f;
            }
        }
        lds_barrier();                                         // attention output image complete; V images dead

        // ---- out_proj + residual -> x_mid
        {
            Acc acc;
            acc_zero(acc);
            gemm_unit(acc, img2, ring, wnext, l15, kg);
            const float *bias = p.b_out + (size_t)l * WD + 64 * w;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const int m = 16 * rb + l15;
                if (m < nrow) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int n = 64 * w + 16 * t + 4 * kg;
                        const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kg);
                        const float4 rv = l == 0 ? *reinterpret_cast<const float4 *>(x_in + (size_t)m * WD + n) : ldf4_nt(x_in + (size_t)m * WD + n);
                        const f32x4_t a = acc.v[rb][t];
                        *reinterpret_cast<float4 *>(x_mid + (size_t)m * WD + n) =
                            make_float4(a[0] + bv.x + rv.x, a[1] + bv.y + rv.y, a[2] + bv.z + rv.z, a[3] + bv.w + rv.w);
                    }
                }
            }
        }
        vm_barrier();                                          // x_mid complete in memory; attention image dead

        // ---- LN2 -> h2 (img2)
        ln_rows<true>(x_mid, p.ln2_w + l * WD, p.ln2_b + l * WD, img2, st ? st + 2 * p.rows : nullptr, st ? st + 3 * p.rows : nullptr, nrow, w, lane);
        lds_barrier();

        // ---- MLP: four hidden slabs of 512; c_fc + QuickGELU -> u (img1), c_proj accumulates over the slabs
        Acc accp;
        acc_zero(accp);
        bf16_t *pre = p.pre ? (bf16_t *)p.pre + (size_t)l * p.pre_stride + row0 * HID : nullptr;
        for (int j = 0; j < 4; ++j) {
            {
                Acc acc;
                acc_zero(acc);
                gemm_unit(acc, img2, ring, wnext, l15, kg);
                const float *bias = p.b_fc + (size_t)l * HID + j * WD + 64 * w;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kg);
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        const int m = 16 * rb + l15;
                        const f32x4_t a = acc.v[rb][t];
                        float v[4] = {a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w};
                        if (pre && m < nrow)
                            *reinterpret_cast<uint2 *>(pre + (size_t)m * HID + j * WD + 64 * w + 16 * t + 4 * kg) =
                                make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] / (1.0f + __expf(-1.702f * v[i]));      // QuickGELU (ULIP_models.py:30-32)
                        *reinterpret_cast<uint2 *>(img1 + m * HP + (64 * w + 16 * t + 4 * kg) * 2) =
                            make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    }
                }
            }
            lds_barrier();                                     // slab complete
            gemm_unit(accp, img1, ring, wnext, l15, kg);
            lds_barrier();                                     // slab consumed
        }
        {
            const float *bias = p.b_proj + (size_t)l * WD + 64 * w;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const int m = 16 * rb + l15;
                if (m < nrow) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int n = 64 * w + 16 * t + 4 * kg;
                        const float4 bv = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * kg);
                        const float4 rv = ldf4_nt(x_mid + (size_t)m * WD + n);
                        const f32x4_t a = accp.v[rb][t];
                        *reinterpret_cast<float4 *>(x_out + (size_t)m * WD + n) =
                            make_float4(a[0] + bv.x + rv.x, a[1] + bv.y + rv.y, a[2] + bv.z + rv.z, a[3] + bv.w + rv.w);
                    }
                }
            }
        }
        vm_barrier();                                          // x_out complete in memory
    }
}

}  // namespace

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
        return (int)hipFuncSetAttribute((const void *)text_tower_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG);
    }();
    (void)once;
    const int groups = (p.C + p.NP - 1) / p.NP;
    hipLaunchKernelGGL(text_tower_fwd_kernel, dim3(groups), dim3(512), 2 * IMG, ppt_stream(stream), p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
