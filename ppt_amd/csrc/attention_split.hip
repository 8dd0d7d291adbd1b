// attention_split.hip -- flash-attention forward for fp32 q / k / v on the 16-bit matrix pipe (precision mode "split16").
//
// The fp32 parity mode runs attention on the vector ALU (attention.hip: attn_fwd_quad<float>, 57 TFLOP/s at the ViT shape:
// 225 us per call, 5.4 of the 13.8 ms of a C2 step once the GEMMs are split16).  This kernel is attention_mfma.hip's streaming
// kernel (point_encoder.py:46-55, T = 513 non-causal; ULIP_models.py:38,49-51, L = 77 causal, prefix-shared) with every operand
// of the two products split into hi + lo IEEE-half pairs (gemm_common.h, "split16": x = half(x) + half(x - half(x)), 22
// significand bits, the lo x lo term dropped):
//     S^T = K . Q^T   = K_lo.Q_hi + K_hi.Q_lo + K_hi.Q_hi          (v_mfma_f32_32x32x16_f16, fp32 accumulation)
//     O^T = V^T . P   = V_lo.P_hi + V_hi.P_lo + V_hi.P_hi
// K and V are split ONCE per workgroup where the register-staged tile goes to LDS (hi and lo images of the 64-key tile, the
// 16-bit kernel's swizzles); Q is split once per wave into registers; P = exp2(S c - m) in [0, 1] is split in registers on its
// way from the S^T accumulator layout to the B operand.  Softmax statistics, the running rescale, the peeled last key of
// T = 64 n + 1 and the output are fp32.  48 MFMAs per 64-key tile and wave instead of 16.
#include "ppt_common.h"
#include <stdlib.h>
#include "attn_rowmap.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

constexpr int HD = 64, KVT = 64, QB = 128, TILE = KVT * 128;   // bytes per hi (or lo) image of a K or V tile
constexpr float P_PRE = 1024.0f;                                // forward: P is split as P x 2^10, O carries the factor to the end

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int key, int dbyte) { return key * 128 + (dbyte ^ (((key >> 1) & 1) << 6)); }

// two fp32 values -> packed halves of their hi parts and of their lo parts
__device__ __forceinline__ void split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const h2_t hh = {ha, hb};
    const h2_t ll = {(_Float16)(a - (float)ha), (_Float16)(b - (float)hb)};
    hi = __builtin_bit_cast(uint32_t, hh);
    lo = __builtin_bit_cast(uint32_t, ll);
}
__device__ __forceinline__ void split8(const float4 &u, const float4 &v, uint4 &hi, uint4 &lo)
{
    split2(u.x, u.y, hi.x, lo.x); split2(u.z, u.w, hi.y, lo.y);
    split2(v.x, v.y, hi.z, lo.z); split2(v.z, v.w, hi.w, lo.w);
}
__device__ __forceinline__ f32x16_t mfma3(uint4 ah, uint4 al, uint4 bh, uint4 bl, f32x16_t c)
{
    c = h16<f16_t>::mfma32(al, bh, c);
    c = h16<f16_t>::mfma32(ah, bl, c);
    return h16<f16_t>::mfma32(ah, bh, c);
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_split16(const float *__restrict__ qkv, float *__restrict__ out,
                                                        float *__restrict__ lse, int Tfull, int H, float c /* scale*log2(e) */,
                                                        int P, int C, int prio, int xcd_map)
{
    __shared__ __align__(16) unsigned char smem[8 * TILE];    // (K_hi K_lo V_hi V_lo) x 2 buffers
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    int bh = blockIdx.y, qblk = blockIdx.x;
    if (xcd_map && (gridDim.y & 7) == 0) {                     // all query blocks of a (batch, head) on one XCD (attention_mfma.hip)
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, slot = lin >> 3;
        bh = (slot / (int)gridDim.x) * 8 + (lin & 7);
        qblk = slot % (int)gridDim.x;
    }
    const int b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD;
    const int T = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);
    const float *qb = qkv + head * HD;
    const float *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const int q0 = qblk * QB + w * 32;
    const int qrow = q0 + r;

    // Q^T fragments: lane (r, h) holds dimensions 16 kk + 8 h .. + 7 of query row r, as hi and lo halves
    uint4 qh[4], ql[4];
    float4 qraw[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qraw[kk][0] = qraw[kk][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qrow < T) {
            const float *qp = qb + am_row(Tfull, P, b, qrow) * rs + 16 * kk + 8 * h;
            qraw[kk][0] = *reinterpret_cast<const float4 *>(qp);
            qraw[kk][1] = *reinterpret_cast<const float4 *>(qp + 4);
        }
        split8(qraw[kk][0], qraw[kk][1], qh[kk], ql[kk]);
    }

    const bool peel = !CAUSAL && T > KVT && (T % KVT) == 1;   // T = 64 n + 1: the last key initialises the running state
    const int Tk = peel ? T - 1 : T;
    float4 sk[4], sv[4];                                       // this thread's 4 + 4 16-byte chunks of the next tile
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int key = kt * KVT + (cidx >> 4), ch = cidx & 15;
            sk[i] = sv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (key < Tk) {
                const int64_t kr = am_row(Tfull, P, b, key) * rs;
                sk[i] = *reinterpret_cast<const float4 *>(kb + kr + ch * 4);
                sv[i] = *reinterpret_cast<const float4 *>(vb + kr + ch * 4);
            }
        }
    };
    auto write_tile = [&](int buf) {                           // the split: 4 floats -> 8 bytes of the hi image + 8 of the lo image
        unsigned char *base = smem + buf * 4 * TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int row = cidx >> 4, ch = cidx & 15;
            uint2 hi, lo;
            split2(sk[i].x, sk[i].y, hi.x, lo.x); split2(sk[i].z, sk[i].w, hi.y, lo.y);
            const int ko = k_off(row, ch >> 1) + (ch & 1) * 8;
            *reinterpret_cast<uint2 *>(base + ko) = hi;
            *reinterpret_cast<uint2 *>(base + TILE + ko) = lo;
            split2(sv[i].x, sv[i].y, hi.x, lo.x); split2(sv[i].z, sv[i].w, hi.y, lo.y);
            const int vo = v_off(row, ch * 8);
            *reinterpret_cast<uint2 *>(base + 2 * TILE + vo) = hi;
            *reinterpret_cast<uint2 *>(base + 3 * TILE + vo) = lo;
        }
    };

    const int q_hi = min(T, (int)(qblk + 1) * QB) - 1;
    const int nkt = CAUSAL ? min((T + KVT - 1) / KVT, q_hi / KVT + 1) : (Tk + KVT - 1) / KVT;

    f32x16_t ot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[i][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    if (peel) {                                                // fp32 throughout: m = c q.k_last, l = 1, O = v_last
        const float *kl = kb + am_row(Tfull, P, b, T - 1) * rs, *vl = vb + am_row(Tfull, P, b, T - 1) * rs;
        float dot = 0.f;                                       // this lane's 32 of the 64 dimensions; the other half-wave has the rest
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 k0 = *reinterpret_cast<const float4 *>(kl + 16 * kk + 8 * h);
            const float4 k1 = *reinterpret_cast<const float4 *>(kl + 16 * kk + 8 * h + 4);
            dot = fmaf(k0.x, qraw[kk][0].x, dot); dot = fmaf(k0.y, qraw[kk][0].y, dot);
            dot = fmaf(k0.z, qraw[kk][0].z, dot); dot = fmaf(k0.w, qraw[kk][0].w, dot);
            dot = fmaf(k1.x, qraw[kk][1].x, dot); dot = fmaf(k1.y, qraw[kk][1].y, dot);
            dot = fmaf(k1.z, qraw[kk][1].z, dot); dot = fmaf(k1.w, qraw[kk][1].w, dot);
        }
        m = xor32_sum(dot) * c;
        l = h == 0 ? 1.0f : 0.0f;                              // (the two half-waves' sums are added at the end)
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 vv = *reinterpret_cast<const float4 *>(vl + 32 * dtile + 8 * gq + 4 * h);
                ot[dtile][4 * gq + 0] = vv.x * P_PRE; ot[dtile][4 * gq + 1] = vv.y * P_PRE;
                ot[dtile][4 * gq + 2] = vv.z * P_PRE; ot[dtile][4 * gq + 3] = vv.w * P_PRE;
            }
    }

    // per-lane constant part of the transposed V reads: lane = 16g + 4q + p supplies row q, columns 4p..4p+3
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_key = 4 * (g >> 1) + tq;                      // + 32*sub + 16*s (+8 for the second read)
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;          // + 64*dt

    load_tile(0);
    write_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const bool active = q0 < T && (!CAUSAL || kt * KVT <= q0 + 31);      // wave-uniform
        if (active) {
            const unsigned char *Kh = smem + cur * 4 * TILE, *Kl = Kh + TILE, *Vh = Kh + 2 * TILE, *Vl = Kh + 3 * TILE;
            f32x16_t st[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) st[i][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                uint4 kh[2], kl2[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int o = k_off(32 * sub + r, 2 * kk + h);
                    kh[sub] = *reinterpret_cast<const uint4 *>(Kh + o);
                    kl2[sub] = *reinterpret_cast<const uint4 *>(Kl + o);
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) st[sub] = mfma3(kh[sub], kl2[sub], qh[kk], ql[kk], st[sub]);
            }
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
            mx = xor32_max(mx);
            const float mn = fmaxf(m, mx * c);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[sub][e], c, -mn));
                    st[sub][e] = pv * P_PRE;                          // (P <= 1: x 2^10 keeps a flat softmax's 1 / T inside hi + lo's 22 bits)
                    psum += pv;
                }
            l = fmaf(l, alpha, psum);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) ot[i][e] *= alpha;
            // P (still in the S^T accumulator layout) -> hi / lo B fragments of the 16-key k-steps, then O^T += V^T . P
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint4 ph, pl;
                    split2(st[sub][8 * s + 0], st[sub][8 * s + 1], ph.x, pl.x);
                    split2(st[sub][8 * s + 2], st[sub][8 * s + 3], ph.y, pl.y);
                    split2(st[sub][8 * s + 4], st[sub][8 * s + 5], ph.z, pl.z);
                    split2(st[sub][8 * s + 6], st[sub][8 * s + 7], ph.w, pl.w);
#pragma unroll
                    for (int dtile = 0; dtile < 2; ++dtile) {
                        const int key0 = 32 * sub + 16 * s + tr_key;
                        const int o0 = v_off(key0, tr_dbyte + 64 * dtile), o1 = v_off(key0 + 8, tr_dbyte + 64 * dtile);
                        struct { s4_t a, b; } vh, vl;
                        vh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(Vh + o0));
                        vh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(Vh + o1));
                        vl.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(Vl + o0));
                        vl.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(Vl + o1));
                        ot[dtile] = mfma3(__builtin_bit_cast(uint4, vh), __builtin_bit_cast(uint4, vl), ph, pl, ot[dtile]);
                    }
                }
        }
        if (kt + 1 < nkt) write_tile(cur ^ 1);
        __syncthreads();
    }

    const float lt = xor32_sum(l);
    if (qrow < T && qrow >= q_lo) {
        const float inv = 1.0f / (lt * P_PRE);
        float *ob = out + am_row(Tfull, P, b, qrow) * (H * HD) + head * HD;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4 *>(ob + 32 * dtile + 8 * gq + 4 * h) =
                    make_float4(ot[dtile][4 * gq + 0] * inv, ot[dtile][4 * gq + 1] * inv, ot[dtile][4 * gq + 2] * inv, ot[dtile][4 * gq + 3] * inv);
        if (lse && h == 0) lse[am_stat(Tfull, P, H, b, head, qrow)] = (m + __log2f(lt)) * 0.6931471805599453f;
    }
}

// =================================================================================================
// Backward, plain layout (P == 0; the un-frozen ViT block at T = 513 is where it matters: the fp32 VALU kernels take 2.7 + 1.4 ms
// per C3 step, a fifth of the split16 step).  attention_mfma.hip's two-kernel form -- dK / dV per 128-key block, dQ per 128-query
// block, delta = rowsum(dO . O) from a small kernel in front -- with every MFMA operand a hi + lo half pair:
//     S = Q K^T, dP = dO V^T                      (recomputed; three MFMAs per product)
//     P = exp2(S c - lse), dS = P (dP - delta) scale
//     dV^T += dO^T P,  dK^T += Q^T dS,  dQ^T += K^T dS^T
// Gradients are small numbers of any size: every ROW of dO is multiplied by the power of two that puts its largest element in
// [2^11, 2^12) before its split (row_pow2), P and dS tiles likewise per wave (tile_pow2 below) -- block floating point -- and the
// factors are divided out again exactly: results as accurate as the fp32 kernels' whatever the gradient's magnitude.
// =================================================================================================
__device__ __forceinline__ float pow2_for(float m, int target)        // power of two f with m * f in [2^target, 2^(target+1)); 1 for m = 0
{
    const int be = (int)((__float_as_uint(m) >> 23) & 0xffu);
    const int k = be == 0 ? 0 : max(-120, min(127 + target - be, 120));
    return __uint_as_float((uint32_t)(127 + k) << 23);
}

__global__ __launch_bounds__(256) void attn_delta_f32(const float *__restrict__ out, const float *__restrict__ dout, float *__restrict__ delta,
                                                      int T, int H, int P, int64_t rows /* physical rows x H */)
{
    // delta = sum_d out[row, head, d] * dout[row, head, d]: one 16-lane group per (physical row, head); stored where am_stat puts
    // it -- [(b * H + head) * T + pos] in the plain layout, [row * H + head] in the prefix-shared one
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (i >= rows) return;
    const int64_t row = i / H;
    const int head = (int)(i % H);
    const float4 a = *reinterpret_cast<const float4 *>(out + (row * H + head) * HD + 4 * l);
    const float4 g = *reinterpret_cast<const float4 *>(dout + (row * H + head) * HD + 4 * l);
    float d = a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
    d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4); d += __shfl_xor(d, 8);
    if (l == 0) delta[P > 0 ? i : ((row / T) * H + head) * T + row % T] = d;
}

// the fp32 partial dK / dV of the shared prefix rows, one slot per virtual sequence, folded in sequence order (attn_rowmap.h;
// attention.hip: attn_prefix_reduce)
__global__ __launch_bounds__(256) void attn_prefix_fold_f32(const float *__restrict__ part, int nseq, int P, int HHD, float *__restrict__ dqkv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P * 2 * HHD) return;
    float acc = 0.f;
    const float *src = part + i;
    const int64_t stride = (int64_t)P * 2 * HHD;
    int v = 0;
    for (; v + 8 <= nseq; v += 8) {                      // eight loads in flight, added in sequence order (41 dependent loads: 11.3 us)
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(v + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; v < nseq; ++v) acc += src[v * stride];
    const int pos = i / (2 * HHD), rem = i - pos * 2 * HHD;
    dqkv[(int64_t)pos * 3 * HHD + HHD + rem] = acc;
}

constexpr int QT = 32, QTILE = QT * 128;          // a 32-row hi (or lo) image

// P = exp2(.) and dS = P (dP - delta) scale span many binades (1 / T for a flat softmax, ~1 for a peaked one; gradients of any size),
// half's hi + lo pair keeps its 22 bits over ~14 of them.  So each wave takes the largest magnitude of its 32 x 32 tile and
// multiplies the tile by the power of two that puts it in [2^13, 2^14) before the split (block floating point); the running
// accumulator carries the scale it was last fed at and is re-scaled -- exactly, powers of two -- only when the next tile's differs.
__device__ __forceinline__ float tile_pow2(const f32x16_t &x)
{
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(x[e]));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    return pow2_for(m, 13);
}
// acc holds (true sum) x run; the next tile arrives multiplied by f
__device__ __forceinline__ void rescale(f32x16_t (&acc)[2], float &run, float f)
{
    if (f != run) {                                                    // wave-uniform
        const float t = f / run;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] *= t;
        run = f;
    }
}

// sa (rows of the accumulator layout = the A operand's rows, column = lane & 31) -> hi / lo B-operand fragments of 16-row step s
__device__ __forceinline__ void split_acc(const f32x16_t &x, int s, uint4 &hi, uint4 &lo)
{
    split2(x[8 * s + 0], x[8 * s + 1], hi.x, lo.x); split2(x[8 * s + 2], x[8 * s + 3], hi.y, lo.y);
    split2(x[8 * s + 4], x[8 * s + 5], hi.z, lo.z); split2(x[8 * s + 6], x[8 * s + 7], hi.w, lo.w);
}
__device__ __forceinline__ uint4 tr_frag(const unsigned char *img, int row0, int dbyte)
{
    struct { s4_t a, b; } f;
    f.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + v_off(row0, dbyte)));
    f.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + v_off(row0 + 8, dbyte)));
    return __builtin_bit_cast(uint4, f);
}

// dK, dV of the 128 keys of block blockIdx.x (wave = 32 keys), walking the queries in 32-row tiles
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_split16(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                            const float *__restrict__ lse, const float *__restrict__ delta,
                                                            float *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C,
                                                            float *__restrict__ part)
{
    // images of the query tile: Q row / Q tr / dO' row / dO' tr, each hi + lo (8 x 4 KiB), + lse2[32] + delta'[32]
    __shared__ __align__(16) unsigned char smem[8 * QTILE + 384];     // ... + 1 / (row scale of dO)[32]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD, os = (int64_t)H * HD;
    const int T = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);      // (attn_rowmap.h: b is a virtual sequence when P > 0)
    const float *qb = qkv + head * HD;
    const float *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const float *gb = dout + head * HD;
    const int k0 = blockIdx.x * 128 + w * 32;
    const int key = k0 + r;
    const float c = scale * 1.4426950408889634f;

    uint4 kh[4], kl[4], vh[4], vl[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, v0 = a0, v1 = a0;
        if (key < T) {
            const int64_t kr = am_row(Tfull, P, b, key) * rs;
            const float *kp = kb + kr + 16 * kk + 8 * h, *vp = vb + kr + 16 * kk + 8 * h;
            a0 = *reinterpret_cast<const float4 *>(kp); a1 = *reinterpret_cast<const float4 *>(kp + 4);
            v0 = *reinterpret_cast<const float4 *>(vp); v1 = *reinterpret_cast<const float4 *>(vp + 4);
        }
        split8(a0, a1, kh[kk], kl[kk]);
        split8(v0, v1, vh[kk], vl[kk]);
    }
    f32x16_t dvt[2], dkt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dvt[i][e] = 0.f; dkt[i][e] = 0.f; }

    float run_v = 1.0f, run_k = 1.0f;                                  // the scales dvt / dkt currently carry
    const int nqt = (T + QT - 1) / QT;
    // queries before the block's first key see none of it; queries below q_lo belong to the prefix sequence, not to this one
    const int qt0 = max(CAUSAL ? (int)(blockIdx.x * 128) / QT : 0, q_lo / QT);
    float4 sq[2], sg[2];
    float sl[2], sd[2];                                               // (used by the thread that holds chunk 0 of the row)
    auto load_tile = [&](int qt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int q = qt * QT + (cidx >> 4), ch = cidx & 15;
            sq[i] = sg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            sl[i] = INFINITY; sd[i] = 0.f;                            // +inf -> p = 0 for padded rows and rows this sequence does not own
            if (q < T) {
                const int64_t qr = am_row(Tfull, P, b, q);
                sq[i] = *reinterpret_cast<const float4 *>(qb + qr * rs + ch * 4);
                sg[i] = *reinterpret_cast<const float4 *>(gb + qr * os + ch * 4);
                if (ch == 0 && q >= q_lo) {
                    sl[i] = lse[am_stat(Tfull, P, H, b, head, q)] * 1.4426950408889634f;
                    sd[i] = delta[am_stat(Tfull, P, H, b, head, q)];
                }
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int row = cidx >> 4, ch = cidx & 15;
            uint2 hi, lo;
            split2(sq[i].x, sq[i].y, hi.x, lo.x); split2(sq[i].z, sq[i].w, hi.y, lo.y);
            const int ko = k_off(row, ch >> 1) + (ch & 1) * 8, vo = v_off(row, ch * 8);
            *reinterpret_cast<uint2 *>(smem + 0 * QTILE + ko) = hi; *reinterpret_cast<uint2 *>(smem + 1 * QTILE + ko) = lo;
            *reinterpret_cast<uint2 *>(smem + 2 * QTILE + vo) = hi; *reinterpret_cast<uint2 *>(smem + 3 * QTILE + vo) = lo;
            // the row's 16 chunks sit in 16 consecutive lanes: its largest |dO| -> the row's power of two
            float m = fmaxf(fmaxf(fabsf(sg[i].x), fabsf(sg[i].y)), fmaxf(fabsf(sg[i].z), fabsf(sg[i].w)));
            m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
            const float gs = pow2_for(m, 11);
            split2(sg[i].x * gs, sg[i].y * gs, hi.x, lo.x); split2(sg[i].z * gs, sg[i].w * gs, hi.y, lo.y);
            *reinterpret_cast<uint2 *>(smem + 4 * QTILE + ko) = hi; *reinterpret_cast<uint2 *>(smem + 5 * QTILE + ko) = lo;
            *reinterpret_cast<uint2 *>(smem + 6 * QTILE + vo) = hi; *reinterpret_cast<uint2 *>(smem + 7 * QTILE + vo) = lo;
            if (ch == 0) {
                reinterpret_cast<float *>(smem + 8 * QTILE)[row] = sl[i];
                reinterpret_cast<float *>(smem + 8 * QTILE + 128)[row] = sd[i] * gs;          // delta in the row's scale
                reinterpret_cast<float *>(smem + 8 * QTILE + 256)[row] = 1.0f / gs;
            }
        }
    };
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * (g >> 1) + tq;
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;
    const float *L2 = reinterpret_cast<const float *>(smem + 8 * QTILE);
    const float *DL = reinterpret_cast<const float *>(smem + 8 * QTILE + 128);
    const float *IS = reinterpret_cast<const float *>(smem + 8 * QTILE + 256);

    if (qt0 < nqt) { load_tile(qt0); write_tile(); }
    __syncthreads();
    for (int qt = qt0; qt < nqt; ++qt) {
        if (qt + 1 < nqt) load_tile(qt + 1);                          // in registers while this tile is multiplied
        const bool active = k0 < T && (!CAUSAL || qt * QT + QT - 1 >= k0);       // wave-uniform
        if (active) {
            f32x16_t sa, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int o = k_off(r, 2 * kk + h);
                const uint4 qa_h = *reinterpret_cast<const uint4 *>(smem + 0 * QTILE + o), qa_l = *reinterpret_cast<const uint4 *>(smem + 1 * QTILE + o);
                const uint4 ga_h = *reinterpret_cast<const uint4 *>(smem + 4 * QTILE + o), ga_l = *reinterpret_cast<const uint4 *>(smem + 5 * QTILE + o);
                sa = mfma3(qa_h, qa_l, kh[kk], kl[kk], sa);           // S    [q][key]
                dp = mfma3(ga_h, ga_l, vh[kk], vl[kk], dp);           // dP'  [q][key]  (x the row's scale)
            }
            const bool diag = CAUSAL && qt * QT < k0 + 32;           // some (q, key) pairs of this tile are masked
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 l4 = *reinterpret_cast<const float4 *>(L2 + 8 * gq + 4 * h);
                const float4 d4 = *reinterpret_cast<const float4 *>(DL + 8 * gq + 4 * h);
                const float4 i4 = *reinterpret_cast<const float4 *>(IS + 8 * gq + 4 * h);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w}, iv[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * gq + j;
                    float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], c, -lv[j]));
                    if (diag && key > qt * QT + 8 * gq + 4 * h + j) pv = 0.f;
                    dp[e] = pv * (dp[e] - dv[j]) * (scale * iv[j]);   // dS: the row's scale divided out again
                    sa[e] = pv * iv[j];                               // P / (row scale): pairs with the scaled dO row in dV
                }
            }
            const float fp = tile_pow2(sa), fd = tile_pow2(dp);
            rescale(dvt, run_v, fp);
            rescale(dkt, run_k, fd);
#pragma unroll
            for (int e = 0; e < 16; ++e) { sa[e] *= fp; dp[e] *= fd; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 ph, pl, dh, dl;
                split_acc(sa, s, ph, pl);
                split_acc(dp, s, dh, dl);
#pragma unroll
                for (int dtile = 0; dtile < 2; ++dtile) {
                    const uint4 gt_h = tr_frag(smem + 6 * QTILE, 16 * s + tr_row, tr_dbyte + 64 * dtile);   // dO'^T
                    const uint4 gt_l = tr_frag(smem + 7 * QTILE, 16 * s + tr_row, tr_dbyte + 64 * dtile);
                    dvt[dtile] = mfma3(gt_h, gt_l, ph, pl, dvt[dtile]);
                    const uint4 qt_h = tr_frag(smem + 2 * QTILE, 16 * s + tr_row, tr_dbyte + 64 * dtile);   // Q^T
                    const uint4 qt_l = tr_frag(smem + 3 * QTILE, 16 * s + tr_row, tr_dbyte + 64 * dtile);
                    dkt[dtile] = mfma3(qt_h, qt_l, dh, dl, dkt[dtile]);
                }
            }
        }
        __syncthreads();                                              // everybody has left this tile's images
        if (qt + 1 < nqt) write_tile();
        __syncthreads();
    }
    if (key < T) {
        // a SHARED key (P > 0, key < P): this virtual sequence's contribution goes to its fp32 partial slot
        // [b][key][K | V][H * HD]; the slots are folded in a fixed order by attn_prefix_fold_f32 (no atomics)
        const bool shared = P > 0 && key < P;
        float *ok = shared ? part + (((int64_t)b * P + key) * 2) * os + head * HD
                           : dqkv + am_row(Tfull, P, b, key) * rs + head * HD + H * HD;
        float *ov = ok + (shared ? os : (int64_t)H * HD);
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d = 32 * dtile + 8 * gq + 4 * h;
                const float pk = 1.0f / run_k, pv = 1.0f / run_v;
                *reinterpret_cast<float4 *>(ok + d) = make_float4(dkt[dtile][4 * gq] * pk, dkt[dtile][4 * gq + 1] * pk,
                                                                  dkt[dtile][4 * gq + 2] * pk, dkt[dtile][4 * gq + 3] * pk);
                *reinterpret_cast<float4 *>(ov + d) = make_float4(dvt[dtile][4 * gq] * pv, dvt[dtile][4 * gq + 1] * pv,
                                                                  dvt[dtile][4 * gq + 2] * pv, dvt[dtile][4 * gq + 3] * pv);
            }
    }
}

// dQ of the 128 queries of block blockIdx.x (wave = 32 queries), walking the keys in 64-row tiles
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_split16(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                           const float *__restrict__ lse, const float *__restrict__ delta,
                                                           float *__restrict__ dqkv, int Tfull, int H, float scale, int P, int C)
{
    __shared__ __align__(16) unsigned char smem[6 * TILE];            // K row / K tr / V row images, hi + lo (6 x 8 KiB)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD, os = (int64_t)H * HD;
    const int T = am_len(Tfull, P, C, b), q_lo = am_qlo(P, C, b);
    const float *qb = qkv + head * HD;
    const float *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const float *gb = dout + head * HD;
    const int q0 = blockIdx.x * QB + w * 32;
    const int qrow = q0 + r;
    const float c = scale * 1.4426950408889634f;

    uint4 qh[4], ql[4], gh[4], gl[4];
    float4 graw[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, g0 = a0, g1 = a0;
        if (qrow < T) {
            const int64_t qr = am_row(Tfull, P, b, qrow);
            const float *qp = qb + qr * rs + 16 * kk + 8 * h, *gp = gb + qr * os + 16 * kk + 8 * h;
            a0 = *reinterpret_cast<const float4 *>(qp); a1 = *reinterpret_cast<const float4 *>(qp + 4);
            g0 = *reinterpret_cast<const float4 *>(gp); g1 = *reinterpret_cast<const float4 *>(gp + 4);
        }
        split8(a0, a1, qh[kk], ql[kk]);
        graw[kk][0] = g0; graw[kk][1] = g1;
    }
    // this query row's power of two (its 64 dO values: 32 here, 32 in lane ^ 32)
    float gm = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            gm = fmaxf(gm, fmaxf(fmaxf(fabsf(graw[kk][i].x), fabsf(graw[kk][i].y)), fmaxf(fabsf(graw[kk][i].z), fabsf(graw[kk][i].w))));
    const float gs = pow2_for(xor32_max(gm), 11);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float4 g0 = graw[kk][0], g1 = graw[kk][1];
        g0.x *= gs; g0.y *= gs; g0.z *= gs; g0.w *= gs; g1.x *= gs; g1.y *= gs; g1.z *= gs; g1.w *= gs;
        split8(g0, g1, gh[kk], gl[kk]);
    }
    const bool own = qrow < T && qrow >= q_lo;
    const float l2 = own ? lse[am_stat(Tfull, P, H, b, head, qrow)] * 1.4426950408889634f : INFINITY;
    const float dl = own ? delta[am_stat(Tfull, P, H, b, head, qrow)] * gs : 0.f;

    float4 sk[4], sv[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int kx = kt * KVT + (cidx >> 4), ch = cidx & 15;
            sk[i] = sv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kx < T) {
                const int64_t kr = am_row(Tfull, P, b, kx) * rs;
                sk[i] = *reinterpret_cast<const float4 *>(kb + kr + ch * 4);
                sv[i] = *reinterpret_cast<const float4 *>(vb + kr + ch * 4);
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int row = cidx >> 4, ch = cidx & 15;
            uint2 hi, lo;
            split2(sk[i].x, sk[i].y, hi.x, lo.x); split2(sk[i].z, sk[i].w, hi.y, lo.y);
            const int ko = k_off(row, ch >> 1) + (ch & 1) * 8, vo = v_off(row, ch * 8);
            *reinterpret_cast<uint2 *>(smem + 0 * TILE + ko) = hi; *reinterpret_cast<uint2 *>(smem + 1 * TILE + ko) = lo;
            *reinterpret_cast<uint2 *>(smem + 2 * TILE + vo) = hi; *reinterpret_cast<uint2 *>(smem + 3 * TILE + vo) = lo;
            split2(sv[i].x, sv[i].y, hi.x, lo.x); split2(sv[i].z, sv[i].w, hi.y, lo.y);
            *reinterpret_cast<uint2 *>(smem + 4 * TILE + ko) = hi; *reinterpret_cast<uint2 *>(smem + 5 * TILE + ko) = lo;
        }
    };
    const int q_hi = min(T, (int)(blockIdx.x + 1) * QB) - 1;
    const int nkt = CAUSAL ? min((T + KVT - 1) / KVT, q_hi / KVT + 1) : (T + KVT - 1) / KVT;
    f32x16_t dqt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqt[i][e] = 0.f;
    float run_q = 1.0f;                                               // the scale dqt currently carries
    const float gsi = 1.0f / gs;
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * (g >> 1) + tq;
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;

    load_tile(0);
    write_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);
        const bool active = q0 < T && (!CAUSAL || kt * KVT <= q0 + 31);
        if (active) {
            const bool need_mask = (kt * KVT + KVT > T) || (CAUSAL && kt * KVT + KVT - 1 > q0);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16_t sa, dp;
#pragma unroll
                for (int e = 0; e < 16; ++e) { sa[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int o = k_off(32 * sub + r, 2 * kk + h);
                    const uint4 ka_h = *reinterpret_cast<const uint4 *>(smem + 0 * TILE + o), ka_l = *reinterpret_cast<const uint4 *>(smem + 1 * TILE + o);
                    const uint4 va_h = *reinterpret_cast<const uint4 *>(smem + 4 * TILE + o), va_l = *reinterpret_cast<const uint4 *>(smem + 5 * TILE + o);
                    sa = mfma3(ka_h, ka_l, qh[kk], ql[kk], sa);       // S^T   [key][q]
                    dp = mfma3(va_h, va_l, gh[kk], gl[kk], dp);       // dP'^T [key][q]  (x this query's scale)
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float pv = __builtin_amdgcn_exp2f(fmaf(sa[e], c, -l2));
                    if (need_mask) {
                        const int kx = kt * KVT + 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (kx >= T || (CAUSAL && kx > qrow)) pv = 0.f;
                    }
                    dp[e] = pv * (dp[e] - dl) * (scale * gsi);        // dS^T: the query's scale divided out again
                }
                const float fd = tile_pow2(dp);
                rescale(dqt, run_q, fd);
#pragma unroll
                for (int e = 0; e < 16; ++e) dp[e] *= fd;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint4 dh, dlo;
                    split_acc(dp, s, dh, dlo);
#pragma unroll
                    for (int dtile = 0; dtile < 2; ++dtile) {
                        const uint4 kt_h = tr_frag(smem + 2 * TILE, 32 * sub + 16 * s + tr_row, tr_dbyte + 64 * dtile);   // K^T
                        const uint4 kt_l = tr_frag(smem + 3 * TILE, 32 * sub + 16 * s + tr_row, tr_dbyte + 64 * dtile);
                        dqt[dtile] = mfma3(kt_h, kt_l, dh, dlo, dqt[dtile]);
                    }
                }
            }
        }
        __syncthreads();
        if (kt + 1 < nkt) write_tile();
        __syncthreads();
    }
    if (own) {
        float *oq = dqkv + am_row(Tfull, P, b, qrow) * rs + head * HD;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4 *>(oq + 32 * dtile + 8 * gq + 4 * h) =
                    make_float4(dqt[dtile][4 * gq] * (1.0f / run_q), dqt[dtile][4 * gq + 1] * (1.0f / run_q),
                                dqt[dtile][4 * gq + 2] * (1.0f / run_q), dqt[dtile][4 * gq + 3] * (1.0f / run_q));
    }
}

}  // namespace

// fp32 qkv [rows, 3, H, 64] -> out [rows, H, 64] fp32, lse fp32 (the layouts of ppt_attention_fwd / ppt_attention_prefix_fwd with
// dtype PPT_F32; P > 0: the prefix-shared causal layout of attn_rowmap.h with C = Bt prompts).
extern "C" int ppt_attention_fwd_split16(const void *qkv, void *out, float *lse, int Bt, int T, int P, int H, int hd, float scale,
                                         int causal, void *stream)
{
    if (!qkv || !out || Bt <= 0 || T <= 0 || H <= 0 || hd != HD || P < 0 || P >= T) return PPT_EINVAL;
    if (P > 0 && !causal) return PPT_EINVAL;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return PPT_EINVAL;
    hipStream_t s = ppt_stream(stream);
    const float c = scale * 1.4426950408889634f;
    const int prio = ppt_get_wave_priority();
    dim3 grid((T + QB - 1) / QB, (Bt + (P > 0)) * H);
    if (grid.y > 65535) return PPT_EUNSUPPORTED;
    static const int xcd_map = getenv("PPT_ATTN_XCD_MAP") == nullptr || atoi(getenv("PPT_ATTN_XCD_MAP")) != 0;
    if (causal) hipLaunchKernelGGL((attn_fwd_split16<true>), grid, dim3(256), 0, s, (const float *)qkv, (float *)out, lse, T, H, c, P, Bt, prio, xcd_map);
    else hipLaunchKernelGGL((attn_fwd_split16<false>), grid, dim3(256), 0, s, (const float *)qkv, (float *)out, lse, T, H, c, P, Bt, prio, xcd_map);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}


// Backward: qkv / dqkv [rows, 3, H, 64], out / dout [rows, H, 64], lse and the scratch `delta` in the layouts of ppt_attention_bwd
// (P == 0: rows = Bt * T, statistics [Bt, H, T]) or ppt_attention_prefix_bwd (P > 0, causal: rows = P + Bt (T - P), statistics
// [rows, H], `part` = the fp32 workspace of ppt_attention_prefix_workspace_bytes) -- all fp32, the results of those entry points
// with dtype PPT_F32, every product on the 16-bit matrix pipe from hi + lo half pairs.
extern "C" int ppt_attention_bwd_split16(const void *qkv, const void *out, const void *dout, const float *lse, float *delta, void *dqkv,
                                         float *part, int Bt, int T, int P, int H, int hd, float scale, int causal, void *stream)
{
    if (!qkv || !out || !dout || !lse || !delta || !dqkv || Bt <= 0 || T <= 0 || H <= 0 || hd != HD || P < 0 || P >= T) return PPT_EINVAL;
    if (P > 0 && (!part || !causal)) return PPT_EINVAL;
    if ((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv | (uintptr_t)part) & 15) != 0) return PPT_EINVAL;
    hipStream_t s = ppt_stream(stream);
    const int64_t rows = (P > 0 ? (int64_t)P + (int64_t)Bt * (T - P) : (int64_t)Bt * T) * H;
    hipLaunchKernelGGL(attn_delta_f32, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, (const float *)out, (const float *)dout, delta, T, H, P, rows);
    PPT_CHECK_LAUNCH();
    dim3 grid((T + 127) / 128, (Bt + (P > 0)) * H);
    if (grid.y > 65535) return PPT_EUNSUPPORTED;
    if (causal) {
        hipLaunchKernelGGL((attn_bwd_dq_split16<true>), grid, dim3(256), 0, s, (const float *)qkv, (const float *)dout, lse, delta, (float *)dqkv, T, H, scale, P, Bt);
        hipLaunchKernelGGL((attn_bwd_dkv_split16<true>), grid, dim3(256), 0, s, (const float *)qkv, (const float *)dout, lse, delta, (float *)dqkv, T, H, scale, P, Bt, part);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_split16<false>), grid, dim3(256), 0, s, (const float *)qkv, (const float *)dout, lse, delta, (float *)dqkv, T, H, scale, P, Bt);
        hipLaunchKernelGGL((attn_bwd_dkv_split16<false>), grid, dim3(256), 0, s, (const float *)qkv, (const float *)dout, lse, delta, (float *)dqkv, T, H, scale, P, Bt, part);
    }
    PPT_CHECK_LAUNCH();
    if (P > 0) {
        const int n = P * 2 * H * HD;
        hipLaunchKernelGGL(attn_prefix_fold_f32, dim3((n + 255) / 256), dim3(256), 0, s, part, Bt + 1, P, H * HD, (float *)dqkv);
        PPT_CHECK_LAUNCH();
    }
    return PPT_OK;
}
