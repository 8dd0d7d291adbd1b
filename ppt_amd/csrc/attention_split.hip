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
                ot[dtile][4 * gq + 0] = vv.x; ot[dtile][4 * gq + 1] = vv.y;
                ot[dtile][4 * gq + 2] = vv.z; ot[dtile][4 * gq + 3] = vv.w;
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
                    st[sub][e] = pv;
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
        const float inv = 1.0f / lt;
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
