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

extern "C" int ppt_attention_fwd_quad_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H, float scale,
                                           int causal, hipStream_t s);

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int HD = 64, KVT = 64, QB = 128, TILE = KVT * 128;   // bytes per K or V tile (64 keys x 128 B)

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int key, int dbyte) { return key * 128 + (dbyte ^ (((key >> 1) & 1) << 6)); }

__device__ __forceinline__ float lane_xor32_max(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane_xor32_sum(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_mfma(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ out,
                                                     float *__restrict__ lse, int T, int H, float c /* scale*log2(e) */)
{
    __shared__ __align__(16) unsigned char smem[4 * TILE];    // K0 K1 V0 V1
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, head = bh % H;
    const int64_t rs = 3 * (int64_t)H * HD;
    const bf16_t *qb = qkv + (int64_t)b * T * rs + head * HD;
    const bf16_t *kb = qb + H * HD, *vb = qb + 2 * H * HD;
    const int q0 = blockIdx.x * QB + w * 32;
    const int qrow = q0 + r;

    bf16x8_t qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qrow < T) v = *reinterpret_cast<const uint4 *>(qb + (int64_t)qrow * rs + 16 * kk + 8 * h);
        qf[kk] = __builtin_bit_cast(bf16x8_t, v);
    }

    uint4 sk[2], sv[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = threadIdx.x + 256 * i;
            const int key = kt * KVT + (cidx >> 3), ch = cidx & 7;
            sk[i] = sv[i] = make_uint4(0, 0, 0, 0);
            if (key < T) {
                sk[i] = *reinterpret_cast<const uint4 *>(kb + (int64_t)key * rs + ch * 8);
                sv[i] = *reinterpret_cast<const uint4 *>(vb + (int64_t)key * rs + ch * 8);
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

    const int q_hi = min(T, (int)(blockIdx.x + 1) * QB) - 1;          // last query row of this workgroup
    const int nkt = CAUSAL ? min((T + KVT - 1) / KVT, q_hi / KVT + 1) : (T + KVT - 1) / KVT;

    f32x16_t ot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[i][e] = 0.f;
    float m = -INFINITY, l = 0.f;

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
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const bf16x8_t kf = __builtin_bit_cast(
                        bf16x8_t, *reinterpret_cast<const uint4 *>(Kc + k_off(32 * sub + r, 2 * kk + h)));
                    st[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[sub], 0, 0, 0);
                }
            const bool need_mask = (kt * KVT + KVT > T) || (CAUSAL && kt * KVT + KVT - 1 > q0);
            if (need_mask) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * KVT + 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (key >= T || (CAUSAL && key > qrow)) st[sub][e] = -INFINITY;
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
            bf16x8_t pf[2][2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 u = make_uint4(pack_bf16x2(st[sub][8 * s + 0], st[sub][8 * s + 1]),
                                               pack_bf16x2(st[sub][8 * s + 2], st[sub][8 * s + 3]),
                                               pack_bf16x2(st[sub][8 * s + 4], st[sub][8 * s + 5]),
                                               pack_bf16x2(st[sub][8 * s + 6], st[sub][8 * s + 7]));
                    pf[sub][s] = __builtin_bit_cast(bf16x8_t, u);
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
                        ot[dtile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf), pf[sub][s],
                                                                             ot[dtile], 0, 0, 0);
                    }
        }
        if (kt + 1 < nkt) write_tile(cur ^ 1);
        __syncthreads();
    }

    const float lt = lane_xor32_sum(l);
    if (qrow < T) {
        const float inv = 1.0f / lt;
        bf16_t *ob = out + ((int64_t)b * T + qrow) * (H * HD) + head * HD;
#pragma unroll
        for (int dtile = 0; dtile < 2; ++dtile)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const uint2 u = make_uint2(pack_bf16x2(ot[dtile][4 * gq + 0] * inv, ot[dtile][4 * gq + 1] * inv),
                                           pack_bf16x2(ot[dtile][4 * gq + 2] * inv, ot[dtile][4 * gq + 3] * inv));
                *reinterpret_cast<uint2 *>(ob + 32 * dtile + 8 * gq + 4 * h) = u;
            }
        if (lse && h == 0) lse[(int64_t)bh * T + qrow] = (m + __log2f(lt)) * 0.6931471805599453f;
    }
}

}  // namespace

extern "C" int ppt_attention_fwd_mfma_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H, float scale,
                                           int causal, hipStream_t s)
{
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return ppt_attention_fwd_quad_bf16(qkv, out, lse, Bt, T, H, scale, causal, s);
    dim3 grid((T + QB - 1) / QB, Bt * H);
    const float c = scale * 1.4426950408889634f;
    if (causal)
        hipLaunchKernelGGL(attn_fwd_mfma<true>, grid, dim3(256), 0, s, (const bf16_t *)qkv, (bf16_t *)out, lse, T, H, c);
    else
        hipLaunchKernelGGL(attn_fwd_mfma<false>, grid, dim3(256), 0, s, (const bf16_t *)qkv, (bf16_t *)out, lse, T, H, c);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
