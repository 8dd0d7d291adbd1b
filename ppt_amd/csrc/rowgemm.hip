// rowgemm.hip -- C[M,N] = epilogue( prologue(A)[M,K] . W[N,K]^T ) for the SHORT-K linears of the two transformers
// (K = 384: PointBERT qkv / proj / fc1, point_encoder.py:46-58,76-79; K = 512: the CLIP text tower's in_proj / out_proj /
// c_fc, ULIP_models.py:35-56), with the weight STATIONARY in registers and the LayerNorm in front of the linear applied
// while the rows are staged.
//
// Why not the tile loop of gemm.hip: with K = 384 a 128 x 128 tile is six 64-wide slabs -- every tile is a pipeline fill,
// and what bounds the loop is what a CU pulls out of L2 into LDS (both operands, 64 flop per byte at best), not the
// matrix pipe: qkv 500, fc1 420, proj 350 TFLOP/s.  Here the B operand moves ONCE per workgroup: a workgroup owns NB output
// columns (384 resp. 256), each of its 8 waves keeps NW = NB / 8 of them -- W[n0 .. n0 + NW) x K, 144 resp. 128 VGPRs --
// for the whole kernel, and walks 32-row tiles of A.  The only operand that moves is A: 32 rows go to LDS once
// (double-buffered, one barrier per tile, the next tile's global loads in flight during the MFMAs) and every wave
// reads its fragments from there.  LDS row pitch 2K + 32 bytes: the four 16-lane groups of a ds_read_b128 of the
// 16x16x32 operand then hit 64 distinct banks.
//
// The product is formed TRANSPOSED, D[n][m] = sum_k W[n][k] A[m][k] (v_mfma_f32_16x16x32_bf16 with W as the first
// operand): a lane then holds four CONSECUTIVE output columns of one row -- an 8-byte bf16 or 16-byte fp32 store, and a
// 16-byte read of the fp32 residual stream -- with no transpose through LDS.
//
// A prologue (LN = true): A is the fp32 residual stream x [M,K]; 16 lanes share a row, each holding K/16 values; mean and
// variance are the two-pass forms of nn.LayerNorm reduced over the 16 lanes with four DPP adds (every lane ends with the
// same bits), and (x - mean) * rstd * gamma + beta goes to LDS as bf16 -- the LayerNorm output never exists in HBM.
// Column groups recompute it (3x for qkv, 4x for fc1): 1.5 KB per row out of L2 against 23 ... 31 GFLOP per launch.
//
// Epilogues: bias; GELU (erf, A&S 7.1.26 as gemm.hip) / QuickGELU with an optional copy of the pre-activation; or the
// residual form out = residual + row_scale[m / rows] * (acc + bias) (+ residual2), fp32, in place allowed.
#include "ppt_common.h"
#include "ppt_act.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int K> struct RG {
    static constexpr int NCB = (K == 384) ? 3 : 2;      // 16-column blocks per wave
    static constexpr int NW = 16 * NCB;                  // columns per wave
    static constexpr int NB = 8 * NW;                    // columns per workgroup
    static constexpr int KS = K / 32;                    // k-steps of the 16x16x32 MFMA
    static constexpr int PITCH = 2 * K + 32;             // bytes per LDS row
    static constexpr int BUF = 32 * PITCH;
    static constexpr int F4 = K / 64;                    // float4 per thread and row, LayerNorm prologue (16 threads per row)
    static constexpr int U4 = K / 128;                   // uint4 per thread and row, bf16 A
    static constexpr int CPITCH = 2 * NB + 16;           // bytes per row of the bf16 C tile image
    static constexpr int CBUF = 32 * CPITCH;
    static constexpr int LDS = 2 * BUF + 2 * K * 4 + NB * 4 + 2 * CBUF;   // two A buffers, gamma, beta, the bias slice, two C tiles
};

// sum over the 16 lanes of a DPP row; every lane gets the same bits (each step adds a value to its mirror image)
__device__ __forceinline__ float row16_sum(float v)
{
    v += __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(v)));       // quad_perm [1,0,3,2]
    v += __uint_as_float(dpp_mov<0x4E, 0xf>(__float_as_uint(v)));       // quad_perm [2,3,0,1]
    v += __uint_as_float(dpp_mov<0x141, 0xf>(__float_as_uint(v)));      // row_half_mirror
    v += __uint_as_float(dpp_mov<0x140, 0xf>(__float_as_uint(v)));      // row_mirror
    return v;
}

enum { EPI_BF16 = 0, EPI_RESIDUAL = 1 };
// Diagnostic build only (tools/rowgemm_stamp.py compiles this file with -DPPT_RG_STAMP into its own library): every wave
// writes s_memtime stamps of its phase boundaries to the buffer passed in `residual2` (unused by the bf16 epilogue) --
// [workgroup][wave][iteration < 8][8 stamps].  No stamp exists in the shipped kernel.
#ifdef PPT_RG_STAMP
#define RG_STAMP(slot) do { if (lane == 0 && it < 8) stamps[((size_t)(blockIdx.x * 8 + w) * 8 + it) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_STAMP(slot) do { } while (0)
#endif
#ifndef PPT_ROWGEMM_STAGGER
#define PPT_ROWGEMM_STAGGER 1
#endif
constexpr bool STAGGER = PPT_ROWGEMM_STAGGER != 0;

template <typename F, int K, bool LN, int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void rowgemm_kernel(const ppt_rowgemm_params p)
{
    using G = RG<K>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one): the column groups of one walker -- which
    // read the same A rows at about the same time -- get linear ids that agree modulo 8, so two of three (three of four)
    // of those reads are hits in the XCD's L2 instead of trips to the Infinity Cache.
    const int b_lo = blockIdx.x & 7, b_q = blockIdx.x >> 3;
    const int group = b_q % p.groups, walker = (b_q / p.groups) * 8 + b_lo;
    const int n0 = group * G::NB + w * G::NW;                            // first column of this wave
    const int n_tiles = (p.M + 31) >> 5;
    const int n_walk = p.walkers;
#ifdef PPT_RG_STAMP
    if (lane == 0) ((unsigned long long *)p.residual2)[((size_t)(blockIdx.x * 8 + w) * 8 + 7) * 8 + 7] = __builtin_amdgcn_s_memtime();   // kernel entry
#endif
    int t = walker;
    if (t >= n_tiles) return;
    float *gam = reinterpret_cast<float *>(smem + 2 * G::BUF), *bet = gam + K, *bia = bet + K;
    unsigned char *cbuf = reinterpret_cast<unsigned char *>(bia + G::NB);       // two bf16 C tiles (bf16 epilogue only)

    // ---- loader: thread -> row r (0..31) of the tile, 16 threads per row
    const int r = threadIdx.x >> 4, j = threadIdx.x & 15;
    float4 xf[LN ? G::F4 : 1];
    uint4 xb0 = {}, xb1 = {}, xb2 = {}, xb3 = {};     // (named values: an array that lives across the barrier is kept in scratch)
    auto load = [&](int tile) {
        const size_t row = (size_t)min(tile * 32 + r, p.M - 1);
        if constexpr (LN) {
            const float *src = (const float *)p.A + row * K;
#pragma unroll
            for (int i = 0; i < G::F4; ++i) xf[i] = *reinterpret_cast<const float4 *>(src + 4 * (j + 16 * i));
        } else {
            const bf16_t *src = (const bf16_t *)p.A + row * K + 8 * j;
            xb0 = *reinterpret_cast<const uint4 *>(src);
            xb1 = *reinterpret_cast<const uint4 *>(src + 128);
            xb2 = *reinterpret_cast<const uint4 *>(src + 256);
            if constexpr (G::U4 > 3) xb3 = *reinterpret_cast<const uint4 *>(src + 384);
        }
    };
    auto stage = [&](int buf, int tile) {
        unsigned char *dst = smem + buf * G::BUF + r * G::PITCH;
        if constexpr (LN) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < G::F4; ++i) s += (xf[i].x + xf[i].y) + (xf[i].z + xf[i].w);
            const float mean = row16_sum(s) * (1.0f / (float)K);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < G::F4; ++i) {
                const float d0 = xf[i].x - mean, d1 = xf[i].y - mean, d2 = xf[i].z - mean, d3 = xf[i].w - mean;
                q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
            }
            const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / (float)K) + p.ln_eps);
            if (p.ln_mean && group == 0 && j == 0 && tile * 32 + r < p.M) {     // the statistics the LayerNorm backward needs
                p.ln_mean[tile * 32 + r] = mean;
                p.ln_rstd[tile * 32 + r] = rstd;
            }
#pragma unroll
            for (int i = 0; i < G::F4; ++i) {
                const int c = 4 * (j + 16 * i);
                const float4 g = *reinterpret_cast<const float4 *>(gam + c), b = *reinterpret_cast<const float4 *>(bet + c);
                const float o0 = (xf[i].x - mean) * rstd * g.x + b.x, o1 = (xf[i].y - mean) * rstd * g.y + b.y;
                const float o2 = (xf[i].z - mean) * rstd * g.z + b.z, o3 = (xf[i].w - mean) * rstd * g.w + b.w;
                *reinterpret_cast<uint2 *>(dst + 2 * c) = make_uint2(h16<F>::pack2(o0, o1), h16<F>::pack2(o2, o3));
            }
        } else {
            *reinterpret_cast<uint4 *>(dst + 16 * j) = xb0;
            *reinterpret_cast<uint4 *>(dst + 16 * j + 256) = xb1;
            *reinterpret_cast<uint4 *>(dst + 16 * j + 512) = xb2;
            if constexpr (G::U4 > 3) *reinterpret_cast<uint4 *>(dst + 16 * j + 768) = xb3;
        }
    };
    // the first tile's rows are requested before the 36 (32) weight loads: they are what the first staging waits for
    load(t);
    __builtin_amdgcn_sched_barrier(0);

    // ---- the stationary operand: W[n0 + 16 nb + l15][32 s + 8 kg .. + 8)
    uint4 wf[G::NCB][G::KS];
    const bf16_t *W = (const bf16_t *)p.W;
#pragma unroll
    for (int nb = 0; nb < G::NCB; ++nb) {
        const int n = min(n0 + 16 * nb + l15, p.N - 1);
#pragma unroll
        for (int s = 0; s < G::KS; ++s)
            wf[nb][s] = *reinterpret_cast<const uint4 *>(W + (size_t)n * K + 32 * s + 8 * kg);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LN) {
        for (int c = threadIdx.x; c < K; c += 512) { gam[c] = p.ln_w[c]; bet[c] = p.ln_b[c]; }
    }
    for (int c = threadIdx.x; c < G::NB; c += 512) {                     // this workgroup's bias slice (0 when there is none)
        const int n = group * G::NB + c;
        bia[c] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }

    // residual-form epilogue: the fp32 residual rows of a tile are fetched a whole phase before they are used (a load
    // consumed where it is issued costs a memory round trip per 16 x 16 block: measured 2x on this epilogue in gemm.hip)
    float4 res[EPI == EPI_RESIDUAL ? 2 : 1][EPI == EPI_RESIDUAL ? G::NCB : 1];
    float rsc[2] = {1.0f, 1.0f};
    auto load_res = [&](int tile) {
        if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int m = min(tile * 32 + 16 * rb + l15, p.M - 1);
                rsc[rb] = p.row_scale ? p.row_scale[m / p.row_scale_rows] : 1.0f;
#pragma unroll
                for (int nb = 0; nb < G::NCB; ++nb) {
                    const int n = min(n0 + 16 * nb + 4 * kg, p.N - 4);
                    res[rb][nb] = *reinterpret_cast<const float4 *>(p.residual + (size_t)m * p.N + n);
                }
            }
        }
    };

    f32x4_t acc[2][G::NCB];
    auto mfma_phase = [&](int cur) {
        const unsigned char *a0 = smem + cur * G::BUF + l15 * G::PITCH + 16 * kg;
        const unsigned char *a1 = a0 + 16 * G::PITCH;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int nb = 0; nb < G::NCB; ++nb) acc[rb][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const uint4 f0 = *reinterpret_cast<const uint4 *>(a0 + 64 * s);
            const uint4 f1 = *reinterpret_cast<const uint4 *>(a1 + 64 * s);
#pragma unroll
            for (int nb = 0; nb < G::NCB; ++nb) {
                acc[0][nb] = h16<F>::mfma16(wf[nb][s], f0, acc[0][nb]);
                acc[1][nb] = h16<F>::mfma16(wf[nb][s], f1, acc[1][nb]);
            }
        }
    };
    // lane holds D[n = n0 + 16 nb + 4 kg + i][m = 32 tile + 16 rb + l15].  Residual form: 16-byte fp32 stores straight
    // from the lane (64-byte row pieces).  bf16 form: a lane's 8 bytes would make 32-byte row pieces -- 16 of them per
    // store instruction -- and such stores cost ~300 cycles each (in-kernel stamps: 1 850 .. 2 400 cycles of a 5 800-cycle
    // iteration); the tile therefore goes to an LDS image and leaves at the top of the NEXT iteration (store_tile) as
    // 16 bytes per lane along the rows.
    auto epilogue = [&](int tile, int cslot) {
        float4 bv[G::NCB];                                                // one batch of LDS reads, one wait
#pragma unroll
        for (int nb = 0; nb < G::NCB; ++nb) bv[nb] = *reinterpret_cast<const float4 *>(bia + w * G::NW + 16 * nb + 4 * kg);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int m = tile * 32 + 16 * rb + l15;
#pragma unroll
            for (int nb = 0; nb < G::NCB; ++nb) {
                const int n = n0 + 16 * nb + 4 * kg;
                float v[4] = {acc[rb][nb][0] + bv[nb].x, acc[rb][nb][1] + bv[nb].y, acc[rb][nb][2] + bv[nb].z, acc[rb][nb][3] + bv[nb].w};
                const size_t o = (size_t)m * p.N + n;
                if constexpr (EPI == EPI_RESIDUAL) {
                    if (m < p.M && n < p.N) {
                        const float4 r1 = res[rb][nb];
                        const float rs = rsc[rb];
                        float4 out = make_float4(v[0] * rs + r1.x, v[1] * rs + r1.y, v[2] * rs + r1.z, v[3] * rs + r1.w);
                        if (p.residual2) {
                            const float4 r2 = *reinterpret_cast<const float4 *>(p.residual2 + o);
                            out.x += r2.x; out.y += r2.y; out.z += r2.z; out.w += r2.w;
                        }
                        *reinterpret_cast<float4 *>((float *)p.C + o) = out;
                    }
                } else {
                    if constexpr (ACT != PPT_ACT_NONE) {
                        if (p.C2 && m < p.M && n < p.N)    // (the saved pre-activation: rare, stays on the direct path)
                            *reinterpret_cast<uint2 *>((bf16_t *)p.C2 + o) = make_uint2(h16<F>::pack2(v[0], v[1]), h16<F>::pack2(v[2], v[3]));
                    }
                    if constexpr (ACT == PPT_ACT_GELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = gelu_poly(v[i]);
                    } else if constexpr (ACT == PPT_ACT_QUICKGELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] / (1.0f + __expf(-1.702f * v[i]));
                    }
                    *reinterpret_cast<uint2 *>(cbuf + cslot * G::CBUF + (16 * rb + l15) * G::CPITCH + (w * G::NW + 16 * nb + 4 * kg) * 2) =
                        make_uint2(h16<F>::pack2(v[0], v[1]), h16<F>::pack2(v[2], v[3]));
                }
            }
        }
    };
    auto store_tile = [&](int tile, int cslot) {
        if constexpr (EPI == EPI_BF16) {
            constexpr int CPR = G::NB / 8;                                // 16-byte chunks per row
#pragma unroll
            for (int i = 0; i < 32 * CPR / 512; ++i) {
                const int c = threadIdx.x + 512 * i;
                const int row = c / CPR, ch = c - row * CPR;
                const uint4 v = *reinterpret_cast<const uint4 *>(cbuf + cslot * G::CBUF + row * G::CPITCH + 16 * ch);
                const int m = tile * 32 + row, n = group * G::NB + 8 * ch;
                if (m < p.M && n < p.N) *reinterpret_cast<uint4 *>((bf16_t *)p.C + (size_t)m * p.N + n) = v;
            }
        }
    };

    __syncthreads();                                                     // gamma / beta / bias are in LDS
    stage(0, t);
    load(min(t + n_walk, n_tiles - 1));
    load_res(t);
    __syncthreads();
    // A workgroup puts two waves on every SIMD (w and w + 4).  Run in lockstep -- MFMAs, then epilogue + LayerNorm staging,
    // barrier -- the pair leaves the matrix pipe idle through every vector phase and the vector unit idle through every
    // MFMA phase.  So the second half stages FIRST and multiplies SECOND:
    //   waves 0-3:  store tile t-1 | MFMAs of tile t | stage tile t+1, issue loads of t+2 | epilogue of tile t | barrier
    //   waves 4-7:  stage tile t+1, issue loads of t+2 | store tile t-1 | MFMAs of tile t | epilogue of tile t | barrier
    // Loads are issued a whole phase before anything waits for them, and no store sits directly in front of such a wait
    // (the compiler's s_waitcnt behind conditional stores is vmcnt(0): the wave would wait for the stores to drain).
    constexpr bool ST = STAGGER && !LN;                                  // (with the 24 LayerNorm registers the two code paths spill)
    const bool late = ST && w >= 4;
#ifdef PPT_RG_STAMP
    unsigned long long *stamps = (unsigned long long *)p.residual2;
#endif
    int t_prev = -1, it = 0;
    for (; t < n_tiles; t += n_walk, ++it) {
        const int cur = it & 1;
        RG_STAMP(0);
        if (!late) {
            if (t_prev >= 0) store_tile(t_prev, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_phase(cur);
            __builtin_amdgcn_sched_barrier(0);
            RG_STAMP(1);
            stage(cur ^ 1, min(t + n_walk, n_tiles - 1));                // (that buffer was last read in iteration it - 1, before its barrier)
            __builtin_amdgcn_sched_barrier(0);
            RG_STAMP(2);
            load(min(t + 2 * n_walk, n_tiles - 1));                      // unconditional (a branch parks the registers in scratch)
            __builtin_amdgcn_sched_barrier(0);
        } else {
            stage(cur ^ 1, min(t + n_walk, n_tiles - 1));
            __builtin_amdgcn_sched_barrier(0);
            RG_STAMP(1);
            load(min(t + 2 * n_walk, n_tiles - 1));
            __builtin_amdgcn_sched_barrier(0);
            if (t_prev >= 0) store_tile(t_prev, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_phase(cur);
            __builtin_amdgcn_sched_barrier(0);
            RG_STAMP(2);
        }
        RG_STAMP(3);
        epilogue(t, cur);
        __builtin_amdgcn_sched_barrier(0);
        RG_STAMP(4);
        load_res(min(t + n_walk, n_tiles - 1));
        t_prev = t;
        // (round 6) an LDS-only barrier: __syncthreads() is a release + acquire around s_barrier, i.e. s_waitcnt vmcnt(0) -- it
        // drained the next tiles' row requests and the previous tile's stores at the end of every iteration.  Nothing here hands
        // GLOBAL data from wave to wave; the A buffers and the C image are LDS.  PPT_RG_SYNC=1 (compile time) restores it for A/B.
#ifdef PPT_RG_SYNC
        __syncthreads();
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#endif
        RG_STAMP(5);
    }
    if (t_prev >= 0) store_tile(t_prev, (it - 1) & 1);
}

template <typename F, int K, bool LN, int EPI, int ACT>
int launch(const ppt_rowgemm_params &p, hipStream_t s, int cus)
{
    using G = RG<K>;
    static const int once = [] {
        (void)hipFuncSetAttribute((const void *)rowgemm_kernel<F, K, LN, EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        return 0;
    }();
    (void)once;
    ppt_rowgemm_params q = p;
    q.groups = (p.N + G::NB - 1) / G::NB;
    const int tiles = (p.M + 31) / 32;
    // walkers per column group: a multiple of 8 (the XCD-aware id mapping), as many as fill the chip -- or the share of it the
    // caller leaves to this stream (ppt_set_persistent_occupancy: these workgroups take a whole CU each) --, no more than tiles
    static const int honour = [] { const char *e = getenv("PPT_ROWGEMM_OCCUPANCY"); return e ? atoi(e) : 1; }();
    if (p.walkers <= 0 && honour) cus = cus * ppt_get_persistent_occupancy() / 100;
    int walkers = p.walkers > 0 ? p.walkers : cus / q.groups;
    if (walkers > tiles) walkers = tiles;
    walkers = (walkers + 7) / 8 * 8;
    if (walkers * q.groups > cus && walkers > 8 && p.walkers <= 0) walkers -= 8;
    q.walkers = walkers;
    hipLaunchKernelGGL((rowgemm_kernel<F, K, LN, EPI, ACT>), dim3(walkers * q.groups), dim3(512), G::LDS, s, q);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PPT_OK : PPT_ELAUNCH;
}

}  // namespace

extern "C" int ppt_rowgemm_bf16(const ppt_rowgemm_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    const ppt_rowgemm_params &p = *pp;
    if (!p.A || !p.W || !p.C || p.M <= 0 || p.N <= 0) return PPT_EINVAL;
    if (p.K != 384 && p.K != 512) return PPT_EUNSUPPORTED;
    if (p.dtype != 0 && p.dtype != PPT_BF16 && p.dtype != PPT_F16) return PPT_EINVAL;          /* (0: ABI-2 callers, bf16) */
    if (p.N % (p.residual_form ? 4 : 8)) return PPT_EUNSUPPORTED;
    if (p.a_ln && (!p.ln_w || !p.ln_b)) return PPT_EINVAL;
    if ((p.ln_mean == nullptr) != (p.ln_rstd == nullptr)) return PPT_EINVAL;
    if (p.residual_form && !p.residual) return PPT_EINVAL;
    if (p.residual_form && p.a_ln) return PPT_EUNSUPPORTED;          /* (no caller: a LayerNorm is never followed by a residual-form linear) */
    if (p.residual_form && (p.act != PPT_ACT_NONE || p.C2)) return PPT_EUNSUPPORTED;
    if (p.C2 && p.act == PPT_ACT_NONE) return PPT_EUNSUPPORTED;     /* the second output is the PRE-activation of an activated launch */
    if (p.row_scale && p.row_scale_rows <= 0) return PPT_EINVAL;
    if (p.act != PPT_ACT_NONE && p.act != PPT_ACT_GELU && p.act != PPT_ACT_QUICKGELU) return PPT_EUNSUPPORTED;
    if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.C | (uintptr_t)p.C2 | (uintptr_t)p.residual | (uintptr_t)p.residual2) & 15)
        return PPT_EINVAL;
    const int cus = ppt_cu_count(ppt_stream(stream));            // (of the stream's device, not process-global state)
    hipStream_t s = ppt_stream(stream);
#define PPT_RG_ACT(FF, KK, LNV)                                                                                        \
    (p.act == PPT_ACT_GELU ? launch<FF, KK, LNV, EPI_BF16, PPT_ACT_GELU>(p, s, cus)                                      \
     : p.act == PPT_ACT_QUICKGELU ? launch<FF, KK, LNV, EPI_BF16, PPT_ACT_QUICKGELU>(p, s, cus)                          \
                                  : launch<FF, KK, LNV, EPI_BF16, PPT_ACT_NONE>(p, s, cus))
#define PPT_RG(FF, KK)                                                                                               \
    (p.a_ln ? PPT_RG_ACT(FF, KK, true) : (p.residual_form ? launch<FF, KK, false, EPI_RESIDUAL, PPT_ACT_NONE>(p, s, cus) : PPT_RG_ACT(FF, KK, false)))
    if (p.dtype == PPT_F16) return p.K == 384 ? PPT_RG(f16_t, 384) : PPT_RG(f16_t, 512);
    return p.K == 384 ? PPT_RG(bf16_t, 384) : PPT_RG(bf16_t, 512);
#undef PPT_RG_ACT
#undef PPT_RG
}
