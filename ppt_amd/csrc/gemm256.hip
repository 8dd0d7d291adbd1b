// gemm256.hip -- the 256-row macro-tile GEMM core (round 5): C[M,N] = epilogue(A[M,K] . B[N,K]^T), 16-bit operands.
//
// Why another tile loop.  Every kernel of gemm.hip moves its operands L2 -> LDS, and what a CU takes in from L2 is ~30 B/clk
// whatever is in flight (16 TB/s over the chip at 16 issuing waves).  A tile loop therefore tops out at
// (FLOP per fill byte) x 16 TB/s: 64 FLOP/B for the 128 x 128 tile = 1.0 PFLOP/s at the very best, 0.15-0.25 of the 2.5 PFLOP/s peak
// in practice.  A 256 x 256 tile needs half the bytes per FLOP (128 FLOP/B), a 256 x 128 tile two thirds (85 FLOP/B).
//
// Geometry.  512 threads = 8 waves, ONE workgroup per CU for the 256 x 256 tile (waves 2 x 4, each 128 x 64 = 4 x 2 MFMA tiles of
// 32 x 32: 128 accumulator registers), two for the 256 x 128 tile (waves 4 x 2, each 64 x 64: 64 accumulators, 128 VGPRs per wave).
// K is walked in 64-BYTE half-slabs (32 values) through a ring of LDS stages filled by LDS-DMA (global_load_lds_dwordx4, the
// XOR swizzle applied to the source address: ppt_common.h lds_dma16, gemm.hip gemm_kernel_glds_h), counted vmcnt, one raw s_barrier
// per stage, NSTAGE - 1 stages in flight.  The epilogues are gemm_common.h's: the register-layout one (bias / GELU / per-group term +
// BatchNorm partials; compile-time variants) and, for row-major fp32 operands (the residual stream of proj / fc2), the 16-byte LDS
// walk over an fp32 park, taken in two 64-row halves for the 128-row wave tile.
#include "gemm_common.h"

namespace {

constexpr int NT2 = 512, ROWH = 64;
#ifdef PPT_GEMM256_WHOLE_EPILOGUE
constexpr bool SLICED_EPILOGUE = false;        // (A/B builds: the whole-tile register epilogue of gemm_common.h)
#else
constexpr bool SLICED_EPILOGUE = true;
#endif

// Diagnostic build only (tools/gemm256_stamp.py compiles this file with -DPPT_GEMM_STAMP into its own library): lane 0 of every wave
// stores s_memtime at entry / prologue DMA issued / first stage readable / K loop done / stores done into the buffer p.pool_min
// points at (unused by these launches; ppt_gemm256 rejects pool_max).
#ifdef PPT_GEMM_STAMP
#define G256_STAMP(slot) do { if (lane == 0 && p.pool_min) reinterpret_cast<unsigned long long *>(p.pool_min)[((size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 8 + w)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G256_STAMP(slot) do { } while (0)
#endif
__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * ROWH + ((chunk ^ ((row >> 2) & 3)) << 4); }

// ROWS x 64 bytes of an operand -> LDS, by the NW waves of the workgroup: ROWS / (16 NW) pieces of 1 KiB (16 rows) per wave
template <typename T, int ROWS, int NW = 8>
__device__ __forceinline__ void glds_half8(const T *base, int64_t ld, int rows, int r0, int k0, unsigned char *tile, int w, int lane)
{
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PER_WAVE = ROWS / (16 * NW);
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int rg = (w * PER_WAVE + i) * 16;
        const int r = rg + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);       // source chunk that belongs in LDS slot lane&3 of row r
        const T *src = base + (int64_t)min(r0 + r, rows - 1) * ld + k0 + c * EPC;
        lds_dma16(src, tile + rg * ROWH);
    }
}

template <typename T, int TI, int TJ>
__device__ __forceinline__ void mma_half8(const unsigned char *As, const unsigned char *Bs, int arow0, int brow0, int lane,
                                          f32x16_t (&acc)[TI][TJ])
{
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        uint4 a[TI], b[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
            b[j] = *reinterpret_cast<const uint4 *>(Bs + lds_off_h(brow0 + j * 32 + r, kk * 2 + h));
#pragma unroll
        for (int i = 0; i < TI; ++i)
            a[i] = *reinterpret_cast<const uint4 *>(As + lds_off_h(arow0 + i * 32 + r, kk * 2 + h));
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = h16<T>::mfma32(a[i], b[j], acc[i][j]);
    }
}

// The two halves of a stage's work as separate steps (the staggered schedule below): all fragments of a 32-deep stage into
// registers (6 x 2 ds_read_b128 for the 128 x 64 wave tile), then 16 MFMAs on registers only.
template <int TI, int TJ> struct StageFrags { uint4 a[2][TI], b[2][TJ]; };

template <int TI, int TJ>
__device__ __forceinline__ void load_frags(StageFrags<TI, TJ> &f, const unsigned char *As, const unsigned char *Bs, int arow0, int brow0, int lane)
{
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
            f.b[kk][j] = *reinterpret_cast<const uint4 *>(Bs + lds_off_h(brow0 + j * 32 + r, kk * 2 + h));
#pragma unroll
        for (int i = 0; i < TI; ++i)
            f.a[kk][i] = *reinterpret_cast<const uint4 *>(As + lds_off_h(arow0 + i * 32 + r, kk * 2 + h));
    }
}

template <typename T, int TI, int TJ>
__device__ __forceinline__ void mma_frags(const StageFrags<TI, TJ> &f, f32x16_t (&acc)[TI][TJ])
{
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = h16<T>::mfma32(f.a[kk][i], f.b[kk][j], acc[i][j]);
}

// epilogue_regs (gemm_common.h) taken ONE 32-ROW SLICE of the wave tile at a time: the slice's values (bias, per-group term, BatchNorm
// partials, activation -- the same expressions in the same order: same bits) are parked as 16-bit, read back as 16-byte row pieces and
// stored, and the NEXT slice's arithmetic is issued while those stores drain.  Why: a CU retires ~12 B/clk of global stores whatever it
// does (in-kernel stamps, tools/gemm256_stamp.py: 9 900 cycles for the 128 KiB of a 256 x 256 tile, 18 500 with the GELU in front of
// them), and with one workgroup per CU nothing else covers that tail; sliced, the tail is max(stores, arithmetic) instead of their sum.
// The park is 4 KiB per wave.
template <int TI, int TJ, int EPI>
__device__ __forceinline__ void epilogue_regs_sliced(const ppt_gemm_params &p, f32x16_t (&acc)[TI][TJ], const EpiPre<TI, TJ> &pre,
                                                     unsigned char *park, int lane, int mw, int nw, int64_t zc)
{
    constexpr int act = EPI >> EPI_ACT_SHIFT;
    constexpr bool has_group = (EPI & EPI_GROUP) != 0, has_stats = (EPI & EPI_STATS) != 0;
    static_assert(TJ == 2 && (EPI & EPI_POOL) == 0, "64-column wave tiles, no pooling");
    constexpr int WN = TJ * 32, ROWBYTES = WN * 2;
    const int cl = lane & 31, h = lane >> 5, odd = cl & 1;
    const int lane_byte = (4 * h + odd) * ROWBYTES + (cl >> 1) * 4;
    constexpr int CPR = WN / 8, RP = 64 / CPR;                        // 16-byte chunks per row, rows per pass
    const int ch = lane % CPR, rl = lane / CPR;
    uint16_t *C = reinterpret_cast<uint16_t *>(p.C) + zc;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int mg = mw + i * 32;                                   // first row of this slice (wave-uniform)
        const bool full = mg + 32 <= p.M;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n = nw + j * 32 + cl;
            const bool nok = n < p.N;
            const float bias = pre.bias[j];
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bias;
            if (has_group) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += pre.g[i][j][r < 8 ? 0 : 1];
            }
            if (has_stats && mg < p.M) {
                bool rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = full || (mg + (r & 3) + 8 * (r >> 2) + 4 * h) < p.M;
                float sacc = 0.f, q = 0.f, mean;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += v[r];
                    sacc = xor32_sum(sacc);
                    mean = sacc * (1.0f / 32.0f);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float d = v[r] - mean; q = fmaf(d, d, q); }
                    q = xor32_sum(q);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += rv[r] ? v[r] : 0.f;
                    sacc = xor32_sum(sacc);
                    mean = sacc / (float)min(32, p.M - mg);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float d = v[r] - mean; q = rv[r] ? fmaf(d, d, q) : q; }
                    q = xor32_sum(q);
                }
                if (h == 0 && nok) {
                    p.col_sum[(int64_t)(mg >> 5) * p.N + n] = sacc;
                    p.col_sqsum[(int64_t)(mg >> 5) * p.N + n] = q;
                }
            }
            if (act != PPT_ACT_NONE) act_fwd_n<true, 16>(v, act);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // even lane keeps row(r): (own, neighbour's); odd lane keeps row(r+1): (neighbour's, own)
                const float give = odd ? v[r] : v[r + 1];
                const float got = __uint_as_float(dpp_mov<0xB1, 0xf>(__float_as_uint(give)));   // quad_perm [1,0,3,2]
                const uint32_t wd = pack2_dt(p.c_dtype, odd ? got : v[r], odd ? v[r + 1] : got);
                const int jb = (j ^ odd) << 6;                        // the written row (r + odd) is odd exactly on odd lanes
                *reinterpret_cast<uint32_t *>(park + ((r & 3) + 8 * (r >> 2)) * ROWBYTES + jb + lane_byte) = wd;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int pass = 0; pass < 32 / RP; ++pass) {
            const int row = pass * RP + rl;
            const int m = mg + row;
            const uint4 d = *reinterpret_cast<const uint4 *>(park + row * ROWBYTES + ((ch * 16) ^ ((row & 1) << 6)));
            const int n = nw + ch * 8;
            if (m < p.M && n < p.N) *reinterpret_cast<uint4 *>(C + (int64_t)m * p.ldc + n) = d;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // (the slice buffer is rewritten by the next slice)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// rows [HALF * PR, (HALF + 1) * PR) of a wave's accumulators -> fp32 park -> epilogue_vec8 (compile-time HALF: a runtime index into
// the accumulator array would send it to scratch)
template <int HALF, int PR, int TI, int TJ>
__device__ __forceinline__ void park_walk(const ppt_gemm_params &p, f32x16_t (&acc)[TI][TJ], float *ct, int lane, int mw, int nw, int64_t zc)
{
    constexpr int WN = TJ * 32;
    const int h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < PR / 32; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * WN + j * 32 + cl] = acc[HALF * (PR / 32) + i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    epilogue_vec8<PR, WN, 1>(p, ct, lane, mw + HALF * PR, nw, zc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int BN, int NSTAGE, int EPI, bool PP>
// (the LDS-walk epilogue of the EPI < 0 kernels needs more than the 128 registers that two workgroups per CU would leave)
__global__ __launch_bounds__(NT2, (BN == 256 || EPI < 0) ? 2 : 4) void gemm256_kernel(const ppt_gemm_params p)
{
    constexpr int BM = 256;
    constexpr int WGM = BN == 256 ? 2 : 4, WGN = 8 / WGM;                     // wave grid
    constexpr int WM = BM / WGM, WN = BN / WGN, TI = WM / 32, TJ = WN / 32;    // 128 x 64 (4 x 2 tiles) or 64 x 64 (2 x 2)
    constexpr int A_BYTES = BM * ROWH, B_BYTES = BN * ROWH, STAGE = A_BYTES + B_BYTES;
    constexpr int LOADS = BM / 128 + BN / 128;                                 // LDS-DMA instructions per wave per stage
    constexpr int PARK16 = 8 * WM * WN * 2;                                    // 16-bit park of epilogue_regs
    constexpr int PR = 32;                                                     // rows of a wave tile parked in fp32 at a time
    constexpr int PARK32 = 8 * PR * 64 * 4;                                    // fp32 park of epilogue_vec8
    constexpr int RING = NSTAGE * STAGE;
    constexpr int SMEM = RING > PARK16 ? (RING > PARK32 ? RING : PARK32) : (PARK16 > PARK32 ? PARK16 : PARK32);
    __shared__ __align__(16) unsigned char smem[SMEM];
    constexpr int BK = ROWH / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    G256_STAMP(0);
    PPT_PRIO(p.wave_prio);
    const int wm = w / WGN, wn = w % WGN;
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;                   // same XCD-aware remap as gemm_kernel
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    // ... and inside an XCD's run of tiles, GROUPS of GM tile rows are walked column-major: the ~32 tiles an XCD works on at one
    // time then cover GM row panels x 32 / GM column panels instead of 1 x 32, so every operand slab is pulled over the fabric
    // once per GM (A) resp. 32 / GM (B) tiles instead of A once per 32 and B once per TILE -- what matters when B does not fit the
    // 4 MiB L2 (8192^3: 33 -> 12 panel streams per 32 tiles); for the tower's N <= 1536 it changes nothing measurable.
    int tm, tn;
    {
        constexpr int GM = 4;
        const int ncol = gridDim.x, nrow = gridDim.y;
        const int grp = lin / (GM * ncol), first = grp * GM;
        const int rows_here = min(GM, nrow - first);
        const int t = lin - grp * GM * ncol;
        tm = first + t % rows_here;
        tn = t / rows_here;
    }
    const int n0 = tn * BN, m0 = tm * BM;
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)blockIdx.z * p.strideA;
    const T *B = reinterpret_cast<const T *>(p.B) + (int64_t)blockIdx.z * p.strideB;

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nslab = p.K / BK, last = nslab - 1;
    auto issue = [&](int slab, int stage) {
        glds_half8<T, BM>(A, p.lda, p.M, m0, slab * BK, smem + stage * STAGE, w, lane);
        glds_half8<T, BN>(B, p.ldb, p.N, n0, slab * BK, smem + stage * STAGE + A_BYTES, w, lane);
    };
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) issue(min(i, last), i);
    EpiPre<TI, TJ> epre;                                  // (behind the first stages: the K loop must not start later for it)
    epilogue_prefetch<TI, TJ, (EPI < 0 || (EPI & EPI_GROUP) != 0)>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);
    G256_STAMP(1);
    int stage = 0;
    if constexpr (PP) {
        // STAGGERED schedule (round 5, v2): a wave alternates LOAD(s) -- the stage's fragments into registers, the LDS-DMA of slab
        // s + 3, the counted wait -- and COMPUTE(s) -- 16 MFMAs on registers -- with a workgroup barrier after each; waves 4-7 run
        // ONE SEGMENT BEHIND waves 0-3 (one extra barrier up front, one less at the end).  Every SIMD holds one wave of each
        // half, so while one computes the other loads: the matrix pipe never waits for a barrier or an LDS read.  (v1 above lets
        // all 8 waves meet at one barrier per stage and then issue DMA + reads together: measured 0.36 of peak at 8192^3.)
        // Slab s is read in segments 2s (waves 0-3) and 2s + 1 (waves 4-7); its slot is refilled with slab s + 4, issued in
        // segments >= 2s + 2; a wave's wait at the end of LOAD(s) covers its pieces of slab s + 1, two segments before anyone
        // reads them.
        static_assert(NSTAGE == 4, "slot arithmetic below");
        const bool late = w >= 4;
        StageFrags<TI, TJ> fr;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * 2) : "memory");        // own pieces of slab 0 have landed
        __builtin_amdgcn_s_barrier();                                              // ... everybody's
        G256_STAMP(2);
        if (late) __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nslab; ++s) {
            load_frags<TI, TJ>(fr, smem + stage * STAGE, smem + stage * STAGE + A_BYTES, wm * WM, wn * WN, lane);
            issue(min(s + 3, last), (stage + 3) & 3);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" :: "n"(LOADS * 2) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma_frags<T, TI, TJ>(fr, acc);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stage = (stage + 1) & 3;
        }
        if (!late) __builtin_amdgcn_s_barrier();
    } else {
    for (int s = 0; s < nslab; ++s) {
        // own copies of slab s have landed (the LOADS * (NSTAGE - 2) youngest, slabs s+1 .., may still fly) ...
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NSTAGE - 2)) : "memory");
        __builtin_amdgcn_s_barrier();                     // ... and so have everybody else's: slab s is readable
        if (s == 0) G256_STAMP(2);
        int nstage = stage + NSTAGE - 1; if (nstage >= NSTAGE) nstage -= NSTAGE;
        issue(min(s + NSTAGE - 1, last), nstage);         // that stage was last read at slab s-1, before this barrier
        mma_half8<T, TI, TJ>(smem + stage * STAGE, smem + stage * STAGE + A_BYTES, wm * WM, wn * WN, lane, acc);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // surplus prefetches must not land on the parked tile
    __builtin_amdgcn_s_barrier();
    G256_STAMP(3);

    const int64_t zc = (int64_t)blockIdx.z * p.strideC;
    const int mw = m0 + wm * WM, nw = n0 + wn * WN;
    if constexpr (EPI >= 0) {
        if constexpr (SLICED_EPILOGUE) epilogue_regs_sliced<TI, TJ, EPI>(p, acc, epre, smem + w * (32 * WN * 2), lane, mw, nw, zc);
        else epilogue_regs<TI, TJ, EPI, 1>(p, acc, epre, smem + w * (WM * WN * 2), lane, mw, nw, zc);
    } else {
        // row-major fp32 operands (residual stream, saved pre-activation, second output): the LDS walk, PR rows of the wave tile
        // at a time (an fp32 park of the whole 128 x 64 wave tile would be 256 KiB per workgroup)
        static_assert(WN == 64 && (WM / PR == 2 || WM / PR == 4), "park geometry");
        float *ct = reinterpret_cast<float *>(smem) + w * (PR * 64);
        park_walk<0, PR, TI, TJ>(p, acc, ct, lane, mw, nw, zc);
        park_walk<1, PR, TI, TJ>(p, acc, ct, lane, mw, nw, zc);
        if constexpr (WM / PR == 4) {
            park_walk<2, PR, TI, TJ>(p, acc, ct, lane, mw, nw, zc);
            park_walk<3, PR, TI, TJ>(p, acc, ct, lane, mw, nw, zc);
        }
    }
#ifdef PPT_GEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G256_STAMP(4);
#endif
}

// (Removed in round 6: the half-height tile -- 128 x 256 per 256-thread workgroup, two workgroups per CU so that one's epilogue
// runs under the other's K loop.  Measured in round 5 (profiles/r05_gemm_core.md): no gain over the 256 x 256 tile -- fc1 47.4 vs
// 45.4 us, qkv 31.0 vs 32.2 us at 16 416 rows; what it wins back under the other workgroup's K loop it loses to the 1.5 x fill
// bytes per FLOP.)

extern "C" int ppt_get_gemm256(void);

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <typename T, bool PPV>
int launch256(const ppt_gemm_params &p, int bn, hipStream_t s)
{
    const int batch = p.batch > 0 ? p.batch : 1;
    dim3 grid((p.N + bn - 1) / bn, (p.M + 255) / 256, batch);
    if (grid.y > 65535 || grid.z > 65535) return PPT_EUNSUPPORTED;
    // the register-layout epilogue exists as three compile-time variants (plain / bias + GELU / per-group term + BatchNorm
    // partials: qkv, fc1, conv3); everything else takes the LDS walk (EPI = -1), which has no statistics / group term
    int epi = (p.C && reg_epilogue_ok<4>(p, 0)) ? epi_mask(p) : -1;
    if (epi != 0 && epi != EPI_GELU && !(bn == 256 && epi == (EPI_GROUP | EPI_STATS))) epi = -1;
    if (bn == 256) {
        switch (epi) {
        case 0: hipLaunchKernelGGL((gemm256_kernel<T, 256, 4, 0, PPV>), grid, dim3(NT2), 0, s, p); break;
        case EPI_GELU: hipLaunchKernelGGL((gemm256_kernel<T, 256, 4, EPI_GELU, PPV>), grid, dim3(NT2), 0, s, p); break;
        case EPI_GROUP | EPI_STATS: hipLaunchKernelGGL((gemm256_kernel<T, 256, 4, EPI_GROUP | EPI_STATS, PPV>), grid, dim3(NT2), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm256_kernel<T, 256, 4, -1, PPV>), grid, dim3(NT2), 0, s, p); break;
        }
    } else {
        switch (epi) {
        case 0: hipLaunchKernelGGL((gemm256_kernel<T, 128, 3, 0, false>), grid, dim3(NT2), 0, s, p); break;
        case EPI_GELU: hipLaunchKernelGGL((gemm256_kernel<T, 128, 3, EPI_GELU, false>), grid, dim3(NT2), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm256_kernel<T, 128, 3, -1, false>), grid, dim3(NT2), 0, s, p); break;
        }
    }
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// ---- split16 on the 256 x 128 tile (fp32 operands as hi + lo half pairs: gemm_common.h) -----------------------------------------
// The split16 products are bound by the L2 -> CU fill -- fp32 operands are twice half's bytes and a CU pulls 21-25 B/clk whatever
// the tile (tools/split16_bench.py: 64 x 64 and 128 x 64 tiles tie at 116 us on fc2 of a C2 batch) -- so what a tile computes per
// byte it loads is its speed: 16 FLOP/B at 64 x 64, 32 at 128 x 128, 43 at 256 x 128.  This is gemm.hip's register-staged split
// loop (global -> registers, three slabs deep -> split -> hi / lo LDS images -> 16-bit fragments) on this file's 512-thread frame:
// waves 4 x 2, wave tile 64 x 64, two 48 KiB LDS buffers, the grouped tile order, the 32-row fp32 LDS-walk epilogue.  Same K
// order, same split, same epilogue arithmetic as the 64 x 64 / 128 x 128 split kernels: the same bits.
template <int NR>
__device__ __forceinline__ void load_rows512(Stage<NR> &st, const float *base, int64_t ld, int rows, int r0, int k0)
{
    const int t = threadIdx.x, ch = t & 7;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = r0 + (t >> 3) + 64 * i;               // (rows beyond the matrix read its last row: their products are never stored)
        st.v[i] = *reinterpret_cast<const uint4 *>(base + (int64_t)min(r, rows - 1) * ld + k0 + ch * 4);
    }
}

template <int BM, int BN, int TI, int TJ>
__device__ __forceinline__ void split256_loop(const ppt_gemm_params &p, const float *A, const float *B, int m0, int n0, unsigned char *smem,
                                              int arow0, int brow0, int lane, f32x16_t (&acc)[TI][TJ], float sa, float sb, uint32_t &over)
{
    constexpr int NRA = BM / 64, NRB = BN / 64, A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    const int nslab = p.K / 32, last = nslab - 1;
    Stage<NRA> a0, a1, a2;
    Stage<NRB> b0, b1, b2;
#define S256_LOAD(SA, SB, S)                                            \
    do {                                                                \
        load_rows512<NRA>(SA, A, p.lda, p.M, m0, (S) * 32);             \
        load_rows512<NRB>(SB, B, p.ldb, p.N, n0, (S) * 32);             \
    } while (0)
#define S256_WRITE(SA, SB, BUFI)                                                                    \
    do {                                                                                            \
        write_stage_split<NRA, BM, 64>(SA, smem + ((BUFI) & 1) * BUF, sa, over);                    \
        write_stage_split<NRB, BN, 64>(SB, smem + ((BUFI) & 1) * BUF + A_BYTES, sb, over);          \
    } while (0)
#define S256_X(S) mma_slab_split<TI, TJ, BM, BN>(smem + ((S) & 1) * BUF, smem + ((S) & 1) * BUF + A_BYTES, arow0, brow0, lane, acc)
    // STEP(s): multiply slab s out of LDS (X); the FREE register set (it held slab s) receives slab s + 3; the NEXT set (slab s + 1)
    // is split into the other LDS buffer; one barrier per slab.  Loads and writes are unconditional (slab index clamped): the
    // compiler's counted vmcnt keeps two slabs in flight.
    // (Measured and not kept: the two wave groups STAGGERED -- waves 0-3 X | barrier | split + write | barrier, waves 4-7 the two
    // halves in the other order, so that each SIMD's matrix pipe and vector ALU work at the same time: fc2 of a C2 batch 78 -> 82 us,
    // 4096^3 405 -> 434 us as two straight-line loops; 135 us with the branch inside the loop, where the compiler drains the global
    // loads.  What the loop waits for is the operand bytes -- ~16 B/clk/CU through global_load_dwordx4 -- not its own pipes.)
#define S256_STEP(S, FA, FB, NA, NB)                                    \
    {                                                                   \
        S256_LOAD(FA, FB, min((S) + 3, last));                          \
        S256_X(S);                                                      \
        S256_WRITE(NA, NB, (S) + 1);                                    \
        __syncthreads();                                                \
    }
    S256_LOAD(a0, b0, 0);
    S256_LOAD(a1, b1, min(1, last));
    S256_LOAD(a2, b2, min(2, last));
    S256_WRITE(a0, b0, 0);
    __syncthreads();
    for (int s = 0;; s += 3) {
        S256_STEP(s, a0, b0, a1, b1)
        if (s + 1 >= nslab) break;
        S256_STEP(s + 1, a1, b1, a2, b2)
        if (s + 2 >= nslab) break;
        S256_STEP(s + 2, a2, b2, a0, b0)
        if (s + 3 >= nslab) break;
    }
#undef S256_STEP
#undef S256_X
#undef S256_WRITE
#undef S256_LOAD
}

template <int BN>
__global__ __launch_bounds__(NT2, 2) void gemm256s_kernel(const ppt_gemm_params p)
{
    constexpr int BM = 256, WGN = 2, WM = 64, WN = 64, TI = 2, TJ = 2;
    static_assert(BN == 128, "wave grid 4 x 2 of 64 x 64 tiles");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;      // hi + lo images: the fp32 image's bytes
    constexpr int PR = 32, PARK32 = 8 * PR * 64 * 4;
    __shared__ __align__(16) unsigned char smem[2 * BUF > PARK32 ? 2 * BUF : PARK32];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PPT_PRIO(p.wave_prio);
    const int wm = w / WGN, wn = w % WGN;
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;                   // XCD-aware remap + grouped tile order (gemm256_kernel)
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    int tm, tn;
    {
        constexpr int GM = 4;
        const int ncol = gridDim.x, nrow = gridDim.y;
        const int grp = lin / (GM * ncol), first = grp * GM;
        const int rows_here = min(GM, nrow - first);
        const int t = lin - grp * GM * ncol;
        tm = first + t % rows_here;
        tn = t / rows_here;
    }
    const int n0 = tn * BN, m0 = tm * BM;
    const float *A = reinterpret_cast<const float *>(p.A) + (int64_t)blockIdx.z * p.strideA;
    const float *B = reinterpret_cast<const float *>(p.B) + (int64_t)blockIdx.z * p.strideB;
    const float sa = pow2f(p.split_a_pow2), sb = pow2f(p.split_b_pow2);

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    uint32_t split_over = 0;
    split256_loop<BM, BN, TI, TJ>(p, A, B, m0, n0, smem, wm * WM, wn * WN, lane, acc, sa, sb, split_over);
    scale_acc<TI, TJ>(acc, pow2f(-(p.split_a_pow2 + p.split_b_pow2)));
    split_report(split_over, p.split_overflow);

    const int64_t zc = (int64_t)blockIdx.z * p.strideC;
    const int mw = m0 + wm * WM, nw = n0 + wn * WN;
    float *ct = reinterpret_cast<float *>(smem) + w * (PR * 64);
    park_walk<0, PR, TI, TJ>(p, acc, ct, lane, mw, nw, zc);
    park_walk<1, PR, TI, TJ>(p, acc, ct, lane, mw, nw, zc);
}

}  // namespace

// split16 (ppt_gemm_params.split16, fp32 operands): the 256 x 128 tile when the problem is large and plain enough for it.
// PPT_OK when launched, PPT_EUNSUPPORTED when gemm.hip's 128 x 128 / 64 x 64 split kernels should take it.
extern "C" int ppt_gemm256_split_dispatch(const ppt_gemm_params *pp, void *stream)
{
    const ppt_gemm_params &p = *pp;
    static const int enabled = env_int("PPT_SPLIT16_256", 1);
    static const int min_tiles = env_int("PPT_SPLIT16_256_MIN_TILES", 128);
    if (!enabled || p.dtype != PPT_F32 || !p.split16 || p.split16 == 2) return PPT_EUNSUPPORTED;      // (2: the caller asks for the tile loops)
    if (p.a_mode != PPT_A_PLAIN || !p.A || (p.K % 32) != 0 || p.K < 96) return PPT_EUNSUPPORTED;
    if (p.pool_max || p.col_sum || p.group_add) return PPT_EUNSUPPORTED;     // (the LDS-walk epilogue: no statistics, no pools)
    bool ok = (p.N % 8) == 0;
    if (p.C) ok = ok && (p.ldc % 8) == 0 && al16(p.C);
    if (p.C2) ok = ok && (p.ldc2 % 8) == 0 && al16(p.C2);
    if (p.bias) ok = ok && al16(p.bias);
    if (p.dact_pre) ok = ok && (p.ld_dact % 8) == 0 && al16(p.dact_pre);
    if (p.residual) ok = ok && (p.ld_res % 8) == 0 && al16(p.residual);
    if (p.residual2) ok = ok && (p.ld_res2 % 8) == 0 && al16(p.residual2);
    if (p.batch > 1 && ((p.strideC % 8) != 0)) ok = false;
    if (!ok) return PPT_EUNSUPPORTED;
    dim3 grid((p.N + 127) / 128, (p.M + 255) / 256, p.batch > 0 ? p.batch : 1);
    if ((int64_t)grid.x * grid.y * grid.z < min_tiles || grid.y > 65535 || grid.z > 65535) return PPT_EUNSUPPORTED;
    hipLaunchKernelGGL((gemm256s_kernel<128>), grid, dim3(NT2), 0, ppt_stream(stream), p);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// The host side of the choice.  force: 0 = ppt_gemm's automatic dispatch (size gates), 1 = explicit call (only the hard limits).
// Returns PPT_OK when launched, PPT_EUNSUPPORTED when this core does not take the problem.
extern "C" int ppt_gemm256_dispatch(const ppt_gemm_params *pp, int force, void *stream)
{
    const ppt_gemm_params &p = *pp;
    if (p.dtype != PPT_BF16 && p.dtype != PPT_F16) return PPT_EUNSUPPORTED;
    if (p.a_mode != PPT_A_PLAIN || !p.A || (p.K % 32) != 0 || p.K < 64) return PPT_EUNSUPPORTED;
    if (p.batch > 1 && (p.strideC % 8) != 0) return PPT_EUNSUPPORTED;
    if (p.pool_max) return PPT_EUNSUPPORTED;               // (pools over 64-row groups assume TI == 2 wave tiles)
    // tile width: 256 columns when N fills them (N % 256 == 0, or wide enough that the ragged last tile is a small share),
    // else 128 (N = 384: three tiles instead of two with a quarter of the second one empty)
    static const int force_bn = env_int("PPT_GEMM256_BN", 0);       // (experiments: 128 / 256 forces the tile width)
    const int bn = force_bn ? force_bn : ((p.N % 256 == 0 || (p.N >= 1024 && (p.N % 256) >= 128)) ? 256 : 128);
    const bool regs = p.C && reg_epilogue_ok<4>(p, 0);
    const int mask = regs ? epi_mask(p) : -1;
    const bool special = mask == 0 || mask == EPI_GELU || (bn == 256 && mask == (EPI_GROUP | EPI_STATS));
    if (!special) {                                       // the LDS-walk epilogue: 16-byte friendly operands only, no statistics
        if (p.col_sum || p.group_add) return PPT_EUNSUPPORTED;
        bool ok = (p.N % 8) == 0;
        if (p.C) ok = ok && (p.ldc % 8) == 0 && al16(p.C);
        if (p.C2) ok = ok && (p.ldc2 % 8) == 0 && al16(p.C2);
        if (p.bias) ok = ok && al16(p.bias);
        if (p.dact_pre) ok = ok && (p.ld_dact % 8) == 0 && al16(p.dact_pre);
        if (p.residual) ok = ok && (p.ld_res % 8) == 0 && al16(p.residual);
        if (p.residual2) ok = ok && (p.ld_res2 % 8) == 0 && al16(p.residual2);
        if (p.batch > 1 && ((p.strideC % 8) != 0)) ok = false;
        if (!ok) return PPT_EUNSUPPORTED;
    }
    static const int enabled = env_int("PPT_GEMM256", 1);
    static const int min_rows = env_int("PPT_GEMM256_MIN_ROWS", 8192);
    if (!force) {
        // Where this core wins (tools/gemm256_bench.py, same-process interleaved A/B against the 64 x 64 / 128 x 128 loops;
        // profiles/r05_gemm_core.md): every long-K problem (K >= 768: +7 ... +50 %), and short-K ones whose epilogue is a plain /
        // residual store and whose grid is at least two full rounds of the 256 CUs (32 768 x 1536 x 384: +20 %).  Where it loses:
        // short K with an activation epilogue or a 1.5-round grid (fc1 / qkv of a C2 batch: 16 416 x 1536 x 384 + GELU 47 vs 42 us)
        // -- with ONE workgroup per CU nothing covers the tile's store tail (a CU retires ~12 B/clk of global stores: 128 KiB of
        // C = 9 900 cycles, the GELU in front of it as long again, against 18 000 cycles of K loop at K = 384).
        const int mode = ppt_get_gemm256();
        if (!(mode < 0 ? enabled : mode) || p.M < min_rows) return PPT_EUNSUPPORTED;
        static const int long_k = env_int("PPT_GEMM256_LONG_K", 768);
        const int64_t tiles = (int64_t)((p.M + 255) / 256) * ((p.N + bn - 1) / bn) * (p.batch > 0 ? p.batch : 1);
        if (p.K < long_k && !(p.act == PPT_ACT_NONE && tiles >= 512)) return PPT_EUNSUPPORTED;
    }
    hipStream_t s = ppt_stream(stream);
    static const int stagger = env_int("PPT_GEMM256_PP", 1);         // the staggered schedule (0: one barrier per stage, v1)
    if (stagger) return p.dtype == PPT_BF16 ? launch256<bf16_t, true>(p, bn, s) : launch256<f16_t, true>(p, bn, s);
    return p.dtype == PPT_BF16 ? launch256<bf16_t, false>(p, bn, s) : launch256<f16_t, false>(p, bn, s);
}

extern "C" int ppt_gemm256(const ppt_gemm_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_gemm_params q = *pp;
    if (!q.wave_prio) q.wave_prio = ppt_get_wave_priority();
    if (q.M <= 0 || q.N <= 0 || q.K <= 0 || !q.B || !q.A) return PPT_EINVAL;
    if (q.K % 8 || q.lda % 8 || q.ldb % 8 || ((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15)) return PPT_EINVAL;
    if ((q.col_sum == nullptr) != (q.col_sqsum == nullptr)) return PPT_EINVAL;
    if (q.group_add && q.group_rows <= 0) return PPT_EINVAL;
    if (q.row_scale && q.row_scale_rows <= 0) return PPT_EINVAL;
    if (q.batch > 1 && (q.C2 || q.col_sum || q.pool_max || q.residual || q.residual2 || q.dact_pre || q.group_add)) return PPT_EUNSUPPORTED;
    return ppt_gemm256_dispatch(&q, 1, stream);
}
