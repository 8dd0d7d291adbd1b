// gemm.hip -- C[M,N] = epilogue( prologue(A)[M,K] . B[N,K]^T ) on the gfx950 matrix cores.
//
// One kernel family serves every nn.Linear / Conv1d(k=1) of the path (see include/ppt_hip.h):
//   * 128x128 (or, for grids that would under-fill the 256 CUs, 64x64) output tile per 256-thread
//     workgroup, 4 waves as 2x2, each wave (BM/2)x(BN/2) in MFMA tiles of 32x32
//     (v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 in parity mode), fp32 accumulators in registers;
//   * K is walked in 128-BYTE slabs (64 bf16 / 32 f32) so both dtypes share one LDS image:
//     [128 rows][8 x 16-B chunks], chunk index XOR-swizzled with (row>>1)&7, which makes the
//     ds_read_b128 fragment reads of the 32x32x16 operand conflict-free;
//   * register-staged pipeline, THREE slabs deep: while slab s is multiplied out of LDS, slabs s+1 and
//     s+2 sit in (or fly into) two register sets and the loads of slab s+3 are issued into the third;
//     slab s+1 moves to the other LDS buffer after the MFMAs -> one barrier per slab and ~2 slab-times
//     of latency cover for every global load (the K loop of these shapes is latency-, not
//     bandwidth-bound: M is huge or the grid is small, K is 128..2048).  Loads are branch-free
//     (clamped address + select), so the compiler's counted vmcnt keeps two slabs in flight.  Staging
//     through registers (rather than LDS-DMA) is what lets the A operand be TRANSFORMED on the way
//     in: BatchNorm+ReLU of the previous layer (A_AFFINE_RELU) or the whole K=3 first conv of the
//     mini-PointNet (A_CONV1) never touch HBM;
//   * epilogue on the accumulator registers: bias, per-group additive term, activation or its
//     derivative, DropPath row scale, up to two residual adds, a second output copy, BatchNorm
//     column statistics and the 32-row max-pool of the mini-PointNet (one MFMA row-tile == one
//     kNN group, so the pool is 15 v_max + one cross-half exchange).
#include "gemm_common.h"

namespace {


template <typename T, int A_MODE, int BM, int BN, bool SPLIT = false>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const ppt_gemm_params p)
{
    static_assert(!SPLIT || sizeof(T) == 4, "split16 is a mode of the fp32-operand kernel");
    const float sa = SPLIT ? pow2f(p.split_a_pow2) : 1.0f, sb = SPLIT ? pow2f(p.split_b_pow2) : 1.0f;
    uint32_t split_over = 0;                 // (SPLIT: did this thread saturate a value beyond half's range -- gemm_common.h)
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 32, TJ = WN / 32, NRA = BM / 32, NRB = BN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES), PARK_BYTES = 4 * WM * WN * 4;
    constexpr int MAIN_BYTES = STAGE_BYTES > PARK_BYTES ? STAGE_BYTES : PARK_BYTES;
    constexpr int TAB_BYTES = A_MODE == PPT_A_PLAIN ? 0 : (A_MODE == PPT_A_CONV1 ? 16 : 8) * PRO_TAB_K;
    __shared__ __align__(16) unsigned char smem[MAIN_BYTES + TAB_BYTES];   // A0 A1 B0 B1 | prologue table
    const float *tab = reinterpret_cast<const float *>(smem + MAIN_BYTES);
    if constexpr (A_MODE != PPT_A_PLAIN) {
        fill_prologue_table<A_MODE>(p, reinterpret_cast<float *>(smem + MAIN_BYTES));
        __syncthreads();
    }
    constexpr int BK = ROWB / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
#ifndef PPT_DBG_NO_XCD_SWIZZLE
    // consecutive tiles (same m-tile, n fastest) onto one XCD: blocks are dealt round-robin over the 8
    // XCDs, each with a private L2, so without this remap the column tiles that share one A row
    // panel land on 8 different L2s (measured +5..8 % on the block GEMMs; speed only, never correctness)
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    const int n0 = (lin % gridDim.x) * BN, m0 = (lin / gridDim.x) * BM;
#else
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
#endif
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)blockIdx.z * p.strideA;
    const T *B = reinterpret_cast<const T *>(p.B) + (int64_t)blockIdx.z * p.strideB;

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    EpiPre<TI, TJ> epre;        // (the CONV1 prologue already fills the register file: it fetches these after the K loop)
    if constexpr (A_MODE != PPT_A_CONV1) epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);

    const int nslab = (p.K + BK - 1) / BK;
    Stage<NRA> a0, a1, a2;
    Stage<NRB> b0, b1, b2;
    unsigned char *const Abuf = smem, *const Bbuf = smem + 2 * A_BYTES;

#ifdef PPT_DBG_SKIP_LDS_WRITE
#define PPT_DBG_WRITE(X) do { if (p.M < 0) { X; } } while (0)
#else
#define PPT_DBG_WRITE(X) X
#endif
#define PPT_LOAD(SA, SB, S)                                                   \
    do {                                                                      \
        load_A<T, A_MODE, NRA>(SA, p, A, m0, (S) * BK);                       \
        load_plain<T, NRB>(SB, B, p.ldb, p.N, p.K, n0, (S) * BK);             \
    } while (0)
#define PPT_WRITE(SA, SB, S, BUF)                                             \
    do {                                                                      \
        finish_A<T, A_MODE, NRA>(SA, p, m0, (S) * BK, tab);                   \
        mask_plain<T, NRB>(SB, p.N, p.K, n0, (S) * BK);                       \
        if constexpr (SPLIT) {                                                \
            write_stage_split<NRA, BM>(SA, Abuf + ((BUF) & 1) * A_BYTES, sa, split_over); \
            write_stage_split<NRB, BN>(SB, Bbuf + ((BUF) & 1) * B_BYTES, sb, split_over); \
        } else {                                                              \
            PPT_DBG_WRITE(write_stage<NRA>(SA, Abuf + ((BUF) & 1) * A_BYTES)); \
            PPT_DBG_WRITE(write_stage<NRB>(SB, Bbuf + ((BUF) & 1) * B_BYTES)); \
        }                                                                     \
    } while (0)
    // STEP(s): multiply slab s; FREE set (held slab s) receives slab s+3; NEXT set (slab s+1) moves to LDS.
    // Loads and LDS writes are UNCONDITIONAL (slab indices clamped to the last slab; the surplus write
    // lands in the buffer nobody reads again): every path then has the same number of loads in flight
    // at every wait, which is what lets the compiler emit counted vmcnt(16) instead of draining to 0.
#define PPT_STEP(S, FA, FB, NA, NB)                                                                       \
    {                                                                                                     \
        PPT_LOAD(FA, FB, min((S) + 3, last));                                                             \
        if constexpr (SPLIT) mma_slab_split<TI, TJ, BM, BN>(Abuf + ((S) & 1) * A_BYTES, Bbuf + ((S) & 1) * B_BYTES, wm * WM, wn * WN, lane, acc); \
        else mma_slab<T, TI, TJ>(Abuf + ((S) & 1) * A_BYTES, Bbuf + ((S) & 1) * B_BYTES, wm * WM, wn * WN, lane, acc); \
        PPT_WRITE(NA, NB, min((S) + 1, last), (S) + 1);                                                   \
        __syncthreads();                                                                                  \
    }

    const int last = nslab - 1;
    PPT_LOAD(a0, b0, 0);
    PPT_LOAD(a1, b1, min(1, last));
    PPT_LOAD(a2, b2, min(2, last));
    PPT_WRITE(a0, b0, 0, 0);
    __syncthreads();
#ifdef PPT_DBG_SKIP_LOADS
#undef PPT_LOAD
#define PPT_LOAD(SA, SB, S) do { } while (0)
#endif
    for (int s = 0;; s += 3) {
        PPT_STEP(s, a0, b0, a1, b1)
        if (s + 1 >= nslab) break;
        PPT_STEP(s + 1, a1, b1, a2, b2)
        if (s + 2 >= nslab) break;
        PPT_STEP(s + 2, a2, b2, a0, b0)
        if (s + 3 >= nslab) break;
    }
#undef PPT_STEP
#undef PPT_WRITE
#undef PPT_LOAD
    if constexpr (SPLIT) {
        scale_acc<TI, TJ>(acc, pow2f(-(p.split_a_pow2 + p.split_b_pow2)));
        split_report(split_over, p.split_overflow);
    }

    // ---------------- epilogue ----------------
    // The operand tiles are dead (the loop ends on a barrier): every wave parks its fp32 accumulators in
    // its own slice of LDS, row-major, and walks them in 16-byte pieces.  That turns the MFMA C layout
    // (column on the lane, rows scattered over 16 registers) into full-row global stores, makes the
    // BatchNorm column sums and the 32-row max-pool per-lane running values, and keeps the flag-driven
    // epilogue body out of the unroller.
    const int64_t zc = (int64_t)blockIdx.z * p.strideC;
#ifdef PPT_DBG_SKIP_EPILOGUE
    if (acc[0][0][0] != 12345.678f) return;
#endif
#ifndef PPT_DBG_NO_REG_EPILOGUE
    if (reg_epilogue_ok<TI>(p, zc)) {
        if constexpr (A_MODE == PPT_A_CONV1) epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);
        epilogue_regs<TI, TJ>(p, acc, epre, smem + w * (WM * WN * 2), lane, m0 + wm * WM, n0 + wn * WN, zc);
        return;
    }
#endif
    float *ct = reinterpret_cast<float *>(smem) + w * (WM * WN);
    {
        const int h = lane >> 5, cl = lane & 31;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * WN + j * 32 + cl] = acc[i][j][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if (vec_epilogue_ok(p, zc)) epilogue_vec8<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, zc);
    else epilogue_scalar<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, m0, wm, zc);
}

// =================================================================================================
// LDS-DMA variant for plain operands (no A prologue): global_load_lds_dwordx4 writes the slab straight
// into LDS, so the ~13-cycle-per-instruction ds_write_b128 path that register staging needs -- measured
// as 35-45 % of the K loop of the register-staged kernel (tools/gemm_tune.py, noloads_noepi vs
// noloads_noepi_nowrite) -- disappears, and no VGPRs are spent on staging.  Three LDS stages; slab s+2 is
// issued right behind the barrier that publishes slab s (its stage was last read two slabs ago); each
// wave waits for its own copies with a COUNTED vmcnt (the youngest slab stays in flight) before a raw
// s_barrier.  The LDS image is the same XOR-swizzled one: the DMA writes 64 lanes x 16 B linearly, so the
// swizzle is applied to each lane's SOURCE chunk (same involution as the fragment reads).
// Requires K % (128 / sizeof(T)) == 0; rows beyond M / N read a clamped (valid) row whose products are
// never stored.
// =================================================================================================
template <typename T, int ROWS>
__device__ __forceinline__ void glds_slab(const T *base, int64_t ld, int rows, int r0, int k0, unsigned char *tile, int w, int lane)
{
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PER_WAVE = ROWS / 32;                  // 1 KiB pieces (8 rows) per wave
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int rg = (w * PER_WAVE + i) * 8;           // first tile row of this piece
        const int r = rg + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);       // source chunk that belongs in LDS slot lane&7 of row r
        const T *src = base + (int64_t)min(r0 + r, rows - 1) * ld + k0 + c * EPC;
        lds_dma16(src, tile + rg * ROWB);
    }
}

// Diagnostic build only (tools/gemm_stamp.py compiles this file with -DPPT_GEMM_STAMP into its own library): lane 0 of every
// wave of the 64x64 LDS-DMA kernel stores s_memtime at entry / loads issued / first slab readable / K loop done / stores done
// into the buffer p.pool_min points at (unused by these launches).
#ifdef PPT_GEMM_STAMP
#define GEMM_STAMP(slot) do { if (lane == 0 && p.pool_min && !p.pool_max) reinterpret_cast<unsigned long long *>(p.pool_min)[((size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEMM_STAMP(slot) do { } while (0)
#endif
// NSTAGE: LDS stages of the ring = slabs in flight + 1.  Three (48 KiB: three workgroups of the 64 x 64 tile per CU).  FOUR
// (64 KiB, three slabs in flight) was measured for the prompt chain's small grids (<= 256 workgroups: one per CU whatever the
// LDS size) after the LDS-DMA ring actually ran as a ring (ppt_common.h lds_dma16): 6.7 -> 6.9 us at K = 512, 13.0 -> 13.4 us at
// K = 2048, C2 3.05 -> 3.10 ms -- the K loop is not short of bytes in flight, it runs at what ONE CU's LDS-DMA path takes
// (~27 B/clk).  PPT_GEMM_DEEP_BELOW=<workgroups> selects it (default 0: never).  (Also measured and removed: four HELPER waves per
// workgroup that only issue half of the DMA pieces and meet the barriers -- 12.6 -> 12.0 us at K = 2048 alone, C2 3.11 -> 3.19 ms in
// the step: the rate is a property of the CU, not of how many waves ask, and 512-thread workgroups are harder to place beside the
// tower.  And 32 x 64 tiles -- 208 two-wave workgroups, five 12 KiB stages, bit-identical results -- to put these 104-workgroup
// launches on more CUs: 12.2 -> 12.8 us plain, 12.6 -> 17.1 us with the residual epilogue, C2 3.08 -> 3.35 ms.  Two waves do not
// keep a CU's DMA path as busy as four.  The 64 x 64 / four-wave / three-stage kernel is where these shapes stay.)
template <typename T, int BM, int BN, int NSTAGE = 3>
__global__ __launch_bounds__(NT, BM == 64 ? 3 : 1) void gemm_kernel_glds(const ppt_gemm_params p)   // (a waves-per-SIMD floor keeps the accumulators out of AGPRs)
{
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 32, TJ = WN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int LOADS_PER_SLAB = BM / 32 + BN / 32;    // LDS-DMA instructions per wave per slab
    constexpr int PARK_BYTES = 4 * WM * WN * 4;
    __shared__ __align__(16) unsigned char smem[NSTAGE * STAGE > PARK_BYTES ? NSTAGE * STAGE : PARK_BYTES];
    constexpr int BK = ROWB / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    GEMM_STAMP(0);
    PPT_PRIO(p.wave_prio);
    const int wm = w >> 1, wn = w & 1;
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    const int n0 = (lin % gridDim.x) * BN, m0 = (lin / gridDim.x) * BM;
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
        glds_slab<T, BM>(A, p.lda, p.M, m0, slab * BK, smem + stage * STAGE, w, lane);
        glds_slab<T, BN>(B, p.ldb, p.N, n0, slab * BK, smem + stage * STAGE + A_BYTES, w, lane);
    };
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) issue(min(i, last), i);
    EpiPre<TI, TJ> epre;                                  // (behind the first slabs, as in gemm_kernel_glds_h)
    epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);
    GEMM_STAMP(1);
    int stage = 0;
    for (int s = 0; s < nslab; ++s) {
        // own copies of slab s have landed (the LOADS_PER_SLAB * (NSTAGE - 2) youngest, slabs s+1 .., may still fly) ...
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS_PER_SLAB * (NSTAGE - 2)) : "memory");
        __builtin_amdgcn_s_barrier();                     // ... and so have everybody else's: slab s is readable
        if (s == 0) GEMM_STAMP(2);
        int nstage = stage + NSTAGE - 1; if (nstage >= NSTAGE) nstage -= NSTAGE;
        issue(min(s + NSTAGE - 1, last), nstage);         // that stage was last read at slab s-1, before this barrier
        mma_slab<T, TI, TJ>(smem + stage * STAGE, smem + stage * STAGE + A_BYTES, wm * WM, wn * WN, lane, acc);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // surplus prefetches must not land on the parked accumulators
    __builtin_amdgcn_s_barrier();
    GEMM_STAMP(3);

    const int64_t zc = (int64_t)blockIdx.z * p.strideC;
#ifndef PPT_DBG_NO_REG_EPILOGUE
    if (reg_epilogue_ok<TI>(p, zc)) {
        epilogue_regs<TI, TJ>(p, acc, epre, smem + w * (WM * WN * 2), lane, m0 + wm * WM, n0 + wn * WN, zc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_STAMP(4);
        return;
    }
#endif
    float *ct = reinterpret_cast<float *>(smem) + w * (WM * WN);
    {
        const int h = lane >> 5, cl = lane & 31;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * WN + j * 32 + cl] = acc[i][j][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vec_epilogue_ok(p, zc)) epilogue_vec8<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, zc);
    else epilogue_scalar<WM, WN>(p, ct, lane, m0 + wm * WM, n0 + wn * WN, m0, wm, zc);
#ifdef PPT_GEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_STAMP(4);
#endif
}

// (Tried and removed: a split-K form of this kernel for the prompt chain's skinny linears -- 817 rows, 104 ... 416 tiles --
// with blockIdx.z cutting K, fp32 partial tiles in a caller-lent workspace and the last workgroup to arrive at the tile's
// counter summing them in split order.  Correct and repeatable, never faster: with release / acquire fences every workgroup
// writes back and invalidates its XCD's whole L2 (52 ... 130 us per launch); with fence-free write-through stores and
// L2-bypassing loads (relaxed agent-scope atomics) the three dependent memory round trips -- partial out, counter, partials
// in -- cost the 4 ... 6 us the shorter K loop saves: 12.5 -> 12.5 us at two splits of K = 2048, slower beyond.  In-kernel
// stamps of the unsplit kernel (tools/gemm_stamp.py): ~1.2 us from entry to the first loads issued, 585 clocks per 16 KiB
// slab = 28 B/clk per CU, the L2 -> LDS rate of one CU.)
// =================================================================================================
// Half-slab LDS-DMA kernel for the big plain-operand problems (qkv, fc1, conv3): 128x128 tiles, 64-byte K slabs
// (32 bf16), 16 KiB stages: two of them (default; FOUR workgroups = 16 waves share a CU) or three (three workgroups).
// Why: tools/lds_fill_bench.hip shows that what a CU can pull out of L2 depends on how many waves are issuing --
// 4 waves 6.5, 8 waves 11, 12 waves 14, 16 waves 16 TB/s over the chip -- and hardly on the bytes each keeps in
// flight, and both existing tile loops sit on that line: 64x64 tiles (12 waves, 32 flop per LDS-fill byte) at
// 11.4 of 14 TB/s, register-staged 128x128 (8 waves, 64 flop/B) at 10 of 11 TB/s.  This kernel takes the 64 flop/B
// of the big tile AND the 12 waves.  It has only the register-layout epilogue (the fp32 park of epilogue_vec8 would
// need 64 KiB): the host sends it launches for which reg_epilogue_ok holds.
// LDS image: 64-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3): four consecutive rows fill one
// 256-byte bank row, and the lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) then touch 16 distinct slots.
// =================================================================================================
constexpr int ROWH = 64;
__device__ __forceinline__ int lds_off_h(int row, int chunk) { return row * ROWH + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <typename T, int ROWS>
__device__ __forceinline__ void glds_half(const T *base, int64_t ld, int rows, int r0, int k0, unsigned char *tile, int w, int lane)
{
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PER_WAVE = ROWS / 64;                  // 1 KiB pieces (16 rows) per wave
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
__device__ __forceinline__ void mma_half(const unsigned char *As, const unsigned char *Bs, int arow0, int brow0, int lane,
                                         f32x16_t (&acc)[TI][TJ])
{
    const int r = lane & 31, h = lane >> 5;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                a[i] = *reinterpret_cast<const uint4 *>(As + lds_off_h(arow0 + i * 32 + r, kk * 2 + h));
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                b[j] = *reinterpret_cast<const uint4 *>(Bs + lds_off_h(brow0 + j * 32 + r, kk * 2 + h));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = h16<T>::mfma32(a[i], b[j], acc[i][j]);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int kq = 2 * kk + h;      // k index (in floats) inside the 16-float slab
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                a[i] = *reinterpret_cast<const float *>(As + lds_off_h(arow0 + i * 32 + r, kq >> 2) + (kq & 3) * 4);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                b[j] = *reinterpret_cast<const float *>(Bs + lds_off_h(brow0 + j * 32 + r, kq >> 2) + (kq & 3) * 4);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

template <typename T, int BM, int BN, int NSTAGE = 3, int EPI = -1>
__global__ __launch_bounds__(NT, NSTAGE == 3 ? 3 : 4) void gemm_kernel_glds_h(const ppt_gemm_params p)
{
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 32, TJ = WN / 32;
    constexpr int A_BYTES = BM * ROWH, B_BYTES = BN * ROWH, STAGE = A_BYTES + B_BYTES;
    constexpr int LOADS_PER_SLAB = BM / 64 + BN / 64;    // LDS-DMA instructions per wave per slab
    static_assert(LOADS_PER_SLAB == 4, "counted vmcnt below");
    constexpr int PARK_BYTES = 4 * WM * WN * 2;          // bf16 park of epilogue_regs
    __shared__ __align__(16) unsigned char smem[NSTAGE * STAGE > PARK_BYTES ? NSTAGE * STAGE : PARK_BYTES];
    constexpr int BK = ROWH / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int nwg = gridDim.x * gridDim.y;
    const int lin0 = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = lin0 % 8;                   // same XCD-aware remap as gemm_kernel
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin0 / 8;
    const int n0 = (lin % gridDim.x) * BN, m0 = (lin / gridDim.x) * BM;
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
        glds_half<T, BM>(A, p.lda, p.M, m0, slab * BK, smem + stage * STAGE, w, lane);
        glds_half<T, BN>(B, p.ldb, p.N, n0, slab * BK, smem + stage * STAGE + A_BYTES, w, lane);
    };
    issue(0, 0);
    if constexpr (NSTAGE == 3) issue(min(1, last), 1);
    EpiPre<TI, TJ> epre;                                  // (behind the first slabs: the K loop must not start later for it)
    epilogue_prefetch<TI, TJ>(p, epre, lane, m0 + wm * WM, n0 + wn * WN);
    int stage = 0;
    for (int s = 0; s < nslab; ++s) {
        if constexpr (NSTAGE == 3) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // own copies of slab s landed (slab s+1's four may still fly)
            __builtin_amdgcn_s_barrier();                     // ... and everybody else's: slab s is readable
            int nstage = stage + 2; if (nstage >= NSTAGE) nstage -= NSTAGE;
            issue(min(s + 2, last), nstage);                  // stage (s+2)%3 was last read at slab s-1, before this barrier
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(min(s + 1, last), stage ^ 1);               // the other stage was last read at slab s-1, before this barrier
        }
        mma_half<T, TI, TJ>(smem + stage * STAGE, smem + stage * STAGE + A_BYTES, wm * WM, wn * WN, lane, acc);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // surplus prefetches must not land on the parked tile
    __builtin_amdgcn_s_barrier();
    epilogue_regs<TI, TJ, EPI, EPI < 0 ? -1 : (sizeof(T) == 2 ? 1 : 0)>(p, acc, epre, smem + w * (WM * WN * 2), lane, m0 + wm * WM,
                                                                        n0 + wn * WN, (int64_t)blockIdx.z * p.strideC);
}

// host side of the choice: plain operands, K in whole half-slabs, an epilogue the register path covers for every
// batch slice, and enough 128x128 tiles to give each CU its three workgroups
template <typename T>
bool glds_h_ok(const ppt_gemm_params &p, int64_t tiles128)
{
    static const int min_tiles = [] { const char *e = getenv("PPT_GEMM_H128_MIN"); return e ? atoi(e) : 768; }();
    constexpr int BKH = ROWH / sizeof(T);
    if (p.a_mode != PPT_A_PLAIN || (p.K % BKH) != 0 || tiles128 < min_tiles) return false;
    if (p.batch > 1 && (p.strideC % 8) != 0) return false;
    return reg_epilogue_ok<2>(p, 0);
}

template <typename T, int BM, int BN, bool SPLIT>
int launch_gemm_tile(const ppt_gemm_params &p, hipStream_t s)
{
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.batch > 0 ? p.batch : 1);
    if (grid.y > 65535 || grid.z > 65535) return PPT_EUNSUPPORTED;
    static const int use_glds = [] { const char *e = getenv("PPT_GEMM_GLDS"); return e ? atoi(e) : 1; }();
    constexpr int BKE = ROWB / sizeof(T);
    // (three 32 KiB stages of the 128x128 tile leave one workgroup per CU: measured slower than register staging)
    if (!SPLIT && use_glds && BM == 64 && p.a_mode == PPT_A_PLAIN && p.K % BKE == 0) {   // (split16 splits between registers and LDS)
        static const int deep_below = [] { const char *e = getenv("PPT_GEMM_DEEP_BELOW"); return e ? atoi(e) : 0; }();
        if constexpr (BM == 64 && BN == 64) {
            if ((int)(grid.x * grid.y * grid.z) < deep_below && p.K / BKE >= 4) {
                hipLaunchKernelGGL((gemm_kernel_glds<T, 64, 64, 4>), grid, dim3(NT), 0, s, p);
                PPT_CHECK_LAUNCH();
                return PPT_OK;
            }
        }
        hipLaunchKernelGGL((gemm_kernel_glds<T, BM, BN>), grid, dim3(NT), 0, s, p);
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    switch (p.a_mode) {
    case PPT_A_PLAIN: hipLaunchKernelGGL((gemm_kernel<T, PPT_A_PLAIN, BM, BN, SPLIT>), grid, dim3(NT), 0, s, p); break;
    case PPT_A_AFFINE_RELU: hipLaunchKernelGGL((gemm_kernel<T, PPT_A_AFFINE_RELU, BM, BN, SPLIT>), grid, dim3(NT), 0, s, p); break;
    case PPT_A_CONV1: hipLaunchKernelGGL((gemm_kernel<T, PPT_A_CONV1, BM, BN, SPLIT>), grid, dim3(NT), 0, s, p); break;
    default: return PPT_EINVAL;
    }
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// tile choice (measured in one process on one MI355X, tools/gemm_shapes.py + bench.py with
// PPT_GEMM_SMALL_BELOW = 0 / 512 / 1e9 -> 8.14 / 7.50 / 7.08 ms per C2 step): 128x128 tiles read one LDS
// fragment per MFMA but keep only 2 workgroups (8 waves) on a CU; 64x64 tiles read two but run 4-5
// workgroups per CU, and for these short-K problems (K = 128..2048, every tile pays a load prologue and a
// store epilogue) the extra latency hiding wins up to several thousand tiles.  Only the half-gigabyte
// mini-PointNet GEMMs stay on 128x128 (and those that emit 64-row BatchNorm partials / 32-row pools).
template <typename T, bool SPLIT = false>
int launch_gemm(const ppt_gemm_params &p, hipStream_t s)
{
    static const int small_below = [] { const char *e = getenv("PPT_GEMM_SMALL_BELOW"); return e ? atoi(e) : 4096; }();
    const int64_t tiles128 = (int64_t)((p.N + 127) / 128) * ((p.M + 127) / 128) * (p.batch > 0 ? p.batch : 1);
    const bool need128 = p.col_sum || p.pool_max;       // 32-row chunk partials / pools need 64-wide wave tiles
    if constexpr (SPLIT) {
        // split16: the register-staged kernel only (the split sits between its registers and LDS); 128 x 128 tiles (24 MFMAs per
        // 32 values a thread splits) as soon as they fill the chip, 64 x 64 (6 per 16) below
        static const int split_small_below = [] { const char *e = getenv("PPT_SPLIT16_SMALL_BELOW"); return e ? atoi(e) : 512; }();
        // (measured and not kept: 128 x 64 tiles for narrow N over many rows -- fc2 of a C2 batch 115.7 vs 116.6 us)
        if (!need128 && tiles128 < split_small_below) return launch_gemm_tile<T, 64, 64, true>(p, s);
        return launch_gemm_tile<T, 128, 128, true>(p, s);
    }
    if (glds_h_ok<T>(p, tiles128)) {
        dim3 grid((p.N + 127) / 128, (p.M + 127) / 128, p.batch > 0 ? p.batch : 1);
        if (grid.y > 65535 || grid.z > 65535) return PPT_EUNSUPPORTED;
        // two stages (32 KiB, 4 workgroups = 16 waves per CU, prefetch distance 1) beat three (48 KiB, 3 workgroups,
        // distance 2) in the step: 4.73 vs 4.82 ms on C2, conv3 346 -> 325 us -- the fill rate follows the wave count
        // (tools/lds_fill_bench.hip) and a fourth co-resident workgroup hides more of the others' epilogues
        static const int two_stage = [] { const char *e = getenv("PPT_GEMM_H128_STAGES"); return !(e && atoi(e) == 3); }();
        if (two_stage) {
            switch (p.C ? epi_mask(p) : -1) {           // the three hot epilogues have their own small kernels
            case 0: hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2, 0>), grid, dim3(NT), 0, s, p); break;
            case EPI_GELU: hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2, EPI_GELU>), grid, dim3(NT), 0, s, p); break;
            case EPI_GROUP | EPI_STATS:
                hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2, EPI_GROUP | EPI_STATS>), grid, dim3(NT), 0, s, p); break;
            default: hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128, 2>), grid, dim3(NT), 0, s, p); break;
            }
        } else {
            hipLaunchKernelGGL((gemm_kernel_glds_h<T, 128, 128>), grid, dim3(NT), 0, s, p);
        }
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    if (!need128 && tiles128 < small_below) return launch_gemm_tile<T, 64, 64, false>(p, s);
    return launch_gemm_tile<T, 128, 128, false>(p, s);
}

}  // namespace

extern "C" int ppt_gemm256_dispatch(const ppt_gemm_params *pp, int force, void *stream);      // gemm256.hip
extern "C" int ppt_gemm256_split_dispatch(const ppt_gemm_params *pp, void *stream);

extern "C" int ppt_gemm(const ppt_gemm_params *pp, void *stream)
{
    if (!pp) return PPT_EINVAL;
    ppt_gemm_params q = *pp;
    if (!q.wave_prio) q.wave_prio = ppt_get_wave_priority();
    const ppt_gemm_params &p = q;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.B) return PPT_EINVAL;
    if (p.dtype != PPT_F32 && p.dtype != PPT_BF16 && p.dtype != PPT_F16) return PPT_EINVAL;
    if (p.split16 && (p.dtype != PPT_F32 || abs(p.split_a_pow2) > 24 || abs(p.split_b_pow2) > 24)) return PPT_EINVAL;
    // one 16-bit format per GEMM: outputs / saved pre-activations / pooled rows are fp32 or the operands' format
    if (p.dtype != PPT_F32) {
        if ((p.C && p.c_dtype != PPT_F32 && p.c_dtype != p.dtype) || (p.C2 && p.c2_dtype != PPT_F32 && p.c2_dtype != p.dtype) ||
            (p.pool_max && p.pool_dtype != PPT_F32 && p.pool_dtype != p.dtype))
            return PPT_EINVAL;
    }
    const int epc = p.dtype != PPT_F32 ? 8 : 4;
    if (p.K % epc || p.ldb % epc || ((uintptr_t)p.B & 15)) return PPT_EINVAL;
    if (p.a_mode != PPT_A_PLAIN && p.K > 1024) return PPT_EUNSUPPORTED;      // prologue constants live in an LDS table
    if (p.a_mode == PPT_A_CONV1) {
        if (!p.pts || !p.w1 || !p.b1) return PPT_EINVAL;
    } else {
        if (!p.A || p.lda % epc || ((uintptr_t)p.A & 15)) return PPT_EINVAL;
        if (p.a_mode == PPT_A_AFFINE_RELU && (!p.a_scale || !p.a_shift)) return PPT_EINVAL;
    }
    if ((p.col_sum == nullptr) != (p.col_sqsum == nullptr)) return PPT_EINVAL;
    if (p.pool_max && p.pool_rows != 0 && p.pool_rows != 16 && p.pool_rows != 32 && p.pool_rows != 64) return PPT_EINVAL;
#ifndef PPT_GEMM_STAMP
    if (p.pool_min && !p.pool_max) return PPT_EINVAL;
#endif
    if ((p.col_sum || p.pool_max) && ((p.N % 8) || ((uintptr_t)p.pool_min & 15) || ((uintptr_t)p.col_sum & 15) || ((uintptr_t)p.col_sqsum & 15) || ((uintptr_t)p.pool_max & 15)))
        return PPT_EUNSUPPORTED;                       // statistics / pooling exist only in the 16-byte epilogue
    if (p.group_add && p.group_rows <= 0) return PPT_EINVAL;
    if (p.row_scale && p.row_scale_rows <= 0) return PPT_EINVAL;
    if (p.batch > 1 && (p.C2 || p.col_sum || p.pool_max || p.residual || p.residual2 || p.dact_pre || p.group_add))
        return PPT_EUNSUPPORTED;
    if (ppt_gemm256_dispatch(&p, 0, stream) == PPT_OK) return PPT_OK;     // the 256-row macro-tile core takes the big plain problems
    hipStream_t s = ppt_stream(stream);
    if (p.dtype == PPT_F32 && p.split16) {                                        // hi + lo half products (gemm_common.h, split16)
        if (ppt_gemm256_split_dispatch(&p, stream) == PPT_OK) return PPT_OK;     // large plain problems: the 256 x 128 tile
        return launch_gemm<float, true>(p, s);
    }
    return p.dtype == PPT_BF16 ? launch_gemm<bf16_t>(p, s) : p.dtype == PPT_F16 ? launch_gemm<f16_t>(p, s) : launch_gemm<float>(p, s);
}
