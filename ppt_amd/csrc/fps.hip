// fps.hip -- H1: farthest point sampling for gfx950.
//
// Replaces the Python loop of models/pointbert/misc.py:44-69 (5 ATen launches x M iterations).
// One workgroup per cloud.  Every thread keeps PPT points (x,y,z) AND their running minimum
// distance in VGPRs for the whole kernel, so the cloud is read from HBM exactly once
// (12*N bytes) and the M dependent steps touch no memory but 32 B of LDS per wave:
//
//   per step:  PPT distance updates / thread            (VALU, ((dx*dx+dy*dy)+dz*dz), no FMA)
//              wave arg-max via two 6-stage DPP reductions (max distance bits, then min index)
//              owner lane publishes {dist, idx, x, y, z} to an LDS slot  (parity double-buffered)
//              ONE s_barrier; every thread folds the W slots redundantly.
//
// Tie rule = torch.max: the FIRST maximal index.  Distances are >= +0, so their IEEE bit patterns
// are monotone as unsigned integers and the arg-max is an integer max followed by an integer min
// over the tied lanes' indices.
#include <cstdlib>

#include "ppt_common.h"

namespace {

struct __align__(16) fps_slot {
    uint32_t dist_bits;
    uint32_t idx;
    float x, y;
    float z;
    uint32_t pad[3];
};

// XYZ_LDS: the cloud is also kept in LDS (12 N bytes, N <= 8192) and the loop tracks only (distance, index) per point;
// the winner's coordinates are one broadcast LDS read per step instead of three selects per point plus three
// v_readlane per wave (the per-point work is what bounds the big clouds: 8192 points, 16 waves: 1.94 -> ~1.3 us per pick).
template <int PPT, int W, bool XYZ_LDS>
__global__ __launch_bounds__(W * 64) void fps_kernel(const float *__restrict__ xyz, int N, int M,
                                                     const int64_t *__restrict__ start,
                                                     int64_t *__restrict__ out_idx,
                                                     float *__restrict__ out_xyz)
{
    constexpr int T = W * 64;
    extern __shared__ __align__(16) unsigned char smem[];
    fps_slot *slots = reinterpret_cast<fps_slot *>(smem);                 // [2][W]
    int32_t *idx_list = reinterpret_cast<int32_t *>(smem + sizeof(fps_slot) * 2 * W);
    float *xyz_list = reinterpret_cast<float *>(idx_list + M);           // [M][3]
    float *cloud = xyz_list + (size_t)M * 3;                             // [N][3] when XYZ_LDS

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int b = blockIdx.x;
    const float *p = xyz + (size_t)b * N * 3;

    float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = t + j * T;
        if (i < N) {
            px[j] = p[i * 3 + 0]; py[j] = p[i * 3 + 1]; pz[j] = p[i * 3 + 2];
            dist[j] = 1e10f;                      // misc.py:57
        } else {
            px[j] = py[j] = pz[j] = 0.0f;
            dist[j] = 0.0f;                       // padding: never beats a real point (index >= N loses ties)
        }
    }

    if constexpr (XYZ_LDS) {
        for (int i = t; i < N * 3; i += T) cloud[i] = p[i];
        __syncthreads();
    }
    int cur = (int)start[b];
    float cx = p[cur * 3 + 0], cy = p[cur * 3 + 1], cz = p[cur * 3 + 2];

    for (int it = 0; it < M; ++it) {
        if (t == 0) {
            idx_list[it] = cur;
            xyz_list[it * 3 + 0] = cx; xyz_list[it * 3 + 1] = cy; xyz_list[it * 3 + 2] = cz;
        }
        float best = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
        uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float dx = __fsub_rn(px[j], cx), dy = __fsub_rn(py[j], cy), dz = __fsub_rn(pz[j], cz);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            const float nd = fminf(dist[j], d);   // misc.py:66
            dist[j] = nd;
            if constexpr (XYZ_LDS) {
                const bool up = nd > best;            // strict: lowest j (= lowest index of this thread) wins ties
                best = up ? nd : best; bi = up ? (uint32_t)(t + j * T) : bi;
            } else if (nd > best) {
                best = nd; bi = (uint32_t)(t + j * T); bx = px[j]; by = py[j]; bz = pz[j];
            }
        }
        const uint32_t bits = __float_as_uint(best);
        // (Removed in round 6, measured in round 4, profiles/r04_fps_variants.md: ONE reduction over the 64-bit key (distance bits
        // << 32) | ~index instead of the 32-bit max + ballot -- bit-exact, 567 ns per pick against 439; and one wave per cloud
        // with 16 points per lane -- 579 ns.)
        const uint32_t wmax = wave_reduce_umax(bits);
        // the lane that holds the wave's arg-max: with a single maximal lane (the common case) a ballot names it; equal
        // distances in several lanes (duplicated points) go through the index reduction, lowest index wins
        uint64_t cand = __ballot(bits == wmax);
        if (__popcll(cand) != 1) {                // wave-uniform
            const uint32_t wlow = wave_reduce_umin(bits == wmax ? bi : 0xFFFFFFFFu);
            cand = __ballot(bits == wmax && bi == wlow);
        }
        const int wl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)cand) - 1);
        const uint32_t widx = (uint32_t)__builtin_amdgcn_readlane((int)bi, wl);
        fps_slot *sl = slots + (it & 1) * W;
        if constexpr (XYZ_LDS) {
            if (lane == 0) { sl[w].dist_bits = wmax; sl[w].idx = widx; }
            __syncthreads();
            uint2 sd[W];
#pragma unroll
            for (int k = 0; k < W; ++k) sd[k] = *reinterpret_cast<const uint2 *>(&sl[k]);
            uint32_t gd = sd[0].x, gi = sd[0].y;
#pragma unroll
            for (int k = 1; k < W; ++k) {
                const bool take = (sd[k].x > gd) | ((sd[k].x == gd) & (sd[k].y < gi));
                gd = take ? sd[k].x : gd; gi = take ? sd[k].y : gi;
            }
            cur = (int)gi;
            cx = cloud[cur * 3 + 0]; cy = cloud[cur * 3 + 1]; cz = cloud[cur * 3 + 2];
            continue;
        }
        const float wx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(bx), wl));
        const float wy = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(by), wl));
        const float wz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(bz), wl));
        if (lane == 0) {                          // the wave's winner sits in SGPRs now: one lane publishes it
            sl[w].dist_bits = wmax; sl[w].idx = widx; sl[w].x = wx; sl[w].y = wy; sl[w].z = wz;
        }
        __syncthreads();
        // every thread folds the W slots: ALL slots are fetched first (independent 16-byte reads, one wait) and folded
        // with selects -- written as a chain of compare-and-reload the compiler emitted W dependent LDS round trips with
        // a branch each, which was most of the step
        uint4 sa[W];
        float sz[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            sa[k] = *reinterpret_cast<const uint4 *>(&sl[k]);
            sz[k] = sl[k].z;
        }
        uint32_t gd = sa[0].x, gi = sa[0].y;
        float gx = __uint_as_float(sa[0].z), gy = __uint_as_float(sa[0].w), gz = sz[0];
#pragma unroll
        for (int k = 1; k < W; ++k) {
            const bool take = (sa[k].x > gd) | ((sa[k].x == gd) & (sa[k].y < gi));      // (no short circuit: no branch)
            gd = take ? sa[k].x : gd; gi = take ? sa[k].y : gi;
            gx = take ? __uint_as_float(sa[k].z) : gx; gy = take ? __uint_as_float(sa[k].w) : gy; gz = take ? sz[k] : gz;
        }
        cur = (int)gi; cx = gx; cy = gy; cz = gz;
    }
    __syncthreads();
    for (int i = t; i < M; i += T) out_idx[(size_t)b * M + i] = (int64_t)idx_list[i];
    if (out_xyz)
        for (int i = t; i < M * 3; i += T) out_xyz[(size_t)b * M * 3 + i] = xyz_list[i];
}

template <int PPT, int W, bool XYZ_LDS>
int launch_fps_v(const float *xyz, int B, int N, int M, const int64_t *start, int64_t *out_idx,
                 float *out_xyz, hipStream_t s, size_t lds)
{
    // opt-in to > 64 KB of dynamic LDS: once per kernel instance (thread-safe static initialisation), to the device limit
    static const hipError_t optin = hipFuncSetAttribute((const void *)fps_kernel<PPT, W, XYZ_LDS>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (lds > 64 * 1024 && optin != hipSuccess) return PPT_ELAUNCH;
    hipLaunchKernelGGL((fps_kernel<PPT, W, XYZ_LDS>), dim3(B), dim3(W * 64), lds, s, xyz, N, M, start, out_idx, out_xyz);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

template <int PPT, int W>
int launch_fps(const float *xyz, int B, int N, int M, const int64_t *start, int64_t *out_idx,
               float *out_xyz, hipStream_t s)
{
    const size_t base = sizeof(fps_slot) * 2 * W + (size_t)M * 4 + (size_t)M * 12;
    const size_t with_cloud = base + (size_t)N * 12;
    static const int allow = [] { const char *e = getenv("PPT_FPS_XYZ_LDS"); return e ? atoi(e) : 1; }();
    if (allow && with_cloud <= 150 * 1024) return launch_fps_v<PPT, W, true>(xyz, B, N, M, start, out_idx, out_xyz, s, with_cloud);
    if (base > 160 * 1024) return PPT_EUNSUPPORTED;
    return launch_fps_v<PPT, W, false>(xyz, B, N, M, start, out_idx, out_xyz, s, base);
}

}  // namespace

extern "C" int ppt_fps_f32(const float *xyz, int B, int N, int M, const int64_t *start, int64_t *out_idx,
                           float *out_xyz, void *stream)
{
    if (!xyz || !start || !out_idx || B <= 0 || N <= 0 || M <= 0 || N > 16384) return PPT_EINVAL;
    hipStream_t s = ppt_stream(stream);
    if (N <= 256) return launch_fps<1, 4>(xyz, B, N, M, start, out_idx, out_xyz, s);
    if (N <= 512) return launch_fps<2, 4>(xyz, B, N, M, start, out_idx, out_xyz, s);
    // points per lane x waves, measured with the cloud in LDS (tools/fps_bench.py, B = 32): more waves only lengthen the
    // barrier and the slot fold -- 8192 points: <8,16> 732 us, <16,8> 495 us, <32,4> 558 us; 2048: <4,8> 300, <8,4> 286,
    // <16,2> 326; 1024: <4,4> 234, <8,2> 244, <16,1> 302
    if (N <= 1024) return launch_fps<4, 4>(xyz, B, N, M, start, out_idx, out_xyz, s);
    if (N <= 2048) return launch_fps<8, 4>(xyz, B, N, M, start, out_idx, out_xyz, s);
    if (N <= 4096) return launch_fps<16, 4>(xyz, B, N, M, start, out_idx, out_xyz, s);
    if (N <= 8192) return launch_fps<16, 8>(xyz, B, N, M, start, out_idx, out_xyz, s);
    return launch_fps<16, 16>(xyz, B, N, M, start, out_idx, out_xyz, s);
}
