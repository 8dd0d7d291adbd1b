// knn_group.hip -- H2 (kNN grouping) and H7 (ball query) for gfx950.
//
// Replaces models/pointbert/dvae.py:116-149 + :171-180 (square_distance -> [B,G,N] matrix ->
// topk -> flat gather -> centre subtract) and models/pointnet2/pointnet2_utils.py:87-107
// (query_ball_point: [B,S,N] index tensor + full sort).  The distance matrix is never written:
//
//   workgroup = (cloud b, tile of centres); the cloud is staged ONCE into LDS as float4
//   {x, y, z, |p|^2} (coalesced 12*N-byte HBM read, conflict-free ds_read_b128 afterwards);
//   one wave per centre:
//     pass 1  every lane scans N/64 candidates, keeps the minimum order-key of its own candidates
//     bound   Tb = k-th smallest of the 64 lane minima (ballot bit-search, registers only); the
//             true k-th distance is <= Tb, so everything > Tb is dead (typically ~97% of the cloud)
//     pass 2  re-scan, compact the survivors' 64-bit keys (order-key << 32 | index) into a per-wave
//             LDS list with ballot/mbcnt
//     rank    each survivor counts the smaller keys -> its rank; rank < k writes output slot `rank`
//             (so the neighbours come out sorted by (distance, index), deterministically)
//   a degenerate cloud that overflows the list (hundreds of exact ties) takes an exact but slow
//   k-round arg-min path.
//
// Arithmetic is the reference's expanded form with its exact rounding sequence (ppt_common.h).
#include "ppt_common.h"

namespace {

constexpr int KNN_CAP = 256;   // survivors per wave kept in LDS

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// k-th smallest (1-based) of the 64 per-lane values; registers + ballot only.
__device__ __forceinline__ uint32_t wave_kth_smallest(uint32_t v, int k)
{
    uint32_t T = 0;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = T | ((1u << bit) - 1u);            // T with all lower bits set
        const int cnt = __popcll(__ballot(v <= cand));
        if (cnt < k) T |= (1u << bit);
    }
    return T;
}

__device__ __forceinline__ void stage_cloud(const float *__restrict__ p, int N, float4 *cloud)
{
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float x = p[i * 3 + 0], y = p[i * 3 + 1], z = p[i * 3 + 2];
        cloud[i] = make_float4(x, y, z, sqnorm3_rn(x, y, z));
    }
}

template <int W>
__global__ __launch_bounds__(W * 64) void knn_group_kernel(const float *__restrict__ xyz,
                                                           const float *__restrict__ center, int N, int G,
                                                           int k, int cpb, int64_t *__restrict__ nbr_idx,
                                                           float *__restrict__ nbhd, float *__restrict__ ndist)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float4 *cloud = reinterpret_cast<float4 *>(smem);                                   // [N]
    unsigned long long *lists = reinterpret_cast<unsigned long long *>(smem + (size_t)N * 16);  // [W][KNN_CAP]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    stage_cloud(xyz + (size_t)b * N * 3, N, cloud);
    __syncthreads();

    unsigned long long *list = lists + (size_t)w * KNN_CAP;
    const int c_end = min(G, (int)(blockIdx.x + 1) * cpb);
    for (int c = blockIdx.x * cpb + w; c < c_end; c += W) {
        const float *q = center + ((size_t)b * G + c) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];
        const float nq = sqnorm3_rn(qx, qy, qz);

        // pass 1: per-lane minimum key
        uint32_t umin = 0xFFFFFFFFu;
        for (int i = lane; i < N; i += 64) {
            const float4 pt = cloud[i];
            umin = ppt_umin(umin, float_order_key(expanded_sqdist_rn(qx, qy, qz, nq, pt.x, pt.y, pt.z, pt.w)));
        }
        const uint32_t Tb = wave_kth_smallest(umin, k);

        // pass 2: compact survivors
        int cnt = 0;
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            bool alive = false;
            uint32_t u = 0;
            if (i < N) {
                const float4 pt = cloud[i];
                u = float_order_key(expanded_sqdist_rn(qx, qy, qz, nq, pt.x, pt.y, pt.z, pt.w));
                alive = u <= Tb;
            }
            const unsigned long long mask = __ballot(alive);
            if (alive) {
                const int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (pos < KNN_CAP) list[pos] = ((unsigned long long)u << 32) | (uint32_t)i;
            }
            cnt += __popcll(mask);
        }
        wave_lds_fence();

        int64_t *oi = nbr_idx ? nbr_idx + ((size_t)b * G + c) * k : nullptr;
        float *on = nbhd ? nbhd + ((size_t)b * G + c) * k * 3 : nullptr;
        float *od = ndist ? ndist + ((size_t)b * G + c) * k : nullptr;
        if (cnt <= KNN_CAP) {
            // rank by counting (keys are unique: the index is part of the key)
            for (int e0 = 0; e0 < cnt; e0 += 64) {
                const int e = e0 + lane;
                const unsigned long long mine = e < cnt ? list[e] : ~0ull;
                int rank = 0;
                for (int f = 0; f < cnt; ++f) rank += (list[f] < mine) ? 1 : 0;
                if (e < cnt && rank < k) {
                    const uint32_t i = (uint32_t)mine;
                    if (oi) oi[rank] = (int64_t)i;
                    if (od) {
                        const float4 pt = cloud[i];
                        od[rank] = expanded_sqdist_rn(qx, qy, qz, nq, pt.x, pt.y, pt.z, pt.w);
                    }
                    if (on) {
                        const float4 pt = cloud[i];
                        on[rank * 3 + 0] = __fsub_rn(pt.x, qx);
                        on[rank * 3 + 1] = __fsub_rn(pt.y, qy);
                        on[rank * 3 + 2] = __fsub_rn(pt.z, qz);
                    }
                }
            }
        } else {
            // exact fallback: k rounds of "smallest key greater than the previous one"
            unsigned long long prev = 0;
            bool have_prev = false;
            for (int r = 0; r < k; ++r) {
                uint32_t bhi = 0xFFFFFFFFu, blo = 0xFFFFFFFFu;
                for (int i = lane; i < N; i += 64) {
                    const float4 pt = cloud[i];
                    const uint32_t u = float_order_key(expanded_sqdist_rn(qx, qy, qz, nq, pt.x, pt.y, pt.z, pt.w));
                    const unsigned long long key = ((unsigned long long)u << 32) | (uint32_t)i;
                    if ((!have_prev || key > prev) && (u < bhi || (u == bhi && (uint32_t)i < blo))) { bhi = u; blo = i; }
                }
                const uint32_t mh = wave_reduce_umin(bhi);
                const uint32_t ml = wave_reduce_umin(bhi == mh ? blo : 0xFFFFFFFFu);
                prev = ((unsigned long long)mh << 32) | ml;
                have_prev = true;
                if (lane == 0) {
                    if (oi) oi[r] = (int64_t)ml;
                    if (od) {
                        const float4 pt = cloud[ml];
                        od[r] = expanded_sqdist_rn(qx, qy, qz, nq, pt.x, pt.y, pt.z, pt.w);
                    }
                    if (on) {
                        const float4 pt = cloud[ml];
                        on[r * 3 + 0] = __fsub_rn(pt.x, qx);
                        on[r * 3 + 1] = __fsub_rn(pt.y, qy);
                        on[r * 3 + 2] = __fsub_rn(pt.z, qz);
                    }
                }
            }
        }
        wave_lds_fence();   // the list is reused by this wave's next centre
    }
}

// Ball query: first K indices (ascending) with NOT(d > r^2); pad with the first hit
// (pointnet2_utils.py:100-107).  Same staging; one wave per centre, 64 candidates per step.
template <int W>
__global__ __launch_bounds__(W * 64) void ball_query_kernel(const float *__restrict__ xyz,
                                                            const float *__restrict__ center, int N, int S,
                                                            float r2, int K, int cpb, int64_t *__restrict__ idx,
                                                            float *__restrict__ gxyz)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float4 *cloud = reinterpret_cast<float4 *>(smem);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    stage_cloud(xyz + (size_t)b * N * 3, N, cloud);
    __syncthreads();
    const int c_end = min(S, (int)(blockIdx.x + 1) * cpb);
    for (int c = blockIdx.x * cpb + w; c < c_end; c += W) {
        const float *q = center + ((size_t)b * S + c) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];
        const float nq = sqnorm3_rn(qx, qy, qz);
        int64_t *o = idx + ((size_t)b * S + c) * K;
        int cnt = 0;
        int first = N;   // reference fill value when nothing is inside the ball
        for (int base = 0; base < N && cnt < K; base += 64) {
            const int i = base + lane;
            bool hit = false;
            if (i < N) {
                const float4 pt = cloud[i];
                hit = !(expanded_sqdist_rn(qx, qy, qz, nq, pt.x, pt.y, pt.z, pt.w) > r2);
            }
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                if (cnt == 0) first = base + (int)__builtin_ctzll(mask);
                if (hit) {
                    const int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                    if (pos < K) o[pos] = (int64_t)i;
                }
                cnt += __popcll(mask);
            }
        }
        for (int j = min(cnt, K) + lane; j < K; j += 64) o[j] = (int64_t)first;
        if (gxyz) {                                   // grouped_xyz - new_xyz (pointnet2_utils.py:243-244)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float *g = gxyz + ((size_t)b * S + c) * K * 3;
            for (int j = lane; j < K; j += 64) {
                const int i = min((int)o[j], N - 1);
                const float4 pt = cloud[i];
                g[j * 3 + 0] = __fsub_rn(pt.x, qx); g[j * 3 + 1] = __fsub_rn(pt.y, qy); g[j * 3 + 2] = __fsub_rn(pt.z, qz);
            }
        }
    }
}

// The ball queries of ONE multi-scale set-abstraction level (PointNetSetAbstractionMsg, pointnet2_utils.py:228-266: the same
// centres, up to three radii) in a single pass: the cloud is staged once, every point's distance is evaluated once and tested
// against all radii (the reference evaluates square_distance once per radius, :99; three launches of ball_query_kernel staged
// the 128 KB of an 8192-point cloud three times and walked it three times).  Hits write the index AND the centred
// coordinates at once (no read-back of the index list); a list is padded from the first hit, which is kept in a register.
template <int W, int NR>
__global__ __launch_bounds__(W * 64) void ball_query_multi_kernel(const float *__restrict__ xyz, const float *__restrict__ center, int N,
                                                                  int S, int cpb, ppt_ball_multi q)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float4 *cloud = reinterpret_cast<float4 *>(smem);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    stage_cloud(xyz + (size_t)b * N * 3, N, cloud);
    __syncthreads();
    const int c_end = min(S, (int)(blockIdx.x + 1) * cpb);
    for (int c = blockIdx.x * cpb + w; c < c_end; c += W) {
        const float *qc = center + ((size_t)b * S + c) * 3;
        const float qx = qc[0], qy = qc[1], qz = qc[2];
        const float nq = sqnorm3_rn(qx, qy, qz);
        int cnt[NR], first[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) { cnt[j] = 0; first[j] = N; }
        bool more = true;
        for (int base = 0; base < N && more; base += 64) {
            const int i = base + lane;
            float d = INFINITY;
            float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < N) {
                pt = cloud[i];
                d = expanded_sqdist_rn(qx, qy, qz, nq, pt.x, pt.y, pt.z, pt.w);
            }
            more = false;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (j < q.n && cnt[j] < q.K[j]) {                          // (wave-uniform)
                    const bool hit = i < N && !(d > q.r2[j]);
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {
                        if (cnt[j] == 0) first[j] = base + (int)__builtin_ctzll(mask);
                        if (hit) {
                            const int pos = cnt[j] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                            if (pos < q.K[j]) {
                                const size_t o = ((size_t)b * S + c) * q.K[j] + pos;
                                q.idx[j][o] = (int64_t)i;
                                if (q.gxyz[j]) {
                                    float *g = q.gxyz[j] + o * 3;
                                    g[0] = __fsub_rn(pt.x, qx); g[1] = __fsub_rn(pt.y, qy); g[2] = __fsub_rn(pt.z, qz);
                                }
                            }
                        }
                        cnt[j] += __popcll(mask);
                    }
                    more = more || cnt[j] < q.K[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (j < q.n) {
                const float4 pf = cloud[min(first[j], N - 1)];
                for (int k = min(cnt[j], q.K[j]) + lane; k < q.K[j]; k += 64) {
                    const size_t o = ((size_t)b * S + c) * q.K[j] + k;
                    q.idx[j][o] = (int64_t)first[j];
                    if (q.gxyz[j]) {
                        float *g = q.gxyz[j] + o * 3;
                        g[0] = __fsub_rn(pf.x, qx); g[1] = __fsub_rn(pf.y, qy); g[2] = __fsub_rn(pf.z, qz);
                    }
                }
            }
        }
    }
}

// dist[b, s, n] of square_distance (dvae.py:130-149), the reference's expanded form with its rounding sequence; one thread
// per 4 consecutive n of one (b, s): 16-byte stores along the rows
__global__ __launch_bounds__(256) void square_distance_kernel(const float *__restrict__ src, const float *__restrict__ dst, int S, int N,
                                                              int64_t total4, float *__restrict__ out)
{
    const int n4 = (N + 3) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t bs = i / n4;
        const int n = 4 * (int)(i - bs * n4);
        const int64_t b = bs / S;
        const float *a = src + bs * 3;
        const float ax = a[0], ay = a[1], az = a[2];
        const float na = sqnorm3_rn(ax, ay, az);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int nn = min(n + e, N - 1);
            const float *c = dst + (b * N + nn) * 3;
            const float cx = c[0], cy = c[1], cz = c[2];
            v[e] = expanded_sqdist_rn(ax, ay, az, na, cx, cy, cz, sqnorm3_rn(cx, cy, cz));
        }
        float *o = out + bs * N + n;
        if (n + 3 < N && (N & 3) == 0) *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int e = 0; e < 4 && n + e < N; ++e) o[e] = v[e];
    }
}

}  // namespace

extern "C" int ppt_square_distance_f32(const float *src, const float *dst, int B, int S, int N, float *out, void *stream)
{
    if (!src || !dst || !out || B <= 0 || S <= 0 || N <= 0) return PPT_EINVAL;
    const int64_t total4 = (int64_t)B * S * ((N + 3) / 4);
    const int grid = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(square_distance_kernel, dim3(grid), dim3(256), 0, ppt_stream(stream), src, dst, S, N, total4, out);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_ball_query_multi_f32(const float *xyz, const float *center, int B, int N, int S, const ppt_ball_multi *q, void *stream)
{
    if (!xyz || !center || !q || B <= 0 || N <= 0 || S <= 0 || N > 10240 || q->n < 1 || q->n > 3) return PPT_EINVAL;
    for (int j = 0; j < q->n; ++j)
        if (!q->idx[j] || q->K[j] <= 0) return PPT_EINVAL;
    constexpr int W = 8;
    const size_t lds = (size_t)N * 16;
    static const hipError_t optin = hipFuncSetAttribute((const void *)ball_query_multi_kernel<W, 3>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 10240 * 16);
    if (lds > 64 * 1024 && optin != hipSuccess) return PPT_ELAUNCH;
    const int cpb = 32;
    dim3 grid((S + cpb - 1) / cpb, B);
    hipLaunchKernelGGL((ball_query_multi_kernel<W, 3>), grid, dim3(W * 64), lds, ppt_stream(stream), xyz, center, N, S, cpb, *q);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_knn_group_f32(const float *xyz, const float *center, int B, int N, int G, int k,
                                 int64_t *nbr_idx, float *neighborhood, float *nbr_dist, void *stream)
{
    if (!xyz || !center || B <= 0 || N <= 0 || G <= 0 || k <= 0 || k > 64 || k > N || N > 8192)
        return PPT_EINVAL;
    constexpr int W = 8;
    const size_t lds = (size_t)N * 16 + (size_t)W * KNN_CAP * 8;
    // the opt-in to > 64 KB of dynamic LDS is made ONCE per kernel (thread-safe static initialisation), for the
    // largest cloud the entry point accepts -- not on every launch
    static const hipError_t optin = hipFuncSetAttribute((const void *)knn_group_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         8192 * 16 + W * KNN_CAP * 8);
    if (lds > 64 * 1024 && optin != hipSuccess) return PPT_ELAUNCH;
    const int cpb = 32;
    dim3 grid((G + cpb - 1) / cpb, B);
    hipLaunchKernelGGL((knn_group_kernel<W>), grid, dim3(W * 64), lds, ppt_stream(stream), xyz, center, N, G, k,
                       cpb, nbr_idx, neighborhood, nbr_dist);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_ball_query_f32(const float *xyz, const float *center, int B, int N, int S, float radius_sq,
                                  int K, int64_t *idx, float *grouped_xyz, void *stream)
{
    if (!xyz || !center || !idx || B <= 0 || N <= 0 || S <= 0 || K <= 0 || N > 10240) return PPT_EINVAL;
    constexpr int W = 8;
    const size_t lds = (size_t)N * 16;
    static const hipError_t optin = hipFuncSetAttribute((const void *)ball_query_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         10240 * 16);
    if (lds > 64 * 1024 && optin != hipSuccess) return PPT_ELAUNCH;
    const int cpb = 32;
    dim3 grid((S + cpb - 1) / cpb, B);
    hipLaunchKernelGGL((ball_query_kernel<W>), grid, dim3(W * 64), lds, ppt_stream(stream), xyz, center, N, S,
                       radius_sq, K, cpb, idx, grouped_xyz);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
