// Dev microbenchmark (round 6): the WEIGHT STREAM of the fused MLP kernel in isolation.  Every workgroup streams the
// same 2.36 MB region (fc1 + fc2 of one block) over and over, wave w reading its own run of 1 KiB fragments ([slab][wave][fragment]
// order, as ppt_vit_mlp_retile lays them out) into a register ring that keeps DEPTH loads in flight continuously (counted vmcnt),
// with nothing else going on.  Question: is the ~21-26 B/clk/CU the kernels see a per-CU limit, a per-wave limit or an aggregate
// (per-XCD L2) limit?  Sweep: workgroups (= CUs in use), waves per workgroup, ring depth.
// hipcc -O3 --offload-arch=gfx950 tools/wstream_bench.hip -o /tmp/wstream && /tmp/wstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NW, int DEPTH, int RUN>     // RUN: fragments per (slab, wave) run (12: v2's 128-unit slabs, 24: 256-unit slabs)
__global__ __launch_bounds__(NW * 64) void wstream(const unsigned char *src, size_t region, int passes, int *sink)
{
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(src), 0, (int)region, 0x00020000);
    const int per_wave = (int)(region / 1024) / NW;          // fragments of this wave per pass
    uint4 ring[DEPTH];
    uint4 acc = make_uint4(0, 0, 0, 0);
    auto frag_off = [&](int f) {                             // f-th fragment of this wave: runs of RUN consecutive KiB, NW runs per slab
        const int run = f / RUN, r = f % RUN;
        return ((run * NW + w) * RUN + r) * 1024;
    };
    int f = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, frag_off(d % per_wave), 0));
    const int total = passes * per_wave;
    for (f = 0; f + DEPTH <= total; f += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
            acc.x ^= ring[d].x; acc.y ^= ring[d].y; acc.z ^= ring[d].z; acc.w ^= ring[d].w;
            ring[d] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, frag_off((f + DEPTH + d) % per_wave), 0));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345 && passes < 0) sink[0] = 1;
}

template <typename F>
static float time_kernel(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main()
{
    const size_t region = 2359296;                           // 2 x 1536 x 384 x 2 bytes
    unsigned char *src; int *sink;
    hipMalloc(&src, region); hipMemset(src, 1, region); hipMalloc(&sink, 4);
    const int passes = 40;
    printf("| CUs in use | waves | ring depth | run | GB/s per CU | B/clk/CU @2.1 GHz | TB/s chip |\n|---|---|---|---|---|---|---|\n");
#define RUN_ONE(NW, D, R)                                                                                                        \
    for (int grid : {32, 64, 128, 171, 206, 256}) {                                                                              \
        float ms = time_kernel([&] { hipLaunchKernelGGL((wstream<NW, D, R>), dim3(grid), dim3(NW * 64), 0, 0, src, region, passes, sink); }); \
        const int per_wave = (int)(region / 1024) / NW;                                                                          \
        double bytes = (double)grid * NW * (double)((passes * per_wave) / D * D) * 1024.0;                                       \
        double per_cu = bytes / grid / ms / 1e6;                                                                                 \
        printf("| %d | %d | %d | %d | %.1f | %.1f | %.2f |\n", grid, NW, D, R, per_cu, per_cu / 2.1, bytes / ms / 1e9);          \
    }
    RUN_ONE(4, 8, 12) RUN_ONE(8, 4, 12) RUN_ONE(8, 8, 12) RUN_ONE(8, 16, 12) RUN_ONE(8, 8, 24) RUN_ONE(8, 24, 24)
    RUN_ONE(12, 8, 12) RUN_ONE(16, 4, 12) RUN_ONE(16, 8, 12) RUN_ONE(16, 16, 12)
    return 0;
}
