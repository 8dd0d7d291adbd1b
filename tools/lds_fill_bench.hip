// Dev microbenchmark: how fast can a CU pull L2-resident data (a) into LDS with global_load_lds_dwordx4,
// (b) into VGPRs with global_load_dwordx4?  Sets the ceiling for the 64x64 / 128x128 GEMM tile loops.
// hipcc -O3 --offload-arch=gfx950 tools/lds_fill_bench.hip -o /tmp/lds_fill && /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH, int LDS_KB>
__global__ __launch_bounds__(256) void fill_lds(const unsigned char *src, size_t region, int iters, int *sink)
{
    __shared__ __align__(16) unsigned char smem[LDS_KB * 1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // every block walks its own window of the region (wraps), 4 KiB per block per step: 1 KiB per wave
    size_t off = ((size_t)blockIdx.x * 65536) % region;
    const int stages = LDS_KB / 4;
    int st = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned char *g = src + (off + (size_t)w * 1024 + lane * 16) % region;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(smem + st * 4096 + w * 1024), 16, 0, 0);
            off = (off + 4096) % region;
            st = st + 1 == stages ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (smem[threadIdx.x * 16] == 0x7b && iters < 0) sink[0] = 1;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void fill_reg(const unsigned char *src, size_t region, int iters, int *sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    size_t off = ((size_t)blockIdx.x * 65536) % region;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            v[d] = *reinterpret_cast<const uint4 *>(src + (off + (size_t)w * 1024 + lane * 16) % region);
            off = (off + 4096) % region;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345 && iters < 0) sink[0] = 1;
}

template <typename F>
float time_kernel(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main()
{
    const size_t cap = 512u << 20;
    unsigned char *src; int *sink;
    hipMalloc(&src, cap); hipMemset(src, 1, cap); hipMalloc(&sink, 4);
    const int iters = 400;
    for (size_t region : {(size_t)1 << 20, (size_t)16 << 20, (size_t)256 << 20}) {
        for (int bpc : {1, 2, 3, 4, 6, 8}) {
            const int grid = 256 * bpc;
#define RUN_LDS(D, KB)                                                                                              \
            {                                                                                                       \
                float ms = time_kernel([&] { hipLaunchKernelGGL((fill_lds<D, KB>), dim3(grid), dim3(256), 0, 0, src, region, iters, sink); }); \
                double bytes = (double)grid * iters * D * 4096;                                                     \
                printf("region %4zu MiB  blocks/CU %d  LDS-DMA depth %2d (%2d KiB LDS): %7.2f TB/s\n", region >> 20, bpc, D, KB, bytes / ms / 1e9); \
            }
            if (bpc * 16 <= 160) RUN_LDS(4, 16)
            if (bpc * 32 <= 160) RUN_LDS(8, 32)
            if (bpc * 48 <= 160) RUN_LDS(12, 48)
#define RUN_REG(D)                                                                                                  \
            {                                                                                                       \
                float ms = time_kernel([&] { hipLaunchKernelGGL((fill_reg<D>), dim3(grid), dim3(256), 0, 0, src, region, iters, sink); }); \
                double bytes = (double)grid * iters * D * 4096;                                                     \
                printf("region %4zu MiB  blocks/CU %d  VGPR    depth %2d              : %7.2f TB/s\n", region >> 20, bpc, D, bytes / ms / 1e9); \
            }
            RUN_REG(4) RUN_REG(8)
        }
    }
    return 0;
}
