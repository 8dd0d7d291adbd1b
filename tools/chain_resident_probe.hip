// Dev microbenchmark for VERDICT r3 #4 (a resident multi-CU kernel for the prompt chain): what does ONE phase boundary cost as
// (a) a dependent kernel launch on a stream -- what the text tower's ~85 forward kernels pay today -- and
// (b) a device-scope grid barrier inside one persistent launch of the same G workgroups,
// alone on the chip and beside a tower-like load on another stream?  Every phase does the same small piece of work: each of
// the G workgroups (256 threads) streams `kb` KiB from an L2-resident buffer (the K loop of a 64 x 64 GEMM tile) and writes 8 KiB.
// The barrier is a monotonic counter: lane 0 release-fences and adds, then polls with L2-coherent loads + s_sleep (bounded:
// a workgroup that was never made resident cannot hang the box); acquire fence behind it.
//   hipcc -O3 --offload-arch=gfx950 -fPIC -shared tools/chain_resident_probe.hip -o tools/_build/libchain_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ void phase_work(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int kb, int phase, int G)
{
    // kb KiB per workgroup: 256 threads x 16 B = 4 KiB per sweep; a different window per phase and workgroup
    const size_t base = ((size_t)((blockIdx.x * 131 + phase * 17) % (G * 8)) * (size_t)kb * 64);     // in uint4 units
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
    for (int i = 0; i < kb / 4; ++i) {
        const uint4 v = src[base + (size_t)i * 256 + threadIdx.x];
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
    for (int i = 0; i < 2; ++i) dst[((size_t)blockIdx.x * 2 + i) * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void phase_kernel(const uint4 *src, uint4 *dst, int kb, int phase, int G)
{
    phase_work(src, dst, kb, phase, G);
}

__global__ __launch_bounds__(256) void resident_kernel(const uint4 *src, uint4 *dst, int kb, int P, int G, unsigned int *counter, int *err)
{
    for (int ph = 0; ph < P; ++ph) {
        phase_work(src, dst, kb, ph, G);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int want = (unsigned int)(ph + 1) * (unsigned int)G;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 20)) { *err = 1; break; }          // (never hang the box)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (*((volatile int *)err)) return;
    }
}

}  // namespace

extern "C" int probe_launches(const void *src, void *dst, int kb, int P, int G, void *stream)
{
    for (int ph = 0; ph < P; ++ph)
        hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst, kb, ph, G);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int probe_resident(const void *src, void *dst, int kb, int P, int G, unsigned int *counter, int *err, void *stream)
{
    hipLaunchKernelGGL(resident_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst, kb, P, G, counter, err);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
