// How fast are float atomics that stay in the XCD's own L2?  Device-scope (agent) global float atomics go to the memory
// side on this multi-XCD part (~18 G/s whatever the pattern, round 1).  A WORKGROUP-scope atomic is performed by the XCD's L2
// without the system-coherence flag; it is only correct when all writers of an address sit on the same XCD -- which a kernel
// can arrange (block b runs on XCD b % 8: pin a hashed level's table gradient to one XCD).  This measures the rate:
//   mode 0  agent scope, every block anywhere in one 4 MiB region per XCD group
//   mode 1  workgroup scope, blocks of XCD k add into region k (4 MiB each, random entries, float2 = two atomics)
//   mode 2  workgroup scope, all blocks into ONE 4 MiB region (wrong in general; shows the cross-XCD cost)
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics l2atomic.hip -o l2atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ table, uint32_t entries_per_region, uint32_t per_thread) {
    const uint32_t xcd = blockIdx.x & 7u;
    float* reg = table + (MODE == 2 ? 0u : (size_t)xcd * entries_per_region * 2);
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t i = 0; i < per_thread; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t e = (s >> 8) % entries_per_region;
        if (MODE == 0) {
            __hip_atomic_fetch_add(reg + 2 * e, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(reg + 2 * e + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(reg + 2 * e, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(reg + 2 * e + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

int main() {
    const uint32_t entries = 1u << 19;                 // 2^19 float2 = 4 MiB per region
    float* t;
    CK(hipMalloc(&t, (size_t)8 * entries * 8));
    CK(hipMemset(t, 0, (size_t)8 * entries * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode)
        for (int blocks : {256 * 4, 256 * 16}) {
            const uint32_t per_thread = 64;
            auto go = [&]() {
                if (mode == 0) k<0><<<blocks, 256>>>(t, entries, per_thread);
                else if (mode == 1) k<1><<<blocks, 256>>>(t, entries, per_thread);
                else k<2><<<blocks, 256>>>(t, entries, per_thread);
            };
            go();
            CK(hipEventRecord(a));
            for (int r = 0; r < 5; ++r) go();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            const double n = (double)blocks * 256 * per_thread * 2 * 5;
            printf("mode %d blocks %5d: %8.1f us per launch, %7.2f G float atomics/s (%.2f G per XCD)\n", mode, blocks,
                   ms / 5 * 1e3, n / (ms * 1e-3) / 1e9, n / (ms * 1e-3) / 1e9 / 8);
        }
    // correctness of mode 1: every region's sum must equal the number of adds into it
    CK(hipMemset(t, 0, (size_t)8 * entries * 8));
    k<1><<<1024, 256>>>(t, entries, 64);
    CK(hipDeviceSynchronize());
    float* h = (float*)malloc((size_t)8 * entries * 8);
    CK(hipMemcpy(h, t, (size_t)8 * entries * 8, hipMemcpyDeviceToHost));
    double s0 = 0, s1 = 0;
    for (size_t i = 0; i < (size_t)8 * entries; ++i) s0 += h[2 * i], s1 += h[2 * i + 1];
    printf("mode 1 check: sum f0 %.0f (want %.0f), sum f1 %.0f (want %.0f)\n", s0, 1024.0 * 256 * 64, s1, 2.0 * 1024 * 256 * 64);
    return 0;
}
