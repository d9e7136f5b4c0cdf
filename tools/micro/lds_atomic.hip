// LDS atomic throughput on MI355X: ds_add_f32 vs ds_add_u32, random vs conflict-free addresses, lane occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 -w -munsafe-fp-atomics tools/micro/lds_atomic.hip -o tools/micro/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int SLICE = 40960;  // floats (160 KB)

template <int MODE, bool INT>
__global__ __launch_bounds__(1024) void k(float* out, int iters, uint32_t keep_mask, uint32_t seed) {
    extern __shared__ float lds[];
    for (int q = threadIdx.x; q < SLICE; q += blockDim.x) lds[q] = 0.f;
    __syncthreads();
    uint32_t s = seed + blockIdx.x * 7919u + threadIdx.x * 104729u;
    const bool active = ((threadIdx.x * 2654435761u) >> 28 & 15u) < keep_mask;   // keep_mask/16 of the lanes
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t idx;
            if (MODE == 0) idx = (s >> 8) % SLICE;                                   // random
            else if (MODE == 1) idx = (threadIdx.x + 64 * u + 1024 * (it & 15)) % SLICE;   // conflict-free, distinct
            else idx = ((s >> 8) % (SLICE / 2)) * 2;                                 // random entry, feature 0 (then +1)
            if (active) {
                if (INT) {
                    atomicAdd(reinterpret_cast<unsigned int*>(lds) + idx, 1u);
                    if (MODE == 2) atomicAdd(reinterpret_cast<unsigned int*>(lds) + idx + 1, 1u);
                } else {
                    atomicAdd(lds + idx, 1.0f);
                    if (MODE == 2) atomicAdd(lds + idx + 1, 1.0f);
                }
            }
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int q = threadIdx.x; q < SLICE; q += blockDim.x) t += lds[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// 64-bit variants: double add and u64 add on SLICE/2 eight-byte cells
template <bool INT, int GROUP = 1>
__global__ __launch_bounds__(1024) void k64(float* out, int iters, uint32_t seed) {
    extern __shared__ float lds[];
    for (int q = threadIdx.x; q < SLICE; q += blockDim.x) lds[q] = 0.f;
    __syncthreads();
    uint32_t s = seed + blockIdx.x * 7919u + (threadIdx.x / GROUP) * 104729u;   // GROUP adjacent lanes share addresses
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t idx = (s >> 8) % (SLICE / 2);
            if (INT) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + idx, 1ull);
            else atomicAdd(reinterpret_cast<double*>(lds) + idx, 1.0);
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int q = threadIdx.x; q < SLICE; q += blockDim.x) t += lds[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <bool INT, int GROUP = 1>
void run64(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipFuncSetAttribute((const void*)k64<INT, GROUP>, hipFuncAttributeMaxDynamicSharedMemorySize, SLICE * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k64<INT, GROUP>), dim3(256), dim3(1024), SLICE * 4, 0, out, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k64<INT, GROUP>), dim3(256), dim3(1024), SLICE * 4, 0, out, iters, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double laneops = 256.0 * 1024 * iters * 8;
    printf("%-34s threads=1024 keep=16/16: %7.3f ms  %7.1f G lane-ops/s  %.3f cycles/lane-op/CU\n", name, ms,
           laneops / ms / 1e6, ms * 1e-3 * 2.4e9 / (laneops / 256));
    (void)hipFree(out);
}


// fp64 pairs as the grid scatter issues them (entry = two doubles, 16 bytes): `keep` of 16 lanes active, entries either random
// (PATTERN 0) or clustered like the corners of neighbouring dense cells (PATTERN 1: a wave's lanes fall into ~24 cells of a
// 17^3 lattice).  The question: does an LDS atomic instruction cost by ACTIVE lanes or by instruction?
template <int PATTERN>
__global__ __launch_bounds__(1024) void k64pair(float* out, int iters, uint32_t keep, uint32_t seed) {
    extern __shared__ float lds[];
    for (int q = threadIdx.x; q < SLICE; q += blockDim.x) lds[q] = 0.f;
    __syncthreads();
    uint32_t s = seed + blockIdx.x * 7919u + threadIdx.x * 104729u;
    const bool active = (threadIdx.x & 15u) < keep;
    double* acc = reinterpret_cast<double*>(lds);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t idx;
            if (PATTERN == 0) idx = (s >> 8) % (SLICE / 4);
            else {
                const uint32_t cell = (threadIdx.x & 63u) / 3u + (it & 7u) * 17u;     // stretches of three lanes, cells along x then y
                idx = (cell + (u & 1u) + ((u >> 1) & 1u) * 17u + ((u >> 2) & 1u) * 289u) % (SLICE / 4);
            }
            if (active) {
                atomicAdd(acc + 2u * idx, 1.0);
                atomicAdd(acc + 2u * idx + 1u, 1.0);
            }
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int q = threadIdx.x; q < SLICE; q += blockDim.x) t += lds[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <int PATTERN>
void run64pair(const char* name, uint32_t keep) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipFuncSetAttribute((const void*)k64pair<PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, SLICE * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k64pair<PATTERN>), dim3(256), dim3(1024), SLICE * 4, 0, out, 10, keep, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k64pair<PATTERN>), dim3(256), dim3(1024), SLICE * 4, 0, out, iters, keep, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = 16.0 * iters * 8 * 2;   // wave instructions per CU
    printf("%-34s keep=%2u/16: %7.3f ms  %.1f cycles per wave instruction per CU (2.4 GHz), %.3f per active lane-op\n", name, keep, ms,
           ms * 1e-3 * 2.4e9 / instr, ms * 1e-3 * 2.4e9 / (instr * 4.0 * keep));
    (void)hipFree(out);
}

template <int MODE, bool INT>
void run(const char* name, int threads, uint32_t keep) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipFuncSetAttribute((const void*)k<MODE, INT>, hipFuncAttributeMaxDynamicSharedMemorySize, SLICE * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, INT>), dim3(256), dim3(threads), SLICE * 4, 0, out, 10, keep, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, INT>), dim3(256), dim3(threads), SLICE * 4, 0, out, iters, keep, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double laneops = 256.0 * threads * iters * 8 * (MODE == 2 ? 2 : 1) * keep / 16.0;
    printf("%-34s threads=%4d keep=%2u/16: %7.3f ms  %7.1f G lane-ops/s  %.3f cycles/lane-op/CU\n", name, threads, keep, ms,
           laneops / ms / 1e6, ms * 1e-3 * 2.4e9 / (laneops / 256));
    (void)hipFree(out);
}

int main() {
    run<0, false>("f32 random", 1024, 16);
    run<0, false>("f32 random", 256, 16);
    run<0, false>("f32 random", 1024, 4);
    run<1, false>("f32 conflict-free", 1024, 16);
    run<2, false>("f32 random pair (F=2)", 1024, 16);
    run<2, false>("f32 random pair (F=2)", 1024, 4);
    run<0, true>("u32 random", 1024, 16);
    run<1, true>("u32 conflict-free", 1024, 16);
    run<2, true>("u32 random pair", 1024, 4);
    run64<false>("f64 random");
    run64<true>("u64 random");
    run64<false, 2>("f64, 2 lanes per address");
    run64<false, 4>("f64, 4 lanes per address");
    run64<false, 8>("f64, 8 lanes per address");
    run64<false, 64>("f64, 64 lanes per address");
    for (uint32_t keep : {16u, 8u, 4u, 2u, 1u}) run64pair<0>("f64 pair, random entries", keep);
    for (uint32_t keep : {16u, 8u, 4u, 2u, 1u}) run64pair<1>("f64 pair, dense-cell corners", keep);
    return 0;
}
