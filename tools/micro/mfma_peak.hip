// Practical fp32 MFMA issue rate on MI355X: v_mfma_f32_32x32x2_f32 with NACC independent accumulators, W waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int threads) {
    float* out;
    (void)hipMalloc(&out, 256 * 16 * 1024 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double mfma = waves * iters * 8.0 * NACC;
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    const double waves_per_simd = waves / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 8.0 * NACC * waves_per_simd);
    printf("NACC=%d blocks/CU=%d threads=%d: %.3f ms  %.1f TF/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", NACC,
           blocks_per_cu, threads, ms, tf, cyc);
    hipFree(out);
}
int main() {
    run<1>(1, 256);
    run<2>(1, 256);
    run<4>(1, 256);
    run<4>(2, 256);
    run<8>(1, 256);
    run<4>(1, 512);
    // sustained: back-to-back launches for ~0.5 s -- does the clock hold?
    for (int r = 0; r < 300; ++r) {
        if (r % 30 == 0) run<4>(1, 256);
        else {
            float* out;
            (void)hipMalloc(&out, 256 * 16 * 1024 * 4);
            hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, 2000, 1.f, 2.f);
            (void)hipDeviceSynchronize();
            (void)hipFree(out);
        }
    }
    return 0;
}
