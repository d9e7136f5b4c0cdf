// What does one v_mfma_f32_32x32x16_f16 cost in s_memtime ticks and in nanoseconds, alone and with a second wave on the SIMD,
// at the clock the device actually runs a matrix-heavy kernel at?  (tools/micro/fwd_probe.py reports forward phases in ticks.)
// build: hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int n) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const _Float16 c = (_Float16)(float)(threadIdx.x & 7);
    h8 x = {c, c, c, c, c, c, c, c}, y = x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[2 * blockIdx.x] = t1 - t0, ticks[2 * blockIdx.x + 1] = w1 - w0;
}

int main() {
    float* out;
    unsigned long long* t;
    CK(hipMalloc(&out, 1024 * 512 * 4));
    CK(hipMalloc(&t, 1024 * 16));
    const int n = 20000;
    for (int threads : {256, 512})
        for (int blocks : {1, 256, 1024}) {
            hipEvent_t a, b;
            CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            k<<<blocks, threads>>>(out, t, n);
            CK(hipEventRecord(a));
            k<<<blocks, threads>>>(out, t, n);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            unsigned long long h[2];
            CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
            const double mf = 4.0 * n;     // MFMAs per wave
            const int per_simd = threads / 256;
            printf("%4d blocks x %d threads (%d waves/SIMD): %.1f us; per MFMA of one wave: %.2f ticks, %.2f ns (wall clock 100 MHz), "
                   "kernel/MFMA %.2f ns; pipe time per MFMA %.2f ns => %.2f GHz if 32 cycles\n", blocks, threads, per_simd,
                   ms * 1e3, h[0] / mf, h[1] * 10.0 / mf, ms * 1e6 / mf, ms * 1e6 / mf / per_simd, 32.0 / (ms * 1e6 / mf / per_simd));
        }
    return 0;
}
