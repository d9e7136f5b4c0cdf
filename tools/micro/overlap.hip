// Do the matrix pipe and the vector ALU of a SIMD overlap?  2 waves per SIMD (512-thread blocks, one per CU).
//   mode 0: every wave: N dependent bf16 MFMAs (32x32x16)                    mode 1: every wave: N x V vector instructions
//   mode 2: every wave: both, independent, interleaved in one instruction stream
//   mode 3: waves 0..3 MFMA only, waves 4..7 vector only (one of each per SIMD)
//   mode 4: every wave alternates a burst of 6 MFMAs and a burst of 6 V vector instructions (the wgrad16 pattern)
// build: hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__device__ __forceinline__ void valu_burst(float (&r)[8]) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
        // cvt_pk + shift + sub: the cut sequence
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            bf2 t;
            t[0] = (__bf16)r[i], t[1] = (__bf16)r[i + 1];
            r[i] = r[i] - (float)t[0] * 0.5f;
            r[i + 1] = r[i + 1] - (float)t[1] * 0.5f;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int n) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(lane + i), b[i] = (__bf16)(float)(lane ^ i);
    f32x16 acc = {0}, acc2 = {0};
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = 1.0f + lane * 0.001f + i;
    const bool do_m = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && w < 4);
    const bool do_v = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && w >= 4);
    if (MODE == 4) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu_burst<6>(r);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (MODE == 2) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                valu_burst<1>(r);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    } else if (do_v) {
        for (int it = 0; it < n; ++it) valu_burst<6>(r);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int n = 20000;
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, n); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, n); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, n); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, n); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, n); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // per wave per iteration: 6 MFMAs (6 x 32 cycles at 1 wave; 2 waves/SIMD -> 12 x 32) ; vector: 6 bursts
        printf("mode %d: %.3f ms  -> %.1f ns per iteration per SIMD-pair\n", mode, best, best * 1e6 / n);
    }
    return 0;
}
