// What does this device stream?  Hand-written read / write / copy kernels over 1 GiB (past the 256 MiB Infinity Cache),
// 16 bytes per lane, persistent workgroups, U independent accesses in flight per lane.
//   mode r    global_load_dwordx4, default policy         mode rn   the same, non-temporal
//   mode rl   global_load_lds_dwordx4 (LDS-DMA)            mode rln  LDS-DMA, non-temporal (aux = 2)
//   mode w    global_store_dwordx4                         mode wn   non-temporal stores
//   mode c    copy (load + store)                          mode cn   copy, nt loads + nt stores
//   mode rw   two independent streams, one read, one written (the decoder kernels' mix; bytes = both)
// bytes in flight per CU = blocks_per_cu x threads x U x 16.
// build: hipcc --offload-arch=gfx950 -O3 stream.hip -o stream ; run: ./stream   (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { M_R, M_RN, M_RL, M_RLN, M_W, M_WN, M_C, M_CN, M_RW };

template <int MODE, int U, int T>
__global__ __launch_bounds__(T) void k(const f4* __restrict__ src, f4* __restrict__ dst, float* sink, size_t n16) {
    extern __shared__ f4 lds[];
    const size_t chunk = (size_t)T * U;                         // f4 elements a block moves per round
    const size_t stride = chunk * gridDim.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const f4 one = {1.f, 2.f, 3.f, 4.f};
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n16; base += stride) {
        const size_t i0 = base + threadIdx.x;
        if (MODE == M_R || MODE == M_RN || MODE == M_C || MODE == M_CN || MODE == M_RW) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                v[u] = (MODE == M_RN || MODE == M_CN) ? __builtin_nontemporal_load(src + i0 + (size_t)u * T) : src[i0 + (size_t)u * T];
            if (MODE == M_C || MODE == M_CN || MODE == M_RW) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (MODE == M_CN || MODE == M_RW) __builtin_nontemporal_store(v[u], dst + i0 + (size_t)u * T);
                    else dst[i0 + (size_t)u * T] = v[u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) acc += v[u];
            }
        } else if (MODE == M_RL || MODE == M_RLN) {
            // every wave lands U x 1 KB in its own LDS region; nothing reads it back (the stream is what is measured)
            const int w = threadIdx.x >> 6;
#pragma unroll
            for (int u = 0; u < U; ++u)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i0 + (size_t)u * T),
                                                 (void __attribute__((address_space(3)))*)(lds + (w * U + u) * 64), 16, 0,
                                                 MODE == M_RLN ? 2 : 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == M_WN) __builtin_nontemporal_store(one, dst + i0 + (size_t)u * T);
                else dst[i0 + (size_t)u * T] = one;
            }
        }
    }
    if (MODE == M_RL || MODE == M_RLN) {
        __syncthreads();
        acc = lds[threadIdx.x];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int MODE, int U, int T>
static float run(const f4* src, f4* dst, float* sink, size_t n16, int bpc, int reps) {
    const int grid = 256 * bpc;
    const size_t lds = (MODE == M_RL || MODE == M_RLN) ? (size_t)(T / 64) * U * 1024 : 16 * T;
    CK(hipFuncSetAttribute((const void*)k<MODE, U, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) k<MODE, U, T><<<grid, T, lds>>>(src, dst, sink, n16);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k<MODE, U, T><<<grid, T, lds>>>(src, dst, sink, n16);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int MODE>
static void sweep(const char* name, const f4* src, f4* dst, float* sink, size_t n16, double bytes_per_elem) {
    struct { int T, U, bpc; } cfg[] = {{256, 2, 1}, {256, 4, 1}, {512, 4, 1}, {512, 8, 1}, {256, 4, 4}, {256, 8, 4}, {512, 8, 2},
                                       {1024, 8, 1}, {1024, 8, 2}, {256, 16, 4}};
    for (auto c : cfg) {
        float ms = 0;
#define GO(TT, UU) if (c.T == TT && c.U == UU) ms = run<MODE, UU, TT>(src, dst, sink, n16, c.bpc, 10);
        GO(256, 2) GO(256, 4) GO(512, 4) GO(512, 8) GO(256, 8) GO(1024, 8) GO(256, 16)
#undef GO
        const double bytes = (double)n16 * 16.0 * bytes_per_elem;
        printf("%-4s threads %4d x %d blocks/CU, %2d x 16 B per lane: %4d KB in flight per CU  %8.1f us  %6.2f TB/s\n", name, c.T,
               c.bpc, c.U, c.T * c.U * c.bpc * 16 / 1024, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << 30;
    const size_t n16 = bytes / 16;
    f4 *src, *dst;
    float* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 0, bytes)); CK(hipMemset(dst, 0, bytes));
    const char* only = argc > 1 ? argv[1] : "";
#define S(M, name, f) if (!*only || !strcmp(only, name)) sweep<M>(name, src, dst, sink, n16, f);
    S(M_R, "r", 1.0) S(M_RN, "rn", 1.0) S(M_RL, "rl", 1.0) S(M_RLN, "rln", 1.0) S(M_W, "w", 1.0) S(M_WN, "wn", 1.0)
    S(M_C, "c", 2.0) S(M_CN, "cn", 2.0) S(M_RW, "rw", 2.0)
    return 0;
}
