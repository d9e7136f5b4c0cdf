// Why do the decoder kernels' record stores run at ~3 TB/s when a plain write stream does 6?  Their pattern: 2048 waves, each
// writing ITS OWN tile record piece by piece (1 KB per wave instruction, 32 pieces at 1 KB steps inside a record), records
// `stride` bytes apart -- so at any moment the device writes ~2048 lines that are `stride` apart.  This kernel reproduces
// the pattern without any compute and varies the stride / the piece order.
//   order 0: piece q of every tile at about the same time (the kernels' order)     order 1: wave w starts at piece (w * 5) % 32
// build: hipcc --offload-arch=gfx950 -O3 record_store.hip -o record_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(512) void k(char* base, size_t stride, int first_piece, int n_pieces, int n_tiles, int order, int spin) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 8 + (threadIdx.x >> 6);
    const f4 v = {1.f, 2.f, 3.f, (float)w};
    for (int tile = w; tile < n_tiles; tile += gridDim.x * 8) {
        char* rec = base + (size_t)tile * stride;
        for (int i = 0; i < n_pieces; ++i) {
            const int q = first_piece + (order ? (i + w * 5) % n_pieces : i);
            f4* p = reinterpret_cast<f4*>(rec + (size_t)q * 1024) + lane;
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);      // (compute between the pieces)
        }
    }
}

int main() {
    const int n_tiles = 8192;
    char* buf;
    const size_t cap = (size_t)n_tiles * 72 * 1024;
    CK(hipMalloc(&buf, cap));
    CK(hipMemset(buf, 0, cap));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t strides[] = {48 * 1024, 32 * 1024, 64 * 1024, 49 * 1024, 48 * 1024 + 256, 52 * 1024, 36 * 1024, 33 * 1024};
    for (int nt = 0; nt < 2; ++nt)
        for (size_t stride : strides)
            for (int order = 0; order < 2; ++order)
                for (int spin : {0, 4}) {
                    const int first = stride >= 48 * 1024 ? 16 : 0, np = 32;
                    auto go = [&]() {
                        if (nt) k<1><<<256, 512>>>(buf, stride, first, np, n_tiles, order, spin);
                        else k<0><<<256, 512>>>(buf, stride, first, np, n_tiles, order, spin);
                    };
                    go();
                    CK(hipEventRecord(a));
                    for (int r = 0; r < 5; ++r) go();
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms;
                    CK(hipEventElapsedTime(&ms, a, b));
                    const double bytes = (double)n_tiles * np * 1024;
                    printf("%s stride %6zu B order %d spin %d: %7.1f us  %5.2f TB/s\n", nt ? "nt     " : "default", stride, order, spin,
                           ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
                }
    return 0;
}
