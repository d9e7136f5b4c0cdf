// Second reproducer for the two-processes-on-one-device fault: does a wavefront ever LOAD something that its own
// process did not put there?  Both processes allocate the same buffers in the same order (so that they sit at the same virtual
// addresses), a one-wave kernel rewrites a 64-word table every round with a pattern that names the process and the round
// (like ro_update_kernel rewriting the search state), and a device-filling kernel reads the table the way ro_particles_kernel
// reads `pst` / `state`: every lane the same address.  A word that is not this round's pattern of this process is classified:
// the OTHER process's pattern (cross-process aliasing in a cache), an EARLIER round of this process (a stale line), or garbage;
// lanes that differ from lane 0 are counted as in cwsr_trans.hip.
//   vm_alias <tag 0|1> <seconds>       run one per process, at the same time
//   hipcc --offload-arch=gfx950 -O3 -o vm_alias vm_alias.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

__device__ __host__ inline unsigned pattern(unsigned tag, unsigned round, unsigned t) { return (tag << 30) | ((round & 0xfffffu) << 8) | t; }

__global__ void writer(unsigned* table, unsigned tag, unsigned round) { table[threadIdx.x] = pattern(tag, round, threadIdx.x); }

__global__ __launch_bounds__(256) void reader(const unsigned* __restrict__ table, unsigned tag, unsigned round, unsigned reps,
                                              unsigned long long* out) {
    unsigned foreign = 0, stale = 0, garbage = 0, lanes = 0;
    unsigned long long mask_or = 0ull;
    unsigned z = 0;
    asm volatile("" : "+v"(z));                    // a vector register: the loads below are vector loads of a wave-uniform address
    for (unsigned r = 0; r < reps; ++r) {
#pragma unroll 1
        for (unsigned k = 0; k < 18; ++k) {
            const unsigned idx = (k * 7 + r) & 63u;
            const unsigned v = table[idx + z];
            const unsigned want = pattern(tag, round, idx);
            const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            const unsigned long long diff = __ballot(v != first);
            if (diff) ++lanes, mask_or |= diff;
            if (v != want) {
                if ((v >> 30) != tag && (v & 0xffu) == idx) ++foreign;
                else if ((v >> 30) == tag && (v & 0xffu) == idx) ++stale;
                else ++garbage;
            }
        }
    }
    if (foreign | stale | garbage | lanes) {
        atomicAdd(out, (unsigned long long)foreign);
        atomicAdd(out + 1, (unsigned long long)stale);
        atomicAdd(out + 2, (unsigned long long)garbage);
        if ((threadIdx.x & 63) == 0) atomicAdd(out + 3, (unsigned long long)lanes);
        atomicOr(out + 4, mask_or);
    }
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)

int main(int argc, char** argv) {
    const unsigned tag = argc > 1 ? (unsigned)atoi(argv[1]) : 0u;
    const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
    unsigned* table;
    unsigned long long* out;
    CHECK(hipMalloc(&table, 64 * sizeof(unsigned)));
    CHECK(hipMalloc(&out, 8 * sizeof(unsigned long long)));
    CHECK(hipMemset(out, 0, 8 * sizeof(unsigned long long)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;
    unsigned round = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 64; ++k, ++round) {          // 64 rounds queued back to back, like the rounds of a frame
            hipLaunchKernelGGL(writer, dim3(1), dim3(64), 0, 0, table, tag, round);
            hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, 0, table, tag, round, 40u, out);
        }
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long h[8];
    CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("tag %u pid %d table at %p: %u rounds: loads that returned the OTHER process's pattern %llu, an EARLIER round of this process %llu, "
           "garbage %llu; wavefront-loads whose lanes differ from lane 0: %llu (OR of lane masks 0x%016llx)\n",
           tag, (int)getpid(), (void*)table, round, h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
