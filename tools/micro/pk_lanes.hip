// Stand-alone reproducer for the "lanes 48..63" fault: a kernel shaped like ro_particles_kernel -- one wavefront per
// particle, four global loads per lattice point, the rigid transform as hipcc packs it (v_pk_mul_f32 / v_pk_add_f32), three fp64
// normalisations, a 12-byte store -- launched over and over on the same inputs; every launch's output is compared with the first
// launch's on the device.  With one process on the GPU no launch ever differs.  Beside a SECOND PROCESS whose kernels hold the CUs'
// LDS and keep the matrix cores busy (this program's `neighbour` mode, or tools/ba_load.py) some launches differ: 16 values, the
// x coordinate of lanes 48..63 of one wavefront-iteration, short of exactly the product that hipcc placed in the HIGH half of a
// v_pk_mul_f32.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_lanes pk_lanes.hip
//   ./pk_lanes neighbour 60 &            # second process: persistent 160 KB-LDS MFMA kernels for 60 s
//   ./pk_lanes packed 20                 # 20 s of launches, packed multiplies      -> "N of M launches differ"
//   ./pk_lanes single 20                 # the same with nine single v_mul_f32      -> 0
//   ./pk_lanes asm 20                    # hipcc's sequence of the library kernel verbatim (a crossed v_pk_add_f32)
//   ./pk_lanes asmfix 20                 # ... with that one instruction replaced by two v_add_f32
//   ./pk_lanes asmmul | asmfma | asmmov  # the same sum through a v_pk_mul_f32 / v_pk_fma_f32 crossed on the FIRST source / a v_pk_mov_b32 half swap
//   ./pk_lanes asmadd0 | asmmul1 | asmfma1 | asmfma2 | asmnop   # the add crossed on its first source; multiply / fma crossed on the second
//                                          (fma: third) source; the verbatim add 32 idle cycles behind its producers
//   ./pk_lanes packed 20 inproc[:kind]   # neighbour kernels on a second stream of THIS process instead of a second process
//   ./pk_lanes neighbour 60 kind:6       # (kinds: see neighbour<KIND>)
//   ./pk_lanes lib:<path to a libmipsf_hip build> 20     # the library's own mipsf_ro_particles (point-major) on the same inputs
//                                          (tools/micro/libv_ropk1.so = the packed build, mipsfusion_amd/libmipsf_hip.so = the product)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include "mipsf.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Norm { double sub[3], div[3], nf; };

// MODE 0: nine single multiplies; 1: as hipcc packs the source below (no crossed operand in this file's build); 2: the instruction
// sequence hipcc emits inside the library's kernel, verbatim (its packed add takes the HIGH half of a source for the LOW result:
// v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]); 3: the same with that one instruction replaced by two v_add_f32
template <int MODE>
__global__ __launch_bounds__(256) void particles(const float* __restrict__ pose, const float* __restrict__ dirs,
                                                 const float* __restrict__ depth, Norm nc, float* __restrict__ xn,
                                                 unsigned P, unsigned n) {
    const unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) / 64u, lane = threadIdx.x & 63u;
    if (p >= P) return;
    float a[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(pose[12 * p + k])));
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(pose[12 * p + 9 + k])));
    for (unsigned i = lane; i < n; i += 64u) {
        const float d = depth[i];
        const float c0 = dirs[3 * i] * d, c1 = dirs[3 * i + 1] * d, c2 = dirs[3 * i + 2] * d;
        float w0, w1, w2;
        if (MODE == 1) {
            w0 = ((a[0] * c0 + a[1] * c1) + a[2] * c2) + t[0];
            w1 = ((a[3] * c0 + a[4] * c1) + a[5] * c2) + t[1];
            w2 = ((a[6] * c0 + a[7] * c1) + a[8] * c2) + t[2];
        } else if (MODE >= 2) {
            const float r0 = dirs[3 * i], r1 = dirs[3 * i + 1], r2 = dirs[3 * i + 2];
#define PK_S(v) (int)__float_as_uint(v)
#define PK_HEAD                                                                                                                \
            "v_mov_b32 v122, 1.0\n v_mov_b32 v123, 1.0\n"                                                                      \
            "s_mov_b32 s68, %[a1]\n s_mov_b32 s69, %[a3]\n s_mov_b32 s62, %[a4]\n s_mov_b32 s63, %[a0]\n"                     \
            "s_mov_b32 s70, %[a2]\n s_mov_b32 s71, %[a5]\n s_mov_b32 s73, %[a6]\n s_mov_b32 s82, %[a7]\n"                     \
            "s_mov_b32 s83, %[a8]\n s_mov_b32 s74, %[t0]\n s_mov_b32 s75, %[t1]\n s_mov_b32 s84, %[t2]\n"                     \
            "v_mov_b32 v112, %[d]\n v_mov_b32 v113, %[r2]\n v_mov_b32 v114, %[r1]\n v_mov_b32 v115, %[r0]\n v_mov_b32 v117, 0\n" \
            "v_mul_f32_e32 v116, v112, v113\n"                                                                                  \
            "v_pk_mul_f32 v[112:113], v[112:113], v[114:115] op_sel_hi:[0,1]\n"                                                 \
            "v_pk_mul_f32 v[114:115], s[68:69], v[112:113]\n"                                                                   \
            "v_pk_mul_f32 v[118:119], s[62:63], v[112:113]\n"                                                                   \
            "v_mul_f32_e32 v113, s73, v113\n"
#define PK_TAIL                                                                                                                \
            "v_pk_mul_f32 v[118:119], s[70:71], v[116:117] op_sel_hi:[1,0]\n"                                                   \
            "v_mul_f32_e32 v112, s82, v112\n"                                                                                   \
            "v_pk_add_f32 v[114:115], v[114:115], v[118:119]\n"                                                                 \
            "v_add_f32_e32 v112, v113, v112\n"                                                                                  \
            "v_mul_f32_e32 v113, s83, v116\n"                                                                                   \
            "v_pk_add_f32 v[114:115], s[74:75], v[114:115]\n"                                                                   \
            "v_add_f32_e32 v112, v112, v113\n"                                                                                  \
            "v_add_f32_e32 v130, s84, v112\n"                                                                                   \
            "v_mov_b32 %[o0], v114\n v_mov_b32 %[o1], v115\n v_mov_b32 %[o2], v130\n"
#define PK_OPS                                                                                                                 \
            : [o0] "=v"(w0), [o1] "=v"(w1), [o2] "=v"(w2)                                                                      \
            : [d] "v"(d), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [a0] "s"(PK_S(a[0])), [a1] "s"(PK_S(a[1])), [a2] "s"(PK_S(a[2])), \
              [a3] "s"(PK_S(a[3])), [a4] "s"(PK_S(a[4])), [a5] "s"(PK_S(a[5])), [a6] "s"(PK_S(a[6])), [a7] "s"(PK_S(a[7])),       \
              [a8] "s"(PK_S(a[8])), [t0] "s"(PK_S(t[0])), [t1] "s"(PK_S(t[1])), [t2] "s"(PK_S(t[2]))                              \
            : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v130", "s62", "s63", "s68", "s69", "s70", "s71",  \
              "s73", "s74", "s75", "s82", "s83", "s84"
            if (MODE == 2)
                asm volatile(PK_HEAD "v_pk_add_f32 v[114:115], v[114:115], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n" PK_TAIL PK_OPS);
            else if (MODE == 3)
                asm volatile(PK_HEAD "v_add_f32_e32 v114, v114, v119\n v_add_f32_e32 v115, v115, v118\n" PK_TAIL PK_OPS);
            // the same sum through OTHER packed instructions that cross halves (x * 1 and fma(x, 1, y) are exact), each followed /
            // replaced so that the result is bit for bit the reference's when nothing goes wrong:
            else if (MODE == 4)      // a packed MULTIPLY whose low result takes the high half of its FIRST source, then an uncrossed add
                asm volatile(PK_HEAD "v_pk_mul_f32 v[120:121], v[118:119], v[122:123] op_sel:[1,0] op_sel_hi:[0,1]\n"
                             "v_pk_add_f32 v[114:115], v[114:115], v[120:121]\n" PK_TAIL PK_OPS);
            else if (MODE == 5)      // a packed FMA with the crossed operand
                asm volatile(PK_HEAD "v_pk_fma_f32 v[114:115], v[118:119], v[122:123], v[114:115] op_sel:[1,0,0] op_sel_hi:[0,1,1]\n" PK_TAIL PK_OPS);
            else if (MODE == 6)      // v_pk_mov_b32 swapping the halves (hipcc's shuffle), then an uncrossed add
                asm volatile(PK_HEAD "v_pk_mov_b32 v[120:121], v[118:119], v[118:119] op_sel:[1,0]\n"
                             "v_pk_add_f32 v[114:115], v[114:115], v[120:121]\n" PK_TAIL PK_OPS);
            else if (MODE == 7)      // the crossed add with the operands exchanged: the crossing on the FIRST source
                asm volatile(PK_HEAD "v_pk_add_f32 v[114:115], v[118:119], v[114:115] op_sel:[1,0] op_sel_hi:[0,1]\n" PK_TAIL PK_OPS);
            else if (MODE == 8)      // a packed multiply with the crossing on the SECOND source, then an uncrossed add
                asm volatile(PK_HEAD "v_pk_mul_f32 v[120:121], v[122:123], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n"
                             "v_pk_add_f32 v[114:115], v[114:115], v[120:121]\n" PK_TAIL PK_OPS);
            else if (MODE == 9)      // a packed FMA with the crossing on the SECOND source
                asm volatile(PK_HEAD "v_pk_fma_f32 v[114:115], v[122:123], v[118:119], v[114:115] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n" PK_TAIL PK_OPS);
            else if (MODE == 10)     // ... on the THIRD source (the addend)
                asm volatile(PK_HEAD "v_pk_fma_f32 v[114:115], v[114:115], v[122:123], v[118:119] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n" PK_TAIL PK_OPS);
            else                     // MODE 11: the verbatim crossed add, 32 idle cycles behind the instructions that wrote its sources
                asm volatile(PK_HEAD "s_nop 15\n s_nop 15\n v_pk_add_f32 v[114:115], v[114:115], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n" PK_TAIL PK_OPS);
        } else {
            float m[9];
#define MUL(k, c) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m[k]) : "s"(a[k]), "v"(c))
            MUL(0, c0); MUL(1, c1); MUL(2, c2); MUL(3, c0); MUL(4, c1); MUL(5, c2); MUL(6, c0); MUL(7, c1); MUL(8, c2);
#undef MUL
            w0 = ((m[0] + m[1]) + m[2]) + t[0], w1 = ((m[3] + m[4]) + m[5]) + t[1], w2 = ((m[6] + m[7]) + m[8]) + t[2];
        }
        float* o = xn + 3 * ((size_t)i * P + p);
        o[0] = (float)((((double)w0 - nc.sub[0]) / nc.div[0]) / nc.nf);
        o[1] = (float)((((double)w1 - nc.sub[1]) / nc.div[1]) / nc.nf);
        o[2] = (float)((((double)w2 - nc.sub[2]) / nc.div[2]) / nc.nf);
    }
}

// how many values differ from the reference output, and which (component, lane-of-wavefront) they are
__global__ void compare(const float* __restrict__ got, const float* __restrict__ ref, size_t count, unsigned P,
                        unsigned long long* __restrict__ out) {
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < count; k += (size_t)gridDim.x * blockDim.x)
        if (__float_as_uint(got[k]) != __float_as_uint(ref[k])) {
            const unsigned comp = (unsigned)(k % 3), point = (unsigned)((k / 3) / P);
            atomicAdd(out, 1ull);
            atomicOr(out + 1, 1ull << (point & 63u));
            atomicOr(out + 2, 1ull << comp);
        }
}

typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// the neighbour: one workgroup per CU, the CU's whole LDS, two wavefronts per SIMD.  KIND selects the instruction mix of its loop:
//   0 LDS reads feeding a chain of bf16 MFMAs      1 v_cvt_pk_bf16_f32      2 v_pk_fma_f32      3 v_pk_add_f32 with a crossed op_sel
//   4 v_sin_f32      5 ds_bpermute_b32      6 MFMA + cvt_pk + pk_fma (the decoder's mix)      7 ds_read/ds_write_b128 + s_barrier
//   8 global loads and stores      9 straight-line code twice the size of the instruction cache
//   10 back-to-back MFMAs on register operands (the power the decoder kernels draw)      11 the same on ~250 VGPRs per wave      12 the same with fp32-input MFMAs
//   13 / 14 / 15 the same with f16 32x32x16 / bf16 16x16x32 / i8 32x32x32
template <int KIND>
__global__ __launch_bounds__(512) void neighbour(unsigned iters, float* __restrict__ sink) {
    extern __shared__ bf8 tile[];
    const unsigned T = 160 * 1024 / 16;
    for (unsigned k = threadIdx.x; k < T; k += 512) {
        bf8 v;
        for (int j = 0; j < 8; ++j) v[j] = (short)(0x3c00 + ((k + j) & 63));
        tile[k] = v;
    }
    __syncthreads();
    f16v acc = {};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f, c = 0.25f, e = 2.0f;
    for (unsigned i = 0; i < iters; ++i) {
        if (KIND == 0 || KIND == 6) {
            const bf8 x = tile[(threadIdx.x + 61u * i) % T], y = tile[(threadIdx.x * 3u + 17u * i) % T];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc, 0, 0, 0);
        }
        if (KIND == 1 || KIND == 6)
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %1, %0, %2\n v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %1, %0, %2"
                         : "+v"(a), "+v"(b) : "v"(c));
        if (KIND == 2 || KIND == 6) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 x = {a, b}, y = {c, e};
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1"
                         : "+v"(x) : "v"(y));
            a = x[0] * 1e-30f + 1.0f, b = x[1] * 1e-30f + 0.5f;
        }
        if (KIND == 3) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 x = {a, b}, y = {c, e};
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n"
                         "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]"
                         : "+v"(x) : "v"(y));
            a = x[0] * 1e-30f + 1.0f, b = x[1] * 1e-30f + 0.5f;
        }
        if (KIND == 4) asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %0, %0\n v_sin_f32 %1, %1" : "+v"(a), "+v"(b));
        if (KIND == 5) {
            int idx = (int)((threadIdx.x * 4u + 128u) & 255u), v = (int)__float_as_uint(a);
            asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(idx));
            a = __uint_as_float((unsigned)v);
        }
        if (KIND == 7) {
            bf8 x = tile[(threadIdx.x + 61u * i) % T];
            __syncthreads();
            tile[(threadIdx.x * 5u + 13u * i) % T] = x;
            __syncthreads();
        }
        if (KIND == 10) {     // the matrix cores flat out: four independent accumulators, operands with busy bit patterns held in registers
            bf8 x, y;
            for (int j = 0; j < 8; ++j) x[j] = (short)(0x3f80 ^ ((threadIdx.x * 2654435761u) >> (j + 3))), y[j] = (short)(0x3f00 ^ ((threadIdx.x * 40503u) >> j));
            f16v acc1 = acc, acc2 = acc, acc3 = acc;
            for (unsigned q = 0; q < 64; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, acc3, 0, 0, 0);
            }
            for (int j = 0; j < 16; ++j) acc[j] = (acc[j] + acc1[j] + acc2[j] + acc3[j]) * 1e-30f;
        }
        if (KIND == 13 || KIND == 14 || KIND == 15) {     // like 10 with other 16-bit / 8-bit matrix instructions: 13 f16 32x32x16, 14 bf16 16x16x32,
            typedef _Float16 hf8 __attribute__((ext_vector_type(8)));       // 15 i8 32x32x32
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef int i4v __attribute__((ext_vector_type(4)));
            typedef int i16v __attribute__((ext_vector_type(16)));
            bf8 x, y;
            for (int j = 0; j < 8; ++j) x[j] = (short)(0x3c00 ^ ((threadIdx.x * 2654435761u) >> (j + 3)) & 0x3ff), y[j] = (short)(0x3800 ^ ((threadIdx.x * 40503u) >> j) & 0x3ff);
            if (KIND == 13) {
                const hf8 hx = __builtin_bit_cast(hf8, x), hy = __builtin_bit_cast(hf8, y);
                f16v a1 = acc, a2 = acc, a3 = acc;
                for (unsigned q = 0; q < 64; ++q) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, acc, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hx, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hx, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hy, a3, 0, 0, 0);
                }
                for (int j = 0; j < 16; ++j) acc[j] = (acc[j] + a1[j] + a2[j] + a3[j]) * 1e-30f;
            } else if (KIND == 14) {
                f4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
                for (unsigned q = 0; q < 128; ++q) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, c3, 0, 0, 0);
                }
                for (int j = 0; j < 4; ++j) acc[j] = (c0[j] + c1[j] + c2[j] + c3[j]) * 1e-30f;
            } else {
                const i4v ix = __builtin_bit_cast(i4v, x), iy = __builtin_bit_cast(i4v, y);
                i16v d0 = {}, d1 = {}, d2 = {}, d3 = {};
                for (unsigned q = 0; q < 64; ++q) {
                    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ix, iy, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(iy, ix, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ix, ix, d2, 0, 0, 0);
                    d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(iy, iy, d3, 0, 0, 0);
                }
                for (int j = 0; j < 16; ++j) acc[j] = (float)(d0[j] + d1[j] + d2[j] + d3[j]) * 1e-30f;
            }
        }
        if (KIND == 12) {     // like 10 on the fp32-input matrix instruction (v_mfma_f32_32x32x2_f32: the round-1 decoder kernels)
            const float xa = 1.0f + threadIdx.x * 1e-3f, xb = 0.5f - threadIdx.x * 1e-4f;
            f16v acc1 = acc, acc2 = acc, acc3 = acc;
            for (unsigned q = 0; q < 64; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xb, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xb, xa, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xa, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xb, xb, acc3, 0, 0, 0);
            }
            for (int j = 0; j < 16; ++j) acc[j] = (acc[j] + acc1[j] + acc2[j] + acc3[j]) * 1e-30f;
        }
        if (KIND == 11) {     // like 10 with fourteen accumulators: ~250 VGPRs per wave, two waves fill a SIMD's register file (as the
            bf8 x, y;         // decoder's kernels do): a victim wave only fits where one of them has ended
            for (int j = 0; j < 8; ++j) x[j] = (short)(0x3f80 ^ ((threadIdx.x * 2654435761u) >> (j + 3))), y[j] = (short)(0x3f00 ^ ((threadIdx.x * 40503u) >> j));
            f16v ac[14];
#pragma unroll
            for (int q = 0; q < 14; ++q) ac[q] = acc;
            for (unsigned q = 0; q < 24; ++q) {
#pragma unroll
                for (int u = 0; u < 14; ++u) ac[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u & 1 ? y : x, u & 2 ? x : y, ac[u], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 14; ++q)
                for (int j = 0; j < 16; ++j) acc[j] += ac[q][j] * 1e-30f;
        }
        if (KIND == 9) {      // 128 KB of straight-line code, run through over and over: twice the instruction cache the CUs share
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
            R16(R256(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %0, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %0, %2" : "+v"(a), "+v"(b) : "v"(c));))
        }
        if (KIND == 8) {
            float* q = sink + 1024 + ((threadIdx.x + blockIdx.x * 512u + i * 7919u) & ((1u << 22) - 1u));
            *q = *q + 1.0f;
        }
    }
    float s = a + b;
    for (int j = 0; j < 16; ++j) s += acc[j];
    if (s == 12345.f) sink[threadIdx.x] = s;
}

typedef void (*neighbour_fn)(unsigned, float*);
static neighbour_fn neighbour_of(int kind) {
    switch (kind) {
        case 1: return neighbour<1>; case 2: return neighbour<2>; case 3: return neighbour<3>; case 4: return neighbour<4>;
        case 5: return neighbour<5>; case 6: return neighbour<6>; case 7: return neighbour<7>; case 8: return neighbour<8>;
        case 9: return neighbour<9>; case 10: return neighbour<10>; case 11: return neighbour<11>; case 12: return neighbour<12>;
        case 13: return neighbour<13>; case 14: return neighbour<14>; case 15: return neighbour<15>;
        default: return neighbour<0>;
    }
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "packed";
    const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
    const bool inproc = argc > 3 && !strncmp(argv[3], "inproc", 6);
    const int kind = argc > 3 && strchr(argv[3], ':') ? atoi(strchr(argv[3], ':') + 1) : 0;       // "inproc:6", "kind:6" (neighbour mode)
    // "inproc:6,7": two kinds launched alternately, each for ~150 us (a forward and a backward kernel taking turns)
    const int kind2 = argc > 3 && strchr(argv[3], ',') ? atoi(strchr(argv[3], ',') + 1) : -1;
    const neighbour_fn nb = neighbour_of(kind), nb2 = neighbour_of(kind2 < 0 ? kind : kind2);
    const unsigned alt_iters = argc > 4 ? (unsigned)atoi(argv[4]) : 3000u;
    const unsigned nb_iters = kind == 7 ? 3000u : kind == 8 ? 4000u : kind == 9 ? 10u : kind == 10 ? 100u : kind == 11 ? 60u : kind == 12 ? 50u : kind >= 13 ? 100u : 20000u;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    float* sink;
    CHECK(hipMalloc(&sink, 4096 + (sizeof(float) << 22)));
    CHECK(hipFuncSetAttribute((const void*)nb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)nb2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (!strcmp(mode, "neighbour")) {
        unsigned launches = 0;
        while (elapsed() < seconds) {
            for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(nb, dim3(prop.multiProcessorCount), dim3(512), 160 * 1024, 0, nb_iters, sink);
            CHECK(hipDeviceSynchronize());
            launches += 8;
        }
        printf("neighbour kind %d pid %d: %u launches in %.1f s\n", kind, (int)getpid(), launches, elapsed());
        return 0;
    }
    const int vmode = !strcmp(mode, "packed") ? 1 : !strcmp(mode, "asm") ? 2 : !strcmp(mode, "asmfix") ? 3 : !strcmp(mode, "asmmul") ? 4 :
                      !strcmp(mode, "asmfma") ? 5 : !strcmp(mode, "asmmov") ? 6 : !strcmp(mode, "asmadd0") ? 7 : !strcmp(mode, "asmmul1") ? 8 :
                      !strcmp(mode, "asmfma1") ? 9 : !strcmp(mode, "asmfma2") ? 10 : !strcmp(mode, "asmnop") ? 11 : 0;
    typedef int (*ro_fn)(const float*, const float*, const float*, const float*, const mipsf_render_cfg*, float*, float*, uint32_t,
                         uint32_t, int, void*);
    ro_fn ro = nullptr;
    if (!strncmp(mode, "lib:", 4)) {
        void* h = dlopen(mode + 4, RTLD_NOW);
        if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
        ro = (ro_fn)dlsym(h, "mipsf_ro_particles");
        if (!ro) { fprintf(stderr, "no mipsf_ro_particles in %s\n", mode + 4); return 2; }
    }
    const unsigned P = 2000, n = 384;
    std::vector<float> pose(12 * P), dirs(3 * n), depth(n);
    srand(1);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
    for (unsigned p = 0; p < P; ++p) {
        const float base[12] = {0.962f, -0.059f, 0.266f, 0.011f, 0.984f, 0.178f, -0.272f, -0.169f, 0.947f, 1.168f, 3.796f, 0.946f};
        for (int k = 0; k < 12; ++k) pose[12 * p + k] = base[k] + 0.02f * (rnd() - 0.5f);
    }
    for (unsigned i = 0; i < n; ++i) {
        dirs[3 * i] = rnd() - 0.5f, dirs[3 * i + 1] = 0.8f * (rnd() - 0.5f), dirs[3 * i + 2] = 1.0f;
        depth[i] = 0.8f + 2.2f * rnd();
    }
    float *d_pose, *d_dirs, *d_depth, *d_xn, *d_ref;
    unsigned long long* d_out;
    const size_t count = (size_t)3 * P * n;
    CHECK(hipMalloc(&d_pose, pose.size() * 4));
    CHECK(hipMalloc(&d_dirs, dirs.size() * 4));
    CHECK(hipMalloc(&d_depth, depth.size() * 4));
    CHECK(hipMalloc(&d_xn, count * 4));
    CHECK(hipMalloc(&d_ref, count * 4));
    CHECK(hipMalloc(&d_out, 32));
    CHECK(hipMemcpy(d_pose, pose.data(), pose.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dirs, dirs.data(), dirs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice));
    const Norm nc = {{-0.6, 0.5, -1.15}, {3.5500000000000003, 6.55, 4.199999999999999}, 1.0};
    hipStream_t side;
    hipStream_t vs;            // victim stream.  Both non-blocking: the null stream would serialise with `side` (it did, until this fix)
    CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&vs, hipStreamNonBlocking));
    // library mode: a 6-D particle template, the search state (rotation, translation, search size) and the render configuration
    std::vector<float> pst(6 * P), state(MIPSF_RO_STATE_FLOATS, 0.f);
    for (auto& v : pst) v = 2.f * rnd() - 1.f;
    for (int k = 0; k < 6; ++k) pst[k] = 0.f;
    const float base[12] = {0.962f, -0.059f, 0.266f, 0.011f, 0.984f, 0.178f, -0.272f, -0.169f, 0.947f, 1.168f, 3.796f, 0.946f};
    for (int k = 0; k < 12; ++k) state[k] = base[k];
    for (int k = 0; k < 6; ++k) state[12 + k] = 0.02f;
    mipsf_render_cfg rc;
    memset(&rc, 0, sizeof rc);
    rc.n_uniform = 1, rc.use_bound = 0, rc.norm_factor = 1.0;
    rc.half_len[0] = 0.6, rc.half_len[1] = 3.275, rc.half_len[2] = 2.1;
    float *d_pst, *d_state, *d_pst7;
    CHECK(hipMalloc(&d_pst, pst.size() * 4));
    CHECK(hipMalloc(&d_state, state.size() * 4));
    CHECK(hipMalloc(&d_pst7, 7 * P * 4));
    CHECK(hipMemcpy(d_pst, pst.data(), pst.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_state, state.data(), state.size() * 4, hipMemcpyHostToDevice));
    auto launch = [&](float* dst) {
        if (ro) {
            if (ro(d_pst, d_state, d_dirs, d_depth, &rc, dst, d_pst7, P, n, 1, (void*)vs) != 0) { fprintf(stderr, "mipsf_ro_particles failed\n"); exit(2); }
            return;
        }
        if (vmode == 1) hipLaunchKernelGGL(particles<1>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 2) hipLaunchKernelGGL(particles<2>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 3) hipLaunchKernelGGL(particles<3>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 4) hipLaunchKernelGGL(particles<4>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 5) hipLaunchKernelGGL(particles<5>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 6) hipLaunchKernelGGL(particles<6>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 7) hipLaunchKernelGGL(particles<7>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 8) hipLaunchKernelGGL(particles<8>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 9) hipLaunchKernelGGL(particles<9>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 10) hipLaunchKernelGGL(particles<10>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else if (vmode == 11) hipLaunchKernelGGL(particles<11>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
        else hipLaunchKernelGGL(particles<0>, dim3(P / 4), dim3(256), 0, vs, d_pose, d_dirs, d_depth, nc, dst, P, n);
    };
    for (int tries = 0;; ++tries) {      // the reference output: one that two consecutive launches agree on
        launch(d_ref);
        launch(d_xn);
        CHECK(hipMemsetAsync(d_out, 0, 32, vs));
        hipLaunchKernelGGL(compare, dim3(1024), dim3(256), 0, vs, d_xn, d_ref, count, P, d_out);
        unsigned long long h[4];
        CHECK(hipMemcpyAsync(h, d_out, 32, hipMemcpyDeviceToHost, vs));
            CHECK(hipStreamSynchronize(vs));
        if (!h[0]) break;
        if (tries == 20) { fprintf(stderr, "no two consecutive launches agree\n"); return 3; }
    }
    unsigned launches = 0, differing = 0, nb_launched = 0;
    hipEvent_t nb_done[2];
    CHECK(hipEventCreate(&nb_done[0]));
    CHECK(hipEventCreate(&nb_done[1]));
    unsigned long long values = 0, lanes = 0, comps = 0;
    while (elapsed() < seconds) {
        // keep two groups of neighbour kernels in flight on the side stream, no more
        if (inproc && (nb_launched < 2 || hipEventQuery(nb_done[nb_launched & 1]) == hipSuccess)) {
            if (kind2 < 0) hipLaunchKernelGGL(nb, dim3(prop.multiProcessorCount), dim3(512), 160 * 1024, side, nb_iters, sink);
            else
                for (int k = 0; k < 2; ++k) {
                    hipLaunchKernelGGL(nb, dim3(prop.multiProcessorCount), dim3(512), 160 * 1024, side, alt_iters, sink);
                    hipLaunchKernelGGL(nb2, dim3(prop.multiProcessorCount), dim3(512), 64 * 1024, side, alt_iters, sink);
                }
            CHECK(hipEventRecord(nb_done[nb_launched & 1], side));
            ++nb_launched;
        }
        for (int k = 0; k < 16; ++k) {
            CHECK(hipMemsetAsync(d_out, 0, 32, vs));
            launch(d_xn);
            hipLaunchKernelGGL(compare, dim3(1024), dim3(256), 0, vs, d_xn, d_ref, count, P, d_out);
            unsigned long long h[4];
            CHECK(hipMemcpyAsync(h, d_out, 32, hipMemcpyDeviceToHost, vs));
            CHECK(hipStreamSynchronize(vs));
            ++launches;
            if (h[0]) ++differing, values += h[0], lanes |= h[1], comps |= h[2];
        }
    }
    CHECK(hipDeviceSynchronize());
    if (inproc) printf("[neighbour kind %d%s%s] ", kind, kind2 >= 0 ? " alternating with " : "", kind2 >= 0 ? strchr(argv[3], ',') + 1 : "");
    printf("%s%s pid %d: %u of %u launches differ from the first launch (%llu values; lanes 0x%016llx, components 0x%llx) in %.1f s\n",
           mode, inproc ? " (neighbour kernels in this process)" : "", (int)getpid(), differing, launches, values, lanes, comps, elapsed());
    return 0;
}
