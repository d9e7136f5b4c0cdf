// Shared host/device helpers for libmipsf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mipsf.h"

#define MIPSF_WAVE 64

namespace mipsf {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Per-device host-side caches (CU count, opted-in dynamic-LDS sizes): function attributes and device properties
// belong to ONE device, a process may drive several.
constexpr int MAX_DEVICES = 64;
inline int device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) return 0;
    return d;
}
// CU count of the calling thread's current device (cached per device); <= 0 on error
int device_cus();

// buffer sizes behind mipsf_buffer_size (capi.hip); each is defined next to the kernels that use the buffer
uint64_t hashgrid_bwd_scratch_floats(const mipsf_grid_meta* meta, uint32_t M, int need_dx);
uint64_t hashgrid_counter_words(const mipsf_grid_meta* meta);
uint64_t decoder_packed_floats();
uint64_t decoder_saved_floats(uint32_t M);
uint64_t decoder_dact_floats(uint32_t M);
uint64_t decoder_wgrad_partial_floats();
uint64_t decoder_packed16_floats(int precision);
uint64_t decoder_tile_words(uint32_t M);
uint64_t render_partial_floats(uint32_t N);
uint64_t place_pose_scratch_floats(uint32_t F, uint32_t K, uint32_t N);
uint64_t pose_rays_scratch_floats(uint32_t F, uint32_t K, uint32_t N);

#define MIPSF_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::mipsf::set_error(__VA_ARGS__);     \
            return 1;                            \
        }                                        \
    } while (0)

// level table handed to kernels by value
struct GridLevels {
    uint32_t n_levels;
    uint32_t offsets[MIPSF_MAX_LEVELS + 1];
    uint32_t res[MIPSF_MAX_LEVELS];
    float scale[MIPSF_MAX_LEVELS];
};

// normalisation constants handed to kernels by value (fp64 on purpose: mipsfusion.py:94-96 builds
// the bounding box as a float64 tensor, so scene_rep.py:140/142 computes in float64)
struct NormCfg {
    double sub[3];   // x -> (x - sub) / div / norm_factor   (use_bound: sub = bmin, div = bmax - bmin;
    double div[3];   //                                       else:      sub = -L,   div = 2L)
    double norm_factor;
};

inline NormCfg make_norm(const mipsf_render_cfg& c) {
    NormCfg n;
    for (int d = 0; d < 3; ++d) {
        if (c.use_bound) {
            n.sub[d] = c.bound_min[d];
            n.div[d] = c.bound_max[d] - c.bound_min[d];
        } else {
            n.sub[d] = -c.half_len[d];
            n.div[d] = 2 * c.half_len[d];
        }
    }
    n.norm_factor = c.norm_factor;
    return n;
}

// MIPSF_SINGLE_FP32 on a kernel: no packed fp32 instructions in it (v_pk_{add,mul,fma}_f32; everything inlined into the kernel
// follows).  It marks every kernel in which hipcc would otherwise emit a packed operation whose LOW result reads the HIGH half of a
// source (a 1 in op_sel).  On gfx950 such an operation, crossed on its SECOND source, reads that operand as 0 in lanes 48..63 while
// another wavefront of the same SIMD issues 16-bit MFMAs back to back -- the decoder's kernels, from another stream, another
// process or the other wave of the same kernel (stand-alone reproducer tools/micro/pk_lanes.hip:
// `./pk_lanes asm 5 inproc:10`).  tools/audit_packed.py (run by tests/test_host_cpu.py) lists the kernels that hold a crossed
// packed operation: none may.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MIPSF_KEEP_PACKED_FP32)
#define MIPSF_SINGLE_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MIPSF_SINGLE_FP32
#endif

__device__ __forceinline__ float normalise1(float p, double sub, double div, double nf) {
    return (float)((((double)p - sub) / div) / nf);
}

// ---- wave-level helpers (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace mipsf
