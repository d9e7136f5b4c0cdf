// Device helpers shared by the fp32-MFMA decoder kernels (decoder.hip) and the f16-MFMA ones (decoder16.hip):
// accumulator types, ReLU bit masks, buffer addressing, the positional-encoding prologue, bias / feature loads.
#pragma once
#include "common.h"
#include "decoder_layout.h"
#include <type_traits>

namespace mipsf {
using namespace dl;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float PI_F = 3.14159265358979323846f;
constexpr float HALF_PI_F = 1.57079632679489661923f;
constexpr int DEC_BLOCK = 256;
constexpr int TAIL_F4 = (PACKED_FLOATS - OFF_TRGB) / 4;  // rgb / sdf2 head tables, biases (contiguous tail of `packed`)
#ifndef MIPSF_FWD_LDS_MIN_ROUNDS
#define MIPSF_FWD_LDS_MIN_ROUNDS 2u   // experiments: a huge value disables the persistent forward kernel
#endif
#ifndef MIPSF_PIN_ARG
#define MIPSF_PIN_ARG 0          // experiments only: 1 skips the trickled stores (wrong results, timing)
#endif


typedef float f32x2 __attribute__((ext_vector_type(2)));

// ReLU masks for the backward chain.  The chain needs only the SIGN of H1 and H3, but used to read both tiles back
// (2 x 16 KB per 32 samples = 268 MB per launch, ~28 us of its 215).  The forward now appends two 64-bit words per
// lane and layer to `saved` (8 MB): bit 31-k of word w is "H > 0" for accumulator element (row tile 2w + (k >> 4),
// register k & 15).  Building a bit = v_cmp + v_addc (m = 2m + carry), using it = v_bfe_i32 + v_and.
constexpr int MASK_TILE_WORDS = 256;                 // [layer H1, H3][lane][2 words]
__device__ __forceinline__ uint32_t mask_push(uint32_t m, float h) {
    asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(h) : "vcc");
    return m;
}
__device__ __forceinline__ void relu_masks(const f32x16 (&H)[4], uint32_t (&m)[2]) {
    m[0] = 0u, m[1] = 0u;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m[rt >> 1] = mask_push(m[rt >> 1], H[rt][r]);
}
// g where element (rt, r) was active in the forward, 0 elsewhere
__device__ __forceinline__ float mask_apply(const uint32_t (&m)[2], int rt, int r, float g) {
    const int k = (rt & 1) * 16 + r;
    return __uint_as_float(__float_as_uint(g) & (uint32_t)__builtin_amdgcn_sbfe((int)m[rt >> 1], 31 - k, 1));
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc[rt] += A_image(rt, t) * B(t) for all k-steps; bfn(t) must fold to a register after unrolling
// PLACEMENT PIN.  The chain kernels take an always-zero kernel argument `pin`; the trickled stores below are
// wrapped in `if (pin == 0)`.  The branch is never taken, but because the compiler cannot prove that, it cannot
// sink/hoist those stores back into one burst in front of the next layer's operand loads (which it does otherwise:
// measured 410 us vs 313 us for the backward chain at 4096x64).
struct NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};

// acc[rt] += A_image(rt, t) * B(t) for all k-steps; bfn(t) must fold to a register after unrolling.
// side(t4) is invoked once per group of 4 k-steps AFTER the next group's operand loads have been issued: callers
// use it to trickle out stores (saved activations / pre-activation gradients) and reloads.  vmcnt retires in order
// on gfx9, so a burst of stores in front of the next layer's first loads stalls the matrix pipe for a full
// store round trip; one 16-byte store per group behind the loads costs nothing.
template <int RT, int T, typename BFn, typename SideFn = NoSide>
__device__ __forceinline__ void mfma_layer(const float4* img, int lane, f32x16 (&acc)[RT], BFn bfn,
                                           SideFn side = SideFn()) {
    constexpr int T4 = T / 4;
    // the A operands of group t4+1 are requested before the 4*RT MFMAs of group t4 issue, so an L2 round trip
    // (~500-900 cycles) hides under 4*RT*64 cycles of matrix work instead of stalling in front of it
    float4 a[RT], nxt[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = img[(rt * T4) * 64 + lane];
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
        if (t4 + 1 < T4) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) nxt[rt] = img[(rt * T4 + t4 + 1) * 64 + lane];
        }
        side(t4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].x, bfn(t4 * 4 + 0), acc[rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].y, bfn(t4 * 4 + 1), acc[rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].z, bfn(t4 * 4 + 2), acc[rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].w, bfn(t4 * 4 + 3), acc[rt]);
        if (t4 + 1 < T4) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = nxt[rt];
        }
    }
}

template <int RT>
__device__ __forceinline__ void load_bias(const float* tail, int layer, int h, f32x16 (&acc)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = tail[(OFF_BIAS - OFF_TRGB) + ((layer * 64 + rt * 16 + r) << 1) + h];
}

// Buffer addressing: one 128-bit resource (scalar) + a 32-bit lane offset (vector, computed once) + a scalar /
// immediate offset per access.  A flat `ptr[const + lane]` costs one or two 64-bit vector adds per access as soon as
// the constant leaves the 4 KB immediate range -- 350 vector instructions per tile in the backward chain, and vector
// instructions are what the matrix pipe waits for (DESIGN_NOTES.md 4b).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base_u, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_u), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load16(srd_t r, uint32_t lane16, uint32_t off_u) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, lane16, off_u, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
// the same load with a cache policy (aux bit 1 = nt: a stream that is read once and should not displace L2 / MALL lines)
template <int AUX>
__device__ __forceinline__ float4 buf_load16_aux(srd_t r, uint32_t lane16, uint32_t off_u) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, lane16, off_u, AUX);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
__device__ __forceinline__ void buf_store16(srd_t r, uint32_t lane16, uint32_t off_u, const float4& v) {
    u32x4 u;
    u.x = __float_as_uint(v.x), u.y = __float_as_uint(v.y), u.z = __float_as_uint(v.z), u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, lane16, off_u, 0);
}

// the same two stores against the tile's own buffer resource (forward kernels: no vector address arithmetic)
__device__ __forceinline__ void buf_store_act(srd_t sv, uint32_t lane16, int mat, const f32x16 (&acc)[4]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            buf_store16(sv, lane16, (mat * 16 + rt * 4 + g) * 1024,
                        make_float4(acc[rt][4 * g], acc[rt][4 * g + 1], acc[rt][4 * g + 2], acc[rt][4 * g + 3]));
}
__device__ __forceinline__ void buf_store_act_piece(srd_t sv, uint32_t lane16, int mat, const f32x16 (&acc)[4], int q) {
    const int rt = q >> 2, g = q & 3;
    buf_store16(sv, lane16, (mat * 16 + q) * 1024,
                make_float4(acc[rt][4 * g], acc[rt][4 * g + 1], acc[rt][4 * g + 2], acc[rt][4 * g + 3]));
}

// the 26 e values this lane feeds into layer 1 (its half of every k-step)
// sin of an fp32 argument |a| < ~2^10 as the oracle's sinf sees it, without libm's Payne-Hanek path (~65 vector
// instructions per value, 24 values per lane: a third of the forward kernel's non-matrix instructions).  The
// argument is reduced to revolutions with 1/(2 pi) split in two (the fma recovers the product's low bits, error
// < 1e-7 revolutions), then v_sin_f32.  The forward must see the oracle's ROUNDED argument: evaluating the exact
// argument instead moves e by up to 3e-5 and flips enough ReLUs to fail the scene gradient fixtures.
__device__ __forceinline__ float sin_reduced(float a) {
    const float C_HI = 0x1.45f306p-3f, C_LO = 0x1.b9391p-28f;
    const float n = rintf(a * C_HI);
    float f = fmaf(a, C_HI, -n);
    f = fmaf(a, C_LO, f);
    return __builtin_amdgcn_sinf(f);
}
__device__ __forceinline__ float cos_reduced(float a) {
    const float C_HI = 0x1.45f306p-3f, C_LO = 0x1.b9391p-28f;
    const float n = rintf(a * C_HI);
    float f = fmaf(a, C_HI, -n);
    f = fmaf(a, C_LO, f);
    return __builtin_amdgcn_cosf(f);
}
// the same from coordinates already in registers (in-kernel positional encoding)
__device__ __forceinline__ void e_from_x(float x0, float x1, float x2, int h, float (&ev)[E_SLOTS]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float xd = d == 0 ? x0 : (d == 1 ? x1 : x2);
#pragma unroll
        for (int k = 0; k < 8; ++k) ev[d * 8 + k] = sin_reduced(fmaf(ldexpf(xd, k), PI_F, h ? HALF_PI_F : 0.0f));
    }
    ev[24] = h ? x1 : x0;
    ev[25] = h ? 0.0f : x2;
}
template <bool PE_INTERNAL>
__device__ __forceinline__ void load_e(const float* __restrict__ x, const float* __restrict__ embed_pos,
                                       uint32_t s, int h, float (&ev)[E_SLOTS]) {
    const float x0 = x[3 * (size_t)s], x1 = x[3 * (size_t)s + 1], x2 = x[3 * (size_t)s + 2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float xd = d == 0 ? x0 : (d == 1 ? x1 : x2);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (PE_INTERNAL)
                ev[d * 8 + k] = sin_reduced(fmaf(ldexpf(xd, k), PI_F, h ? HALF_PI_F : 0.0f));
            else
                ev[d * 8 + k] = embed_pos[(size_t)s * N_PE + d * 16 + 2 * k + h];
        }
    }
    ev[24] = h ? x1 : x0;
    ev[25] = h ? 0.0f : x2;
}

template <int LAYOUT>
__device__ __forceinline__ float load_feat(const float* __restrict__ feat, uint32_t s, int level, int f, uint32_t M) {
    return LAYOUT == MIPSF_FEAT_AOS ? feat[(size_t)s * N_GRID + 2 * level + f] : feat[((size_t)level * M + s) * 2 + f];
}

int wgrad_reduce_launch(const float* partial, uint32_t nrec, const mipsf_decoder_grads* grads, hipStream_t s);

}  // namespace mipsf
