// SDF/colour decoder forward (model/decoder.py:53-75 of the reference) on the 16-bit matrix cores of gfx950
// (v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate): 16x the rate of the fp32-input MFMA the parity path of decoder.hip uses.
//
// Three arithmetic modes, one kernel template (NP = number of 16-bit planes an fp32 operand is cut into):
//   NP = 3 ("bf16x6") every fp32 operand -- weights and activations -- is carried EXACTLY as three bf16 pieces
//                     (p0 = rne(v), p1 = rne(v - p0), p2 = v - p0 - p1: 8 + 8 + 8 = the 24 significant bits of fp32, with
//                     fp32's exponent range) and a product is SIX MFMAs, p0*p0 + p0*p1 + p1*p0 + p1*p1 + p0*p2 + p2*p0,
//                     accumulated in fp32.  The three dropped pairs (p1*p2, p2*p1, p2*p2) are below 2^-23 of |a||w|, i.e.
//                     under the rounding of the fp32 accumulation itself: the arithmetic of five fp32 nn.Linear layers.
//                     6/16 of the fp32-MFMA time in the matrix pipe.
//   NP = 2 ("f16x3")  hi + lo f16 halves (hi = rne(v), lo = rne(v - hi): 22-23 significant bits), three MFMAs per product,
//                     hi*hi + hi*lo + lo*hi.  Relative error of a dot product ~3e-7 (fp32's own rounding is 6e-8): the
//                     fast training mode.  3/16 of the fp32-MFMA time in the matrix pipe.
//   NP = 1 ("f16")    one MFMA on the hi halves: 11-bit operands, fp32 accumulate.  Forward-only consumers whose
//                     tolerance allows it (RandomOptimizer fitness, BASELINE config 5 "fp16 decoder").
//
// Same transposed, register-chained evaluation as decoder.hip (decoder_layout.h): the C/D register layout of the MFMA
// does not depend on the operand type, so accumulator registers 8m..8m+7 of row tile q ARE the 8 B-operand elements
// of k-step 2q+m of the next layer after a float -> 16-bit conversion; the saved-activation record (`saved`: H1, H2, H3
// as accumulator images + ReLU masks) is bit-for-bit the layout the fp32 backward kernels read.
//
// Range (f16 modes): |activation| and |weight| must stay below 65504; the conversions saturate, they never produce inf.
// Where the operand planes live: planes 0 and 1 of the weight images in LDS (persistent kernels: 152 / 160 KB), plane 2 of
// the bf16 mode (used by ONE of the six products) in L2, requested five product groups ahead of its use.
#include "decoder_dev.h"
#include <stdlib.h>

namespace mipsf {
using namespace dl;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));       // 16 bytes of operand: 8 halves (f16 modes) or 8 bf16 (NP = 3)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <int NP>
__device__ __forceinline__ f32x16 mfmaP(h8 a, h8 b, f32x16 c) {
    if constexpr (NP == 3)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// 8 fp32 values -> hi (and lo) halves: v_cvt_pk_f16_f32 (rne) per pair, two v_cvt_f32_f16, one v_pk_add_f32, one
// more v_cvt_pk_f16_f32 for the residuals = 2.5 vector instructions per value
template <bool SPLIT>
__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
#if D16_ABL & 16
    {
        const _Float16 c = (_Float16)v[0];
        hi = h8{c, c, c, c, c, c, c, c}, lo = hi;
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const h2 p = {(_Float16)v[i], (_Float16)v[i + 1]};
        hi[i] = p.x, hi[i + 1] = p.y;
        if (SPLIT) {
            const h2 q = {(_Float16)(v[i] - (float)p.x), (_Float16)(v[i + 1] - (float)p.y)};
            lo[i] = q.x, lo[i + 1] = q.y;
        }
    }
}
// 8 fp32 values -> the NP operand planes.  NP = 3: three bf16 pieces, exact (v = p0 + p1 + p2): per pair v_cvt_pk_bf16_f32,
// the widening of the pair (a shift and a mask), v_pk_add_f32 -- twice -- and the last v_cvt_pk = 4.5 vector instructions per
// value.  Written with vector types and conversions only (no inline asm: an asm that reads a fresh MFMA result is not
// hazard-padded by hipcc, DESIGN_NOTES.md 4c).
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void cut8(const float (&v)[8], h8 (&pl)[3]) {
    if constexpr (NP == 3) {
        bf8 p0, p1, p2;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f32x2v x = {v[i], v[i + 1]};
            const bf2 a = __builtin_convertvector(x, bf2);
            const f32x2v r = x - __builtin_convertvector(a, f32x2v);
            const bf2 b = __builtin_convertvector(r, bf2);
            const f32x2v q = r - __builtin_convertvector(b, f32x2v);
            const bf2 c = __builtin_convertvector(q, bf2);
            p0[i] = a.x, p0[i + 1] = a.y, p1[i] = b.x, p1[i + 1] = b.y, p2[i] = c.x, p2[i + 1] = c.y;
        }
        pl[0] = __builtin_bit_cast(h8, p0), pl[1] = __builtin_bit_cast(h8, p1), pl[2] = __builtin_bit_cast(h8, p2);
    } else {
        split8<NP == 2>(v, pl[0], pl[1]);
    }
}

// 16-byte buffer store whose data registers may be rewritten right away.  hipcc (ROCm 7.2) pads the "VALU write of the
// data registers of a >8-byte store" hazard only when the store has NO scalar offset register (SIInstrInfo: the hazard
// "only exists if the instruction is not using a register in the soffset field") -- on gfx950 it exists with one as
// well: `buffer_store_dwordx4 v[168:171], .., s44 offen` followed by `v_pk_mul_f32 v[170:171]` stored the NEW values
// (every first of two back-to-back scaled stores of the backward chain was wrong).  So: constant part of the address in
// the vector offset (folds into the 12-bit immediate below 4 KB, one v_add above), soffset = 0.
#ifndef D16_ABL
#define D16_ABL 0   // experiments (wrong results; forward tile): 1 no heads, 2 no sin, 4 no record stores, 8 no LDS operand reads,
#endif              // 16 no hi/lo conversions, 32 no ReLU / unscale passes, 64 no softmax
// Cache policy of the forward's activation record stores: nt (aux bit 1).  The training forward is bound by its stores --
// 268 MB of record per launch; matrix work + stores alone take as long as the whole kernel (tools/micro/fwd_probe.py with
// -DD16_ABL=115: 102 us, the full kernel 98) -- and a CU gets through them at ~11 GB/s with the default policy (2.7 TB/s for
// the device, well under the 6 TB/s a plain write stream reaches); nt stores are acknowledged sooner: the same launch takes
// 72 us (skeleton) / 87 us (full kernel), the headline step 0.629 -> 0.623 ms.
#ifndef D16_FWD_STORE_AUX
#define D16_FWD_STORE_AUX 2
#endif
#ifndef D16_BWD_STORE_AUX
#define D16_BWD_STORE_AUX 0      // cache policy of the chain's gradient record stores (2 = nt)
#endif
template <int AUX = 0>
__device__ __forceinline__ void buf_store16_nosoff(srd_t r, uint32_t lane16, uint32_t const_off, const float4& v) {
    u32x4 u;
    u.x = __float_as_uint(v.x), u.y = __float_as_uint(v.y), u.z = __float_as_uint(v.z), u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, lane16 + const_off, 0, AUX);
}
__device__ __forceinline__ void store_act_piece(srd_t sv, uint32_t lane16, int mat, const f32x16 (&acc)[4], int q) {
    const int rt = q >> 2, g = q & 3;
    if (D16_ABL & 4) return;
    buf_store16_nosoff<D16_FWD_STORE_AUX>(sv, lane16, (mat * 16 + q) * 1024,
                       make_float4(acc[rt][4 * g], acc[rt][4 * g + 1], acc[rt][4 * g + 2], acc[rt][4 * g + 3]));
}

#ifdef D16_TRACE    // diagnosis builds (tools/micro/fwd_probe.py): cycles between the marks of a forward tile, summed per wave
__device__ unsigned long long d16_trace[4096 * 16];
#define D16_MARK(k) do { __builtin_amdgcn_sched_barrier(0); tr_t[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define D16_MARK(k) do { } while (0)
#endif
struct NoSide16 {
    __device__ __forceinline__ void operator()(int) const {}
};

// ReLU as ONE compiler-visible instruction: v_med3_f32(v, 0, +inf).  (fmaxf costs a canonicalising v_max in front of
// the real one; an inline-asm v_max reading an accumulator is invisible to hipcc's hazard padding: MFMA results must not
// be read by asm without the 12 wait states of the 8-pass XDL, and the first version of this file read stale values.)
__device__ __forceinline__ float relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff()); }
// 2^-W16_SHIFT: takes a completed accumulator back to its true magnitude (decoder_layout.h, RANGE)
// (the bf16 mode's images are unscaled: both factors are 1 there and the multiplications fold away)
template <int NP> constexpr float acc_unscale() { return NP == 3 ? 1.0f : 1.0f / (float)(1 << W16_SHIFT); }
template <int NP> constexpr float grid_upscale() { return NP == 3 ? 1.0f : (float)(1 << G16_SHIFT); }

// acc[rt] (+)= A_image(rt, t) * B(t) over a layer's k-steps, software-pipelined and FENCED: the A operands of k-step
// t+1 are requested and the B operand of k-step t+1 is converted while the MFMAs of k-step t run; the
// sched_barrier(0) fences keep hipcc from sinking every operand read next to its consumer (measured with the naive
// loop: ds_read + s_waitcnt lgkmcnt(0) in front of every MFMA pair, the matrix pipe 33 % busy).
//   BIAS: the hi image has one extra k-step in front whose B operand is the constant (1, 1, 0, ...) of half 0 and whose
//   A operand carries the bias as two halves; it also initialises the accumulators (C = 0).  Without BIAS the first
//   data k-step starts from C = 0 (layer 1: the bias sits in that layer's padding elements).
// img_hi: [rt][T + BIAS][lane], img_lo: [rt][T][lane] 16-byte operands (LDS in the persistent kernel, L2 otherwise).
// Where a layer's A operands come from: LDS (persistent kernels) or L2 through ONE buffer resource (16-byte loads with
// a scalar / immediate offset per operand: no 64-bit vector address arithmetic, cf. decoder_dev.h).
// Plane 2 (bf16 mode only) comes from L2 in both: one buffer resource over the plane-2 extension of `packed16`
// (decoder_layout.h, EXT16_*), byte offset `off_p2` of this layer's image inside it.
struct ImgLds {
    const h8* hi;
    const h8* lo;
    srd_t r2;
    uint32_t off_p2;
#if D16_ABL & 8
    __device__ __forceinline__ h8 load_hi(int idx, int lane) const { const _Float16 c = (_Float16)(float)(idx + lane); return h8{c, c, c, c, c, c, c, c}; }
    __device__ __forceinline__ h8 load_lo(int idx, int lane) const { const _Float16 c = (_Float16)(float)(idx - lane); return h8{c, c, c, c, c, c, c, c}; }
#else
    __device__ __forceinline__ h8 load_hi(int idx, int lane) const { return hi[idx * 64 + lane]; }
    __device__ __forceinline__ h8 load_lo(int idx, int lane) const { return lo[idx * 64 + lane]; }
#endif
    __device__ __forceinline__ h8 load_p2(int idx, int lane) const {
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r2, 16u * (uint32_t)lane, off_p2 + (uint32_t)idx * 1024u, 0));
    }
    // hi_halves / lo_halves: this layer's offsets in the plane-0 and plane-1 image sets; p2_halves: in the plane-2 set of the
    // same direction (forward: = lo_halves, the data k-steps only; backward: the same offset in all three)
    __device__ __forceinline__ ImgLds at(int hi_halves, int lo_halves) const {
        return ImgLds{hi + hi_halves / 8, lo + lo_halves / 8, r2, off_p2 + 2u * (uint32_t)lo_halves};
    }
};
struct ImgBuf {
    srd_t r;
    uint32_t off_hi, off_lo;      // byte offsets of the two image sets inside the resource
    srd_t r2;
    uint32_t off_p2;
    __device__ __forceinline__ h8 load_hi(int idx, int lane) const {
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r, 16u * (uint32_t)lane, off_hi + (uint32_t)idx * 1024u, 0));
    }
    __device__ __forceinline__ h8 load_lo(int idx, int lane) const {
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r, 16u * (uint32_t)lane, off_lo + (uint32_t)idx * 1024u, 0));
    }
    __device__ __forceinline__ h8 load_p2(int idx, int lane) const {
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r2, 16u * (uint32_t)lane, off_p2 + (uint32_t)idx * 1024u, 0));
    }
    __device__ __forceinline__ ImgBuf at(int hi_halves, int lo_halves) const {
        return ImgBuf{r, off_hi + 2u * (uint32_t)hi_halves, off_lo + 2u * (uint32_t)lo_halves, r2, off_p2 + 2u * (uint32_t)lo_halves};
    }
};
// the plane-2 extension of a bf16x6 `packed16` as a buffer resource (an empty one for the f16 modes: never read)
template <int NP>
__device__ __forceinline__ srd_t ext16_srd(const float* packed16) {
    return make_srd(NP == 3 ? reinterpret_cast<const _Float16*>(packed16 + PACKED16_FLOATS) : nullptr, NP == 3 ? EXT16_HALVES * 2 : 0);
}

constexpr int INIT_ACC = 0, INIT_ZERO = 1, INIT_BIAS = 2;
// number of B-operand elements of a bias k-step that are the constant 1.0 (the bias rides as that many pieces)
template <int NP> constexpr int bias_ones() { return NP == 3 ? 3 : 2; }
template <int RT, int T, int NP, int INIT, typename Img, typename BFn, typename SideFn = NoSide16>
__device__ __forceinline__ void mfma16_layer(const Img img, int lane, int h, f32x16 (&acc)[RT],
                                             BFn bfn, SideFn side = SideFn()) {
    constexpr bool BIAS = INIT == INIT_BIAS;
    constexpr bool SPLIT = NP >= 2;
    constexpr int TH = T + (BIAS ? 1 : 0);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (NP == 3) {
        // SIX products per k-step, in three groups around the operand fetches (A = weight planes a0, a1, a2; B = activation
        // planes b[0..2]):   a0*b0, a0*b1, a0*b2 | a1*b0, a1*b1 | a2*b0
        // a1 (LDS) is requested at the top of its k-step and first used a group later; the NEXT k-step's a0 (LDS) is requested
        // into the same registers once this k-step's a0 group has issued; the next k-step's a2 (L2: ~1 us away) right after
        // this k-step's only use of a2 -- five product groups ahead of its own.  48 operand registers for RT = 4.
        h8 a0[RT], a1[RT], a2[RT], b[3], nb[3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a0[rt] = img.load_hi(rt * TH + (BIAS ? 1 : 0), lane);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a2[rt] = img.load_p2(rt * T, lane);
        if (BIAS) {
            const _Float16 one = h == 0 ? __builtin_bit_cast(_Float16, (unsigned short)0x3f80) : (_Float16)0.0f;    // bf16 1.0
            const h8 ones = {one, one, one, 0, 0, 0, 0, 0};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfmaP<3>(img.load_hi(rt * TH, lane), ones, zero);
        }
        bfn(0, b);
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a1[rt] = img.load_lo(rt * T + t, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfmaP<3>(a0[rt], b[0], (INIT == INIT_ZERO && t == 0) ? zero : acc[rt]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfmaP<3>(a0[rt], b[1], acc[rt]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfmaP<3>(a0[rt], b[2], acc[rt]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < T) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a0[rt] = img.load_hi(rt * TH + t + 1 + (BIAS ? 1 : 0), lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfmaP<3>(a1[rt], b[1], acc[rt]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfmaP<3>(a1[rt], b[0], acc[rt]);
            __builtin_amdgcn_sched_barrier(0);
            // the next B operand is cut here, under the MFMAs in flight: planes 1 and 2 of this k-step are dead by now (their
            // registers take the new planes), only plane 0 is still wanted
            if (t + 1 < T) bfn(t + 1, nb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfmaP<3>(a2[rt], b[0], acc[rt]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < T) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a2[rt] = img.load_p2(rt * T + t + 1, lane);
            }
            side(t);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < T) b[0] = nb[0], b[1] = nb[1], b[2] = nb[2];
        }
        return;
    } else {
        // registers: the hi operands are double-buffered (requested one k-step ahead), the lo operands are requested at
        // the top of their own k-step -- they are first needed two MFMA groups (8 MFMAs, 256 cycles) later
        h8 ah[RT], al[RT], nh[RT], b[3], nb[3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ah[rt] = img.load_hi(rt * TH + (BIAS ? 1 : 0), lane);
        if (BIAS) {
            const _Float16 one = h == 0 ? (_Float16)1.0f : (_Float16)0.0f;
            const h8 ones = {one, one, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16(img.load_hi(rt * TH, lane), ones, zero);
        }
        bfn(0, b);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (SPLIT) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) al[rt] = img.load_lo(rt * T + t, lane);
            }
            if (t + 1 < T) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) nh[rt] = img.load_hi(rt * TH + t + 1 + (BIAS ? 1 : 0), lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16(ah[rt], b[0], (INIT == INIT_ZERO && t == 0) ? zero : acc[rt]);
            if (SPLIT) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16(ah[rt], b[1], acc[rt]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16(al[rt], b[0], acc[rt]);
            }
            if (t + 1 < T) bfn(t + 1, nb);           // conversions of the next B operand ride under these MFMAs
            side(t);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < T) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) ah[rt] = nh[rt];
                b[0] = nb[0], b[1] = nb[1];
            }
        }
    }
}

// A narrow head (decoder_layout.h, HEAD16): out = W_head * B over 8 k-steps; NPH = 2: hi/lo split like the layers (three MFMAs
// per k-step, two accumulators so that no MFMA waits for the one in front of it: acc0 = hi*hi + lo*hi, acc1 = hi*lo), NPH = 1
// the hi halves only, NPH = 3 the six products of the bf16 mode (acc0 and acc1 in turn).  Both f16 modes use the split heads:
// with hi halves only the plain f16 mode's forward error
// doubles (4.3e-4 against its stated 2e-4, test_decoder_true_error_of_every_arithmetic_against_fp64) for 6 % of a
// RandomOptimizer round.  himg: the head's
// compact image in LDS ([t][plane 0, 1][SLOTS] 16-byte operands), slot: this lane's operand slot; plane 2 ([t][SLOTS]) at byte
// offset off2 of the L2 resource r2.
template <int SLOTS, int NPH, typename BFn, typename SideFn = NoSide16>
__device__ __forceinline__ void mfma16_head(const h8* himg, int slot, srd_t r2, uint32_t off2, f32x16& acc0, f32x16& acc1, BFn bfn,
                                            SideFn side = SideFn()) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    h8 ah = himg[slot], al = himg[SLOTS + slot], nh, nl, b[3], nb[3], a2, na2;
    auto load2 = [&](int t) {
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r2, 16u * (uint32_t)slot, off2 + (uint32_t)(t * SLOTS) * 16u, 0));
    };
    if constexpr (NPH == 3) a2 = load2(0);
    bfn(0, b);
#pragma unroll
    for (int t = 0; t < T16_HEAD; ++t) {
        if (t + 1 < T16_HEAD) nh = himg[(2 * t + 2) * SLOTS + slot], nl = himg[(2 * t + 3) * SLOTS + slot];
        if constexpr (NPH == 3) {
            if (t + 1 < T16_HEAD) na2 = load2(t + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NPH == 3) {
            acc0 = mfmaP<3>(ah, b[0], t == 0 ? zero : acc0);
            acc1 = mfmaP<3>(ah, b[1], t == 0 ? zero : acc1);
            acc0 = mfmaP<3>(al, b[0], acc0);
            acc1 = mfmaP<3>(al, b[1], acc1);
            acc0 = mfmaP<3>(ah, b[2], acc0);
            acc1 = mfmaP<3>(a2, b[0], acc1);
        } else {
            acc0 = mfma16(ah, b[0], t == 0 ? zero : acc0);
            if (NPH == 2) {
                acc1 = mfma16(ah, b[1], t == 0 ? zero : acc1);
                acc0 = mfma16(al, b[0], acc0);
            } else if (t == 0) {
                acc1 = zero;
            }
        }
        if (t + 1 < T16_HEAD) bfn(t + 1, nb);
        side(t);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T16_HEAD) {
            ah = nh, al = nl, b[0] = nb[0], b[1] = nb[1], b[2] = nb[2];
            if constexpr (NPH == 3) a2 = na2;
        }
    }
}

// one wave, one tile of 32 samples.  tail: head images + head biases (LDS); img_hi / img_lo: the two operand image sets
// SAVE: 0 no record, 1 the full activation record, 2 the LEAN record -- H2, H3 and the ReLU masks; H1 (a third of the
// record) is left out: the streaming weight-gradient kernel recomputes it from x (wgrad16.hip) and nothing else reads it;
// 3 the ReLU MASKS only (32 B per sample instead of 1 KB): all the backward chain reads of the record -- for a frozen decoder
// (tracking: pose-only optimisation) no weight gradients follow and the activations would be written for nobody
template <int LAYOUT, int SAVE, bool SDF_ONLY, int NP, typename Img>
__device__ __forceinline__ void decoder16_fwd_tile(const float* tail, const Img img,
                                                   const float* __restrict__ feat, const float* __restrict__ x,
                                                   float* __restrict__ out, float* __restrict__ saved, uint32_t M,
                                                   int pin, int64_t tile, int lane_in, float (&xq)[3], int64_t next_tile) {
    constexpr int NPH = NP == 3 ? 3 : 2;       // planes of the two narrow heads (the f16 modes both use the split heads)
    // bf16 mode: the lane index is made opaque per tile, so that the dozen per-lane offsets derived from it (operand slots of
    // the heads, LDS and record addresses) are recomputed by every tile -- a few vector instructions -- instead of living across
    // the persistent kernel's tile loop: with 48 operand registers per layer there they were spilled, and a scratch reload waits
    // for every memory operation in flight (DESIGN_NOTES.md 4b)
    int lane = lane_in;
    if constexpr (NP == 3) asm volatile("" : "+v"(lane));
    const int j = lane & 31, h = lane >> 5;
    const uint32_t s_raw = (uint32_t)(tile * 32 + j);
    const bool live = s_raw < M;
    const uint32_t s = live ? s_raw : M - 1;   // tail lanes recompute the last sample (finite values, no stores)
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const srd_t sv = make_srd(SAVE ? saved + (size_t)tile * ACT_TILE_FLOATS : saved, SAVE ? ACT_TILE_FLOATS * 4 : 0);

#ifdef D16_TRACE
    unsigned long long tr_t[12];
    const unsigned long long tr_w0 = wall_clock64();
#endif
    D16_MARK(0);
    // xq: this tile's coordinates, loaded by the caller / by the previous tile of this wave BEFORE its record stores: memory
    // operations retire in order, and a load issued behind the 16 + stores of a tile waits for all of them (the e phase of the
    // training forward took 2800 cycles instead of 1600, tools/micro/fwd_probe.py)
    // the f16 modes keep the 26 e values of this lane for layer 1 and the rgb head; the bf16 mode has no registers to spare
    // for them (48 operand registers per layer) and evaluates the 8 values of a k-step where it cuts them -- twice per tile,
    // under the MFMAs of the k-step in front
    constexpr bool KEEP_E = NP != 3;
    float ev[KEEP_E ? E_SLOTS : 1];
    const float xe0 = xq[0], xe1 = xq[1], xe2 = xq[2];
    auto e_step = [&](int t, int u) -> float {        // e slot 8 t + u of this half (0 beyond the 26 slots)
        if constexpr (KEEP_E) {
            return 8 * t + u < E_SLOTS ? ev[(8 * t + u) < E_SLOTS ? 8 * t + u : 0] : 0.0f;
        } else {
            if (t < 3) return sin_reduced(fmaf(ldexpf(t == 0 ? xe0 : (t == 1 ? xe1 : xe2), u), PI_F, h ? HALF_PI_F : 0.0f));
            return u == 0 ? (h ? xe1 : xe0) : (u == 1 ? (h ? 0.0f : xe2) : 0.0f);
        }
    };
    if constexpr (KEEP_E) {
#if D16_ABL & 2
#pragma unroll
        for (int k = 0; k < E_SLOTS; ++k) ev[k] = xq[k % 3] * (float)(k + 1);
#else
        e_from_x(xq[0], xq[1], xq[2], h, ev);
#endif
    }

    D16_MARK(1);
    // ---- layer 1: pts_linear.0 + ReLU   (bias: elements BIAS16_U, +1 of k-step BIAS16_T meet the constant 1.0)
    f32x16 H1[4];
    mfma16_layer<RT_F1, T16_F1, NP, INIT_ZERO>(img.at(OFF16H_F1, OFF16L_F1), lane, h, H1,
        [&](int t, h8 (&b)[3]) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = e_step(t, u);
                if (t == BIAS16_T && u >= BIAS16_U && u < BIAS16_U + bias_ones<NP>()) v[u] = 1.0f;
            }
            cut8<NP>(v, b);
        });
    D16_MARK(2);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(D16_ABL & 32)) H1[rt][r] = relu1(H1[rt][r] * acc_unscale<NP>());
    uint32_t m1[2] = {0u, 0u};
    if (SAVE) relu_masks(H1, m1);

    D16_MARK(3);
    // ---- layer 2: pts_linear.2 -> [sdf_emb | rgb_emb]   (H1 leaves in two 16-byte pieces per k-step)
    constexpr int RT2 = SDF_ONLY ? 2 : RT_F2;
    f32x16 H2[RT2];
    mfma16_layer<RT2, T16_F2, NP, INIT_BIAS>(img.at(OFF16H_F2, OFF16L_F2), lane, h, H2,
        [&](int t, h8 (&b)[3]) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = H1[t >> 1][8 * (t & 1) + u];
            cut8<NP>(v, b);
        },
        [&](int t) {
            if constexpr (SAVE == 1) {
                store_act_piece(sv, lane16, 0, H1, 2 * t);
                store_act_piece(sv, lane16, 0, H1, 2 * t + 1);
            }
        });

    D16_MARK(4);
#pragma unroll
    for (int rt = 0; rt < RT2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(D16_ABL & 32)) H2[rt][r] = H2[rt][r] * acc_unscale<NP>();

    // grid features of layer 3 (feature h of the 16 levels): requested before the rgb head, which covers the latency
    float gf[16];
    // (the bf16 mode's launcher requires M < 2^24: no flat-pointer path, whose hoisted 64-bit addresses would spill)
    if (LAYOUT == MIPSF_FEAT_LEVEL_MAJOR && (NP == 3 || M < (1u << 24))) {
        const srd_t fs = make_srd(feat, M * 128u);
        const uint32_t voff = (2u * s + (uint32_t)h) * 4u;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            gf[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fs, voff, (uint32_t)u * M * 8u, 0));
    } else if (LAYOUT == MIPSF_FEAT_AOS && (NP == 3 || M < (1u << 24))) {
        const srd_t fs = make_srd(feat, M * 128u);        // [sample][level][feature]: one lane offset, 16 immediates
        const uint32_t voff = s * 128u + (uint32_t)h * 4u;
#pragma unroll
        for (int u = 0; u < 16; ++u) gf[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fs, voff, (uint32_t)u * 8u, 0));
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) gf[u] = load_feat<LAYOUT>(feat, s, u, h, M);
    }

    D16_MARK(5);
    // ---- rgb_linear.0 (3 outputs from [rgb_emb = H2 row tiles 2, 3 | e]).  f16 modes: on the matrix pipe, rows 0..2 of one
    // tile.  bf16 mode: on the VECTOR ALU in plain fp32 fmas against the fp32 table in LDS (this lane's half of every dot product,
    // then one swap) -- that kernel is bound by the matrix pipe (94 % busy at the sustained clock), the two heads were 96 of its
    // 544 MFMAs per tile for 8 useful output rows, and the vector ALU has the room: 308 instructions + 186 broadcast LDS reads.
    float rgb[3] = {0.f, 0.f, 0.f};
    if constexpr (!SDF_ONLY && NP == 3) {
        const float4* trgb = reinterpret_cast<const float4*>(tail) + h * TRGB_SLOTS;
        f32x2 p01 = {0.f, 0.f};
        float p2 = 0.f;
#pragma unroll
        for (int slot = 0; slot < 32; ++slot) {
            const float4 wv = trgb[slot];
            const float v = H2[2 + (slot >> 4)][slot & 15];
            p01 = __builtin_elementwise_fma(f32x2{wv.x, wv.y}, f32x2{v, v}, p01);
            p2 = fmaf(wv.z, v, p2);
            if constexpr (SAVE == 1 || SAVE == 2) {        // the rgb_emb half of H2 leaves piece by piece, as under the MFMA head
                if ((slot & 7) == 7) {
                    store_act_piece(sv, lane16, 1, H2, 8 + 2 * (slot >> 3));
                    store_act_piece(sv, lane16, 1, H2, 8 + 2 * (slot >> 3) + 1);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < E_SLOTS; ++t) {
            const float4 wv = trgb[32 + t];
            const float ev_t = e_step(t >> 3, t & 7);
            p01 = __builtin_elementwise_fma(f32x2{wv.x, wv.y}, f32x2{ev_t, ev_t}, p01);
            p2 = fmaf(wv.z, ev_t, p2);
        }
        const float pr[3] = {p01.x, p01.y, p2};
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = (pr[c] + __shfl_xor(pr[c], 32, 64)) + tail[OFF16_BSMALL + c];
    } else if constexpr (!SDF_ONLY) {
        f32x16 r0, r1;
        mfma16_head<HEAD16_RGB_SLOTS, NPH>(reinterpret_cast<const h8*>(tail) + HEAD16_SDF_HALVES / 8, head16_rgb_slot(j, h),
            img.r2, (uint32_t)EXT16_HEAD_RGB * 2u, r0, r1,
            [&](int t, h8 (&b)[3]) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = t < 4 ? H2[2 + (t >> 1)][8 * (t & 1) + u] : e_step(t < 4 ? 0 : t - 4, u);
                cut8<NPH>(v, b);
            },
            [&](int t) {       // the rgb_emb half of H2 leaves piece by piece behind the k-steps that read it: a burst of
                               // 8 (and of 16 for H3 below) stalls on the store path's back pressure (1300 / 1900 cycles)
                if constexpr (SAVE == 1 || SAVE == 2) {
                    if (t < 4) {
                        store_act_piece(sv, lane16, 1, H2, 8 + 2 * t);
                        store_act_piece(sv, lane16, 1, H2, 8 + 2 * t + 1);
                    }
                }
            });
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = (r0[c] + r1[c]) * acc_unscale<NP>() + tail[OFF16_BSMALL + c];      // (half 0's; half 1 holds zeros)
    }

    D16_MARK(6);
    D16_MARK(7);
    // ---- layer 3: sdf_linear.0 + ReLU on [sdf_emb (H2 tiles 0,1) | grid features]
    f32x16 H3[4];
    mfma16_layer<RT_F3, T16_F3, NP, INIT_BIAS>(img.at(OFF16H_F3, OFF16L_F3), lane, h, H3,
        [&](int t, h8 (&b)[3]) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = t < 4 ? H2[(t >> 1) & 1][8 * (t & 1) + u] : gf[(8 * (t - 4) + u) & 15] * grid_upscale<NP>();
            cut8<NP>(v, b);
        },
        [&](int t) {                                   // the sdf_emb half of H2 (8 pieces) over the first 4 k-steps
            if constexpr ((SAVE == 1 || SAVE == 2) && !SDF_ONLY) {
                if (t < 4) {
                    store_act_piece(sv, lane16, 1, H2, 2 * t);
                    store_act_piece(sv, lane16, 1, H2, 2 * t + 1);
                }
            }
        });
    if (next_tile >= 0) {            // the next tile's coordinates, in front of this tile's remaining stores
        const uint32_t sn_raw = (uint32_t)(next_tile * 32 + j);
        const uint32_t sn = sn_raw < M ? sn_raw : M - 1;
        xq[0] = x[3 * (size_t)sn], xq[1] = x[3 * (size_t)sn + 1], xq[2] = x[3 * (size_t)sn + 2];
    }
    D16_MARK(8);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(D16_ABL & 32)) H3[rt][r] = relu1(H3[rt][r] * acc_unscale<NP>());
    if (SAVE) {
        uint32_t m3[2];
        relu_masks(H3, m3);
        uint2* mk = reinterpret_cast<uint2*>(saved + (((size_t)M + 127) / 128) * 4 * ACT_TILE_FLOATS) +
                    (size_t)tile * (MASK_TILE_WORDS / 2) + lane;
        mk[0] = make_uint2(m1[0], m1[1]);
        mk[64] = make_uint2(m3[0], m3[1]);
    }

    D16_MARK(9);
    // ---- sdf_linear.2 (5 logits from H3): f16 modes on the matrix pipe -- logit c lands in register c of BOTH halves
    // (decoder_layout.h) --, bf16 mode on the vector ALU (see the rgb head); then softmax, entropy, expected class -> SDF
    float lg[N_CLASS], mx = -3.0e38f;
    if constexpr (NP == 3) {
        const float4* ts2 = reinterpret_cast<const float4*>(tail) + (OFF_TS2 - OFF_TRGB) / 4 + h * 128;
        f32x2 q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
        float q4 = 0.f;
#pragma unroll
        for (int slot = 0; slot < 64; ++slot) {
            const float4 w0 = ts2[2 * slot], w1 = ts2[2 * slot + 1];
            const float v = H3[slot >> 4][slot & 15];
            q01 = __builtin_elementwise_fma(f32x2{w0.x, w0.y}, f32x2{v, v}, q01);
            q23 = __builtin_elementwise_fma(f32x2{w0.z, w0.w}, f32x2{v, v}, q23);
            q4 = fmaf(w1.x, v, q4);
#ifndef D16_ABL_NO_H3_STORE
            if constexpr (SAVE == 1 || SAVE == 2) {
                if ((slot & 7) == 7) {
                    store_act_piece(sv, lane16, 2, H3, 2 * (slot >> 3));
                    store_act_piece(sv, lane16, 2, H3, 2 * (slot >> 3) + 1);
                }
            }
#endif
        }
        const float pl[N_CLASS] = {q01.x, q01.y, q23.x, q23.y, q4};
#pragma unroll
        for (int c = 0; c < N_CLASS; ++c) {
            lg[c] = (pl[c] + __shfl_xor(pl[c], 32, 64)) + tail[OFF16_BSMALL + 4 + c];
            mx = fmaxf(mx, lg[c]);
        }
    } else {
        f32x16 s0, s1;
        mfma16_head<HEAD16_SDF_SLOTS, NPH>(reinterpret_cast<const h8*>(tail), head16_sdf_slot(j, h), img.r2,
            (uint32_t)EXT16_HEAD_SDF * 2u, s0, s1,
            [&](int t, h8 (&b)[3]) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = H3[t >> 1][8 * (t & 1) + u];
                cut8<NPH>(v, b);
            },
            [&](int t) {
#ifndef D16_ABL_NO_H3_STORE         // experiments: what the forward would take without H3 in its record
                if constexpr (SAVE == 1 || SAVE == 2) {
                    store_act_piece(sv, lane16, 2, H3, 2 * t);
                    store_act_piece(sv, lane16, 2, H3, 2 * t + 1);
                }
#endif
            });
#pragma unroll
        for (int c = 0; c < N_CLASS; ++c) {
            lg[c] = (s0[c] + s1[c]) * acc_unscale<NP>() + tail[OFF16_BSMALL + 4 + c];
            mx = fmaxf(mx, lg[c]);
        }
    }
    D16_MARK(10);
    float p[N_CLASS], den = 0.f;
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) {
        p[c] = (D16_ABL & 64) ? lg[c] : expf(lg[c] - mx);
        den += p[c];
    }
    float ent = 0.f, cls = 0.f;
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) {
        p[c] = p[c] / den;
        ent += (D16_ABL & 64) ? p[c] : p[c] * log2f(p[c] + 1e-5f);
        cls += p[c] * (float)c;
    }
    const float sdf = (cls / 4.0f - 0.5f) * 2.0f;
#ifdef D16_TRACE
    D16_MARK(11);
    if (lane == 0) {
        const unsigned w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095u;
        for (int k = 0; k < 11; ++k) d16_trace[w * 16 + k] += tr_t[k + 1] - tr_t[k];
        d16_trace[w * 16 + 12] += wall_clock64() - tr_w0;       // 100 MHz
        d16_trace[w * 16 + 15] += 1ull;
    }
#endif
    if (SDF_ONLY) {
        if (live && h == 0) out[s] = sdf;
        return;
    }
    if (live) {
        float* o = out + (size_t)s * 10;
        if (h == 0) {
            o[0] = rgb[0], o[1] = rgb[1], o[2] = rgb[2], o[3] = sdf, o[4] = -1.0f * ent;
        } else {
            o[5] = p[0], o[6] = p[1], o[7] = p[2], o[8] = p[3], o[9] = p[4];
        }
    }
}

// n16 16-byte pieces of src -> LDS, by a whole workgroup of BLOCK threads: ten loads in flight per thread, then their LDS
// stores (a load -> store loop waits out one L2 round trip per iteration: 20 of them for the 160 KB of operand images, 6 us of
// a 70 us launch; this form 3 us)
template <int BLOCK>
__device__ __forceinline__ void lds_preload(float4* dst, const void* src_base, int n16) {
    constexpr int BATCH = 10;
    const srd_t src = make_srd(src_base, (uint32_t)n16 * 16u);
#pragma unroll 1
    for (int q0 = threadIdx.x; q0 < n16; q0 += BATCH * BLOCK) {
        float4 r[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) r[k] = buf_load16(src, 16u * (uint32_t)(q0 + k * BLOCK), 0u);   // (past the end: zeros)
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (q0 + k * BLOCK < n16) dst[q0 + k * BLOCK] = r[k];
    }
}

// Small batches: four independent waves per workgroup, operand images from L2, head tables + biases in LDS.
template <int LAYOUT, int SAVE, bool SDF_ONLY, int NP>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(DEC_BLOCK, 2) void decoder16_fwd_kernel(const float* __restrict__ packed16,
                                                                     const float* __restrict__ feat,
                                                                     const float* __restrict__ x,
                                                                     float* __restrict__ out,
                                                                     float* __restrict__ saved, uint32_t M, int pin,
                                                                     uint32_t* __restrict__ clear_hdr) {
    // the header (counters) of the backward chain's live-tile lists, cleared here for the chain kernel that follows this
    // forward: a memset in front of that kernel was a launch of its own (4.4 us of a step)
    if (clear_hdr != nullptr && blockIdx.x == 0)
        for (int q = threadIdx.x; q < (int)TL_HEADER; q += DEC_BLOCK) clear_hdr[q] = 0u;
    __shared__ float4 tailbuf[TAIL16_FLOATS / 4];
    for (int q = threadIdx.x; q < TAIL16_FLOATS / 4; q += DEC_BLOCK) tailbuf[q] = reinterpret_cast<const float4*>(packed16)[q];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (DEC_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile * 32 >= (int64_t)M) return;
    const ImgBuf img{make_srd(packed16 + TAIL16_FLOATS, (IMG16H_HALVES + IMG16L_HALVES) * 2), 0u, (uint32_t)IMG16H_HALVES * 2u,
                     ext16_srd<NP>(packed16), (uint32_t)EXT16_FWD * 2u};
    float xq[3];
    {
        const uint32_t s_raw = (uint32_t)(tile * 32 + (lane & 31));
        const uint32_t s0 = s_raw < M ? s_raw : M - 1;
        xq[0] = x[3 * (size_t)s0], xq[1] = x[3 * (size_t)s0 + 1], xq[2] = x[3 * (size_t)s0 + 2];
    }
    decoder16_fwd_tile<LAYOUT, SAVE, SDF_ONLY, NP>(reinterpret_cast<const float*>(tailbuf), img, feat, x, out, saved, M,
                                                      pin, tile, lane, xq, (int64_t)-1);
}

// Large batches: persistent, one 8-wave workgroup per CU with the operand images (80 KB hi, + 72 KB lo when SPLIT) and
// the head tables in LDS for its whole share of the batch (cf. decoder_fwd_lds_kernel).
constexpr int F16_LDS_BLOCK = 512;
template <int NP>
constexpr int f16_lds_bytes() { return TAIL16_FLOATS * 4 + IMG16H_HALVES * 2 + (NP >= 2 ? IMG16L_HALVES * 2 : 0); }
template <int LAYOUT, int SAVE, bool SDF_ONLY, int NP>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(F16_LDS_BLOCK, 1) void decoder16_fwd_lds_kernel(const float* __restrict__ packed16,
                                                                             const float* __restrict__ feat,
                                                                             const float* __restrict__ x,
                                                                             float* __restrict__ out,
                                                                             float* __restrict__ saved, uint32_t M,
                                                                             int pin, uint32_t n_tiles,
                                                                             uint32_t* __restrict__ clear_hdr) {
    if (clear_hdr != nullptr && blockIdx.x == 0)      // (see decoder16_fwd_kernel)
        for (int q = threadIdx.x; q < (int)TL_HEADER; q += F16_LDS_BLOCK) clear_hdr[q] = 0u;
    extern __shared__ __attribute__((aligned(16))) float4 wbuf[];
#ifdef D16_TRACE
    const unsigned long long tr_k0 = wall_clock64();
#endif
    lds_preload<F16_LDS_BLOCK>(wbuf, packed16, f16_lds_bytes<NP>() / 16);
    __syncthreads();
#ifdef D16_TRACE
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095u;
        d16_trace[w * 16 + 13] = tr_k0, d16_trace[w * 16 + 14] = wall_clock64();
    }
#endif
    const int lane = threadIdx.x & 63;
    // EXPERIMENT (pin >> 8 = number of s_sleep(127)): delay the second wave of every SIMD so that the two co-resident
    // waves are not in the same phase (both converting, then both wanting the matrix pipe)
    if ((threadIdx.x >> 6) >= 4)
        for (int q = 0; q < (pin >> 8); ++q) __builtin_amdgcn_s_sleep(127);
    pin &= 255;
    const uint32_t tile0 = blockIdx.x * (F16_LDS_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t stride = gridDim.x * (F16_LDS_BLOCK / 64);
    float xq[3] = {0.f, 0.f, 0.f};
    if (tile0 < n_tiles) {
        const uint32_t s_raw = tile0 * 32u + (uint32_t)(lane & 31);
        const uint32_t s0 = s_raw < M ? s_raw : M - 1;
        xq[0] = x[3 * (size_t)s0], xq[1] = x[3 * (size_t)s0 + 1], xq[2] = x[3 * (size_t)s0 + 2];
    }
    for (uint32_t tile = tile0; tile < n_tiles; tile += stride) {
        uint32_t z = 0;                       // opaque zero: keeps the loop-invariant LDS operand reads inside the loop
        asm volatile("" : "+v"(z));
        const float4* w4 = wbuf + z;
        const h8* imgp = reinterpret_cast<const h8*>(w4 + TAIL16_FLOATS / 4);
        decoder16_fwd_tile<LAYOUT, SAVE, SDF_ONLY, NP>(reinterpret_cast<const float*>(w4),
                                                          ImgLds{imgp, imgp + IMG16H_HALVES / 8, ext16_srd<NP>(packed16), (uint32_t)EXT16_FWD * 2u},
                                                          feat, x, out, saved, M,
                                                          pin, (int64_t)tile, lane, xq,
                                                          tile + stride < n_tiles ? (int64_t)(tile + stride) : (int64_t)-1);
    }
#ifdef D16_TRACE
    if (lane == 0) d16_trace[((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095u) * 16 + 11] = wall_clock64();
#endif
}

// ============================================================================ backward chain on the f16 matrix cores
// d(out) -> d(grid features), d(x) and the pre-activation gradients dG3, dH2, dG1 + (d logits, d rgb) the
// weight-gradient kernel consumes (`dact`, the fp32 record of decoder_bwd_lds_kernel, same layout).  Register-chained
// like the forward: every gradient tile is an accumulator image whose registers are the next product's B operand after
// a float -> (hi, lo) conversion; the two narrow products (Ws2^T dlogits, Wrgb^T drgb: K = 5 and 3) ride on the matrix
// pipe as one k-step each instead of ~420 fmas and 160 table reads per tile.  Operand images come from L2.
// LIVE-TILE BUFFER (mipsf_decoder_bwd_chain16 -> mipsf_decoder_wgrad16), in words:
//     [64 q]        hand-out counter of queue q = blockIdx.x & 7 (persistent kernel: tiles are dealt dynamically)
//     [64 q + 32]   number of live tiles in list q
//     [512 + q cap] list q: the tiles with a non-zero gradient, cap = tl_cap(n_tiles)
// Eight queues / lists because ONE counter is one L2 atomic unit: ~12 ns per operation, 8192 tile grabs = 100 us (measured:
// the kernel took 194 us instead of 97).  Workgroup b runs on XCD b % 8, so a queue stays within one XCD's L2; the
// counters sit 128 bytes apart.  Queue q owns the groups of 8 consecutive tiles g with g % 8 == q.
// (TL_HEADER, tl_cap: decoder_layout.h)
__device__ __forceinline__ void tl_append(uint32_t* tl, uint32_t n_tiles, uint32_t tile) {
    const uint32_t q = (tile >> 3) & 7u;
    tl[TL_HEADER + q * tl_cap(n_tiles) + atomicAdd(tl + 64 * q + 32, 1u)] = tile;
}

template <int LAYOUT, int NP, typename Img>
__device__ __forceinline__ void decoder16_bwd_tile(const Img bimg, const float* __restrict__ x,
                                                   const float* __restrict__ out, const float* __restrict__ dout,
                                                   const float* __restrict__ saved, float* __restrict__ dfeat,
                                                   float* __restrict__ dx, float* __restrict__ dact,
                                                   float* __restrict__ dsmall, uint32_t M, int64_t tile, int lane_in,
                                                   uint32_t* __restrict__ tile_live = nullptr, bool lean_dact = false) {
    static_assert(NP == 2 || NP == 3, "the chain runs on split operands");
    int lane = lane_in;                        // opaque per tile: per-lane offsets are recomputed by every tile instead of being
    asm volatile("" : "+v"(lane));             // kept (and spilled) across the persistent kernel's tile loop (cf. the forward)
    const int j = lane & 31, h = lane >> 5;
    const uint32_t s_raw = (uint32_t)(tile * 32 + j);
    const bool live = s_raw < M;
    const uint32_t s = live ? s_raw : M - 1;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    // dact == nullptr: nobody will ask for weight gradients (a frozen decoder: tracking) -- the 48 KB of pre-activation
    // gradients per tile are not written (an empty buffer resource drops the stores: no branch around them)
    const srd_t da = make_srd(dact ? dact + (size_t)tile * ACT_TILE_FLOATS : nullptr, dact ? ACT_TILE_FLOATS * 4 : 0);
    // LEAN gradient record (MIPSF_CHAIN_LEAN_DACT; the exchange form of the weight-gradient kernel follows): dG3 and the
    // rgb_emb half of dH2 are not written -- half of the record's 48 KB per tile.  Both are one narrow product of values the
    // weight-gradient kernel has anyway (dG3 = relu'(H3) (Ws2^T dlogits): the 5 logit gradients and the mask bits; d rgb_emb =
    // Wrgb^T drgb: 3 values): it recomputes them with this function's own operations, bit for bit (wgrad16.hip).  This
    // kernel is bound by its stores.
    const srd_t da_opt = make_srd(dact ? dact + (size_t)tile * ACT_TILE_FLOATS : nullptr, (dact && !lean_dact) ? ACT_TILE_FLOATS * 4 : 0);

#ifdef D16_TRACE          // (tools/replay.py prints these for a -DD16_TRACE build)
    unsigned long long tr_t[12];
    const unsigned long long tr_w0 = wall_clock64();
#endif
    D16_MARK(0);
    float2 o2[5], g2[5];
    {
        const float2* g = reinterpret_cast<const float2*>(dout + (size_t)s * 10);
#pragma unroll
        for (int c = 0; c < 5; ++c) g2[c] = g[c];
    }
    // ZERO TILES.  A sample behind the truncation band has no loss term and no rendering weight: its incoming gradient
    // is exactly zero, and so is everything this function derives from it.  Along a ray those samples are the tail (33 of
    // 64 on the mapping workload), so a third of the 32-sample tiles are zero throughout (tools/micro/dout_zero_probe.py).
    // With `tile_live` the caller asks for them to be short-cut: d(features) and d(x) are written as zeros, nothing else is
    // read or written, and the tile is left out of the list of live tiles the weight-gradient kernel works from.
    if (tile_live != nullptr) {
        bool any = false;
#pragma unroll
        for (int c = 0; c < 5; ++c) any = any || (live && !(g2[c].x == 0.0f && g2[c].y == 0.0f));
        const bool tile_any = __any(any) != 0;
        if (tile_any && lane == 0) tl_append(tile_live, (uint32_t)(((uint64_t)M + 31) / 32), (uint32_t)tile);
        if (!tile_any) {
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = rowmap(r, h);
                    if (LAYOUT == MIPSF_FEAT_AOS)
                        *reinterpret_cast<float2*>(dfeat + (size_t)s * N_GRID + row) = make_float2(0.f, 0.f);
                    else
                        *reinterpret_cast<float2*>(dfeat + ((size_t)(row >> 1) * M + s) * 2) = make_float2(0.f, 0.f);
                }
                float z0 = 0.f;                       // (made here: a loop-invariant zero triple was hoisted out of the tile
                asm volatile("" : "+v"(z0));          // loop and spilled; its reload waits for every memory operation in flight)
                if (h == 0) dx[3 * (size_t)s] = z0, dx[3 * (size_t)s + 1] = z0, dx[3 * (size_t)s + 2] = z0;
            }
            return;
        }
    }
    D16_MARK(1);
    {
        const float2* o = reinterpret_cast<const float2*>(out + (size_t)s * 10);
#pragma unroll
        for (int c = 0; c < 5; ++c) o2[c] = o[c];
    }
    // (the coordinates are needed last, but a load issued behind this tile's 48 record stores would wait for all of them:
    // memory operations retire in order)
#ifndef D16_BWD_X_LATE
    const float x0 = x[3 * (size_t)s], x1 = x[3 * (size_t)s + 1], x2 = x[3 * (size_t)s + 2];
#endif
    const uint2* mk = reinterpret_cast<const uint2*>(saved + (((size_t)M + 127) / 128) * 4 * ACT_TILE_FLOATS) +
                      (size_t)tile * (MASK_TILE_WORDS / 2) + lane;
    const uint2 mk1 = mk[0], mk3 = mk[64];
    const uint32_t m1[2] = {mk1.x, mk1.y}, m3[2] = {mk3.x, mk3.y};

    D16_MARK(2);
    // ---- softmax / entropy / expected-class backward -> d logits; d rgb is the incoming gradient itself
    float dlg[N_CLASS], drgb[3];
    {
        const float gv[10] = {g2[0].x, g2[0].y, g2[1].x, g2[1].y, g2[2].x, g2[2].y, g2[3].x, g2[3].y, g2[4].x, g2[4].y};
        const float ov[10] = {o2[0].x, o2[0].y, o2[1].x, o2[1].y, o2[2].x, o2[2].y, o2[3].x, o2[3].y, o2[4].x, o2[4].y};
        const float g_sdf = live ? gv[3] : 0.f, g_ent = live ? gv[4] : 0.f;
        float p[N_CLASS], dp[N_CLASS], dot = 0.f;
#pragma unroll
        for (int c = 0; c < N_CLASS; ++c) {
            p[c] = ov[5 + c];
            const float q = p[c] + 1e-5f;
            const float dent = -1.0f * (log2f(q) + p[c] / (q * 0.69314718055994530942f));
            dp[c] = (live ? gv[5 + c] : 0.f) + g_sdf * (0.5f * (float)c) + g_ent * dent;
            dot += p[c] * dp[c];
        }
#pragma unroll
        for (int c = 0; c < N_CLASS; ++c) dlg[c] = p[c] * (dp[c] - dot);
#pragma unroll
        for (int c = 0; c < 3; ++c) drgb[c] = live ? gv[c] : 0.f;
        if (h == 0 && dact != nullptr) {
            float4* d4 = reinterpret_cast<float4*>(dsmall + (size_t)(tile * 32 + j) * 8);
            d4[0] = make_float4(dlg[0], dlg[1], dlg[2], dlg[3]);
            d4[1] = make_float4(dlg[4], drgb[0], drgb[1], drgb[2]);
        }
    }
    // RANGE.  Loss gradients are tiny (1e-7 .. 1e-3: a mean over N*S samples) and f16 has no exponent to spare below
    // 6e-5: unscaled, the hi halves are subnormal and the chain keeps 4-8 bits (measured: grid gradient 1e-3 off).  The
    // chain is LINEAR in (d logits, d rgb) and every sample is its own column of every product, so each sample scales
    // its incoming gradient by a power of two that brings its largest component to [0.5, 1) -- exact -- and every
    // stored result is multiplied by the inverse power of two -- exact again.
    float up = 1.0f, down = 1.0f;
    {
        float mx = 0.0f;
#pragma unroll
        for (int c = 0; c < N_CLASS; ++c) mx = fmaxf(mx, fabsf(dlg[c]));
#pragma unroll
        for (int c = 0; c < 3; ++c) mx = fmaxf(mx, fabsf(drgb[c]));
        if (mx > 0.0f && mx < 3.0e38f) {
            // mx = f * 2^e, f in [0.5, 1).  A SUBNORMAL maximum (e < -126: a sample whose whole incoming gradient is below
            // 1.2e-38 -- a softmax probability of e^-87 times a loss gradient) would ask for 2^-e > 2^127 = inf, and inf x 0
            // = NaN would reach the grid through d feat: the exponent is held at -126 (the scaled values are then below
            // 0.5: still exact, just not normalised).  csrc/wgrad16.hip, w16x_updown, replays this bit for bit.
            const int e = max(__builtin_amdgcn_frexp_expf(mx), -126);
            up = ldexpf(1.0f, -e), down = ldexpf(1.0f, e);
        }
    }
    // B operands of the two narrow products: half 0 carries the 5 (3) values in elements 0..4 (0..2), half 1 zeros
    h8 lgp[3], rgp[3];
    {
        float v[8], r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = (h == 0 && u < N_CLASS) ? dlg[u < N_CLASS ? u : 0] * up : 0.0f;
            r[u] = (h == 0 && u < 3) ? drgb[u < 3 ? u : 0] * up : 0.0f;
        }
        cut8<NP>(v, lgp);
        cut8<NP>(r, rgp);
    }
    // a 16-byte piece of a (scaled) gradient tile, back at its true magnitude, into `dact`
    auto store_piece = [&](int mat, const f32x16 (&acc)[4], int q) {
        const int rt = q >> 2, g = q & 3;
        buf_store16_nosoff<D16_BWD_STORE_AUX>((mat == 2 || (mat == 1 && q >= 8)) ? da_opt : da, lane16, (mat * 16 + q) * 1024,
                           make_float4(acc[rt][4 * g] * down, acc[rt][4 * g + 1] * down, acc[rt][4 * g + 2] * down,
                                       acc[rt][4 * g + 3] * down));
    };

    D16_MARK(3);
    // ---- dG3 = relu'(H3) * (Ws2^T dlogits)
    f32x16 dG3[4];
    mfma16_layer<RT16_S2T, T16_S2T, NP, INIT_ZERO>(bimg.at(OFF16B_S2T, OFF16B_S2T), lane, h, dG3,
        [&](int, h8 (&b)[3]) { b[0] = lgp[0], b[1] = lgp[1], b[2] = lgp[2]; });
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dG3[rt][r] = mask_apply(m3, rt, r, dG3[rt][r] * acc_unscale<NP>());

    D16_MARK(4);
    // ---- d[sdf_emb | grid] = Ws1^T dG3   (row tiles 0,1 -> d sdf_emb, 2 -> d grid features); dG3 leaves for `dact`
    f32x16 dIn3[3];
    mfma16_layer<RT16_B3, T16_B3, NP, INIT_ZERO>(bimg.at(OFF16B_B3, OFF16B_B3), lane, h, dIn3,
        [&](int t, h8 (&b)[3]) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dG3[t >> 1][8 * (t & 1) + u];
            cut8<NP>(v, b);
        },
        [&](int t) {
            store_piece(2, dG3, 2 * t);
            store_piece(2, dG3, 2 * t + 1);
        });
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dIn3[rt][r] = dIn3[rt][r] * acc_unscale<NP>();
    if (live) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int row = rowmap(r, h);
            const float2 v = make_float2(dIn3[2][r] * down, dIn3[2][r + 1] * down);
            if (LAYOUT == MIPSF_FEAT_AOS)
                *reinterpret_cast<float2*>(dfeat + (size_t)s * N_GRID + row) = v;
            else
                *reinterpret_cast<float2*>(dfeat + ((size_t)(row >> 1) * M + s) * 2) = v;
        }
    }

    D16_MARK(5);
    // ---- dH2 = [d sdf_emb | d rgb_emb = Wrgb[:, :64]^T drgb]
    f32x16 dH2[4];
    dH2[0] = dIn3[0], dH2[1] = dIn3[1];
    {
        f32x16 dRgb[2];
        mfma16_layer<RT16_RGBT, T16_RGBT, NP, INIT_ZERO>(bimg.at(OFF16B_RGBT, OFF16B_RGBT), lane, h, dRgb,
            [&](int, h8 (&b)[3]) { b[0] = rgp[0], b[1] = rgp[1], b[2] = rgp[2]; });
        dH2[2] = dRgb[0] * acc_unscale<NP>(), dH2[3] = dRgb[1] * acc_unscale<NP>();
    }

    D16_MARK(6);
    // ---- dG1 = relu'(H1) * (W2^T dH2)
    f32x16 dG1[4];
    mfma16_layer<RT16_B2, T16_B2, NP, INIT_ZERO>(bimg.at(OFF16B_B2, OFF16B_B2), lane, h, dG1,
        [&](int t, h8 (&b)[3]) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dH2[t >> 1][8 * (t & 1) + u];
            cut8<NP>(v, b);
        },
        [&](int t) {
            store_piece(1, dH2, 2 * t);
            store_piece(1, dH2, 2 * t + 1);
        });
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dG1[rt][r] = mask_apply(m1, rt, r, dG1[rt][r] * acc_unscale<NP>());

    D16_MARK(7);
    // ---- d e = W1^T dG1 + Wrgb[:, 64:]^T drgb; rows are arranged so that e-slot (t, h) lands in THIS lane
    f32x16 dE[2];
    mfma16_layer<RT16_B1, T16_B1, NP, INIT_ZERO>(bimg.at(OFF16B_B1, OFF16B_B1), lane, h, dE,
        [&](int t, h8 (&b)[3]) {
            if (t == 8) {
                b[0] = rgp[0], b[1] = rgp[1], b[2] = rgp[2];
            } else {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = dG1[(t >> 1) & 3][8 * (t & 1) + u];
                cut8<NP>(v, b);
            }
        },
        [&](int t) {
            if (t < 8) {
                store_piece(0, dG1, 2 * t);
                store_piece(0, dG1, 2 * t + 1);
            }
        });

    D16_MARK(8);
#ifdef D16_BWD_X_LATE      // experiments: the old place of the load
    const float x0 = x[3 * (size_t)s], x1 = x[3 * (size_t)s + 1], x2 = x[3 * (size_t)s + 2];
#endif
    float de[E_SLOTS];
#pragma unroll
    for (int t = 0; t < E_SLOTS; ++t) de[t] = dE[t >> 4][t & 15] * (down * acc_unscale<NP>());
    float g3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float xd = d == 0 ? x0 : (d == 1 ? x1 : x2);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // derivative at the argument the forward used (cos_reduced: 6 instructions)
            const float arg = fmaf(ldexpf(xd, k), PI_F, h ? HALF_PI_F : 0.0f);
            g3[d] = g3[d] + de[d * 8 + k] * ((ldexpf(1.0f, k) * PI_F) * cos_reduced(arg));
        }
    }
    g3[0] += h == 0 ? de[24] : 0.0f;   // slot 24 carries x0 (lower half) / x1 (upper half)
    g3[1] += h == 1 ? de[24] : 0.0f;
    g3[2] += h == 0 ? de[25] : 0.0f;   // slot 25 carries x2 (lower half only)
#pragma unroll
    for (int d = 0; d < 3; ++d) g3[d] = g3[d] + __shfl_xor(g3[d], 32, 64);
    if (live && h == 0) {
        dx[3 * (size_t)s] = g3[0], dx[3 * (size_t)s + 1] = g3[1], dx[3 * (size_t)s + 2] = g3[2];
    }
#ifdef D16_TRACE
    D16_MARK(9);
    if (lane == 0) {
        const unsigned w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095u;
        for (int k = 0; k < 9; ++k) d16_trace[w * 16 + k] += tr_t[k + 1] - tr_t[k];
        d16_trace[w * 16 + 12] += wall_clock64() - tr_w0;
        d16_trace[w * 16 + 15] += 1ull;
    }
#endif
}

// Small batches: four independent waves per workgroup, operand images from L2.
template <int LAYOUT, int NP>
__global__ __launch_bounds__(DEC_BLOCK, 2) void decoder16_bwd_kernel(const float* __restrict__ packed16,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ out,
                                                                     const float* __restrict__ dout,
                                                                     const float* __restrict__ saved,
                                                                     float* __restrict__ dfeat, float* __restrict__ dx,
                                                                     float* __restrict__ dact, float* __restrict__ dsmall,
                                                                     uint32_t M, uint32_t* __restrict__ tile_live, uint32_t lean_dact) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (DEC_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile * 32 >= (int64_t)M) return;
    const ImgBuf bimg{make_srd(reinterpret_cast<const _Float16*>(packed16 + TAIL16_FLOATS) + OFF16_BWD_HALVES,
                               IMG16B_HALVES * 4), 0u, (uint32_t)IMG16B_HALVES * 2u, ext16_srd<NP>(packed16), (uint32_t)EXT16_BWD * 2u};
    decoder16_bwd_tile<LAYOUT, NP>(bimg, x, out, dout, saved, dfeat, dx, dact, dsmall, M, tile, lane, tile_live, lean_dact != 0u);
}

// Large batches: persistent, one 8-wave workgroup per CU holding BOTH backward image sets in LDS: 2 x 80 KB = all 160 KB
// of the CU (the backward needs no tables besides them).
constexpr int B16_LDS_BYTES = IMG16B_HALVES * 4;
static_assert(B16_LDS_BYTES <= 160 * 1024, "the backward image sets must fit the LDS of a CU");
template <int LAYOUT, int NP>
__global__ __launch_bounds__(F16_LDS_BLOCK, 1) void decoder16_bwd_lds_kernel(const float* __restrict__ packed16,
                                                                             const float* __restrict__ x,
                                                                             const float* __restrict__ out,
                                                                             const float* __restrict__ dout,
                                                                             const float* __restrict__ saved,
                                                                             float* __restrict__ dfeat,
                                                                             float* __restrict__ dx,
                                                                             float* __restrict__ dact,
                                                                             float* __restrict__ dsmall, uint32_t M,
                                                                             uint32_t n_tiles,
                                                                             uint32_t* __restrict__ tile_live, uint32_t lean_dact) {
    extern __shared__ __attribute__((aligned(16))) float4 wbuf[];
    lds_preload<F16_LDS_BLOCK>(wbuf, reinterpret_cast<const _Float16*>(packed16 + TAIL16_FLOATS) + OFF16_BWD_HALVES, B16_LDS_BYTES / 16);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // Round k: the workgroup's 8 waves take 8 consecutive tiles, wave w the ((w + k) & 7)-th of them (every wave
    // alternates between first and second halves of rays, whose gradients differ in sparsity).
    constexpr uint32_t WPB = F16_LDS_BLOCK / 64;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile_live != nullptr && gridDim.x >= 8u) {           // (all eight queues need a workgroup: always, on 256 CUs)
        // With the short cut a wave's four tiles cost anything between 0 and 4 live ones, and the kernel would last as
        // long as its unluckiest wave (measured: -12 % instead of the -35 % of the tiles skipped): tiles are handed out
        // through counters instead (one per queue, see the buffer layout above; cleared by the launcher).
        const uint32_t q = blockIdx.x & 7u;
        uint32_t* counter = tile_live + 64 * q;
        // (grabbing the next tile before working on this one was measured slower, 95 vs 86 us: the atomic's round trip is
        // longer than a load's and the tile's first loads queue behind it)
        for (;;) {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(counter, 1u);
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
            const uint32_t tile = (((i >> 3) * 8u + q) << 3) + (i & 7u);          // group (i / 8) * 8 + q, member i % 8
            if ((tile & ~7u) >= n_tiles) break;
            if (tile >= n_tiles) continue;
            uint32_t z = 0;
            asm volatile("" : "+v"(z));
            const h8* imgp = reinterpret_cast<const h8*>(wbuf + z);
            decoder16_bwd_tile<LAYOUT, NP>(ImgLds{imgp, imgp + IMG16B_HALVES / 8, ext16_srd<NP>(packed16), (uint32_t)EXT16_BWD * 2u},
                                       x, out, dout, saved, dfeat, dx, dact, dsmall,
                                       M, (int64_t)tile, lane, tile_live, lean_dact != 0u);
        }
        return;
    }
    uint32_t k = 0;
    for (uint32_t first = blockIdx.x * WPB; first < n_tiles; first += gridDim.x * WPB, ++k) {
        const uint32_t tile = first + ((w + k) & (WPB - 1));
        if (tile >= n_tiles) continue;
        uint32_t z = 0;
        asm volatile("" : "+v"(z));
        const h8* imgp = reinterpret_cast<const h8*>(wbuf + z);
        decoder16_bwd_tile<LAYOUT, NP>(ImgLds{imgp, imgp + IMG16B_HALVES / 8, ext16_srd<NP>(packed16), (uint32_t)EXT16_BWD * 2u},
                                   x, out, dout, saved, dfeat, dx, dact, dsmall, M,
                                   (int64_t)tile, lane, tile_live, lean_dact != 0u);
    }
}

// NP = 2: f16 hi / lo images (the f16x3 and plain f16 modes).  NP = 3: the three bf16 planes of the bf16x6 mode -- planes 0, 1
// where the f16 layout has hi, lo, plane 2 in the extension behind it (decoder_layout.h, EXT16_*).
template <int NP>
__global__ __launch_bounds__(256) void decoder_pack16_kernel(W w, float* __restrict__ packed16) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    uint16_t* ext = reinterpret_cast<uint16_t*>(packed16 + PACKED16_FLOATS);
    // element of plane `pl` behind the fp32 value v, as its 16 storage bits
    auto piece = [&](float v, int pl) -> uint16_t {
        if (NP == 3) return bf16_bits(bf16_piece(v, pl));
        const _Float16 hi = (_Float16)v;
        return __builtin_bit_cast(uint16_t, pl == 0 ? hi : (_Float16)(v - (float)hi));
    };
    if (NP == 3) {
        // the bf16 mode runs its two narrow heads on the VECTOR ALU in plain fp32 (the matrix pipe is what bounds that kernel):
        // the head region holds the fp32 tables of decoder.hip (rgb: [half][58 slots][4], sdf: [half][64 slots][8])
        static_assert(OFF_BIAS - OFF_TRGB <= HEAD16_HALVES / 2, "the fp32 head tables fit the head region of the tail");
        if (idx < OFF_BIAS - OFF_TRGB) packed16[idx] = packed_value(w, OFF_TRGB + idx);
    } else if (idx < HEAD16_HALVES) {
        int plane;
        const float v = head16_weight(w, idx, plane, NP);
        reinterpret_cast<uint16_t*>(packed16)[idx] = piece(v, plane);
    }
    if (idx < 12) packed16[OFF16_BSMALL + idx] = packed_value(w, OFF_BSMALL + idx);
    if (idx < IMG16H_HALVES) {
        uint16_t* img = reinterpret_cast<uint16_t*>(packed16 + TAIL16_FLOATS);
        const float v = img16_weight(w, idx, NP);
        img[idx] = piece(v, 0);
        const int lo = img16_lo_index(idx);
        if (lo >= 0) {
            img[IMG16H_HALVES + lo] = piece(v, 1);
            if (NP == 3) ext[EXT16_FWD + lo] = piece(v, 2);
        }
    }
    if (idx < IMG16B_HALVES) {
        uint16_t* img = reinterpret_cast<uint16_t*>(packed16 + TAIL16_FLOATS) + OFF16_BWD_HALVES;
        const float v = img16b_weight(w, idx, NP);
        img[idx] = piece(v, 0);
        img[IMG16B_HALVES + idx] = piece(v, 1);
        if (NP == 3) ext[EXT16_BWD + idx] = piece(v, 2);
    }
}

static W to_w16(const mipsf_decoder_weights& s) {
    W w;
    w.w_pts0 = s.w_pts0, w.b_pts0 = s.b_pts0, w.w_pts2 = s.w_pts2, w.b_pts2 = s.b_pts2, w.w_rgb0 = s.w_rgb0;
    w.b_rgb0 = s.b_rgb0, w.w_sdf0 = s.w_sdf0, w.b_sdf0 = s.b_sdf0, w.w_sdf2 = s.w_sdf2, w.b_sdf2 = s.b_sdf2;
    return w;
}

}  // namespace mipsf

using namespace mipsf;

// tiles per CU from which the persistent (LDS-resident operand images) kernels are used (experiments: MIPSF_PERSIST_MIN)
static uint32_t persist_min_tiles_per_cu() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MIPSF_PERSIST_MIN");
        v = e ? atoi(e) : 16;
        if (v < 1) v = 1;
    }
    return (uint32_t)v;
}

namespace mipsf {
uint64_t decoder_packed16_floats(int precision) {
    return precision == MIPSF_PREC_BF16X6 ? (uint64_t)PACKED16X_FLOATS : (uint64_t)PACKED16_FLOATS;
}
}  // namespace mipsf

extern "C" {

#ifdef D16_TRACE
int mipsf_d16_trace_read(unsigned long long* host, int clear) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(d16_trace), sizeof(unsigned long long) * 4096 * 16) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long z[4096 * 16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(d16_trace), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
// precision MIPSF_PREC_F16X3 / MIPSF_PREC_F16: the f16 hi / lo images (one buffer serves both modes);
// MIPSF_PREC_BF16X6: the three bf16 planes (MIPSF_SIZE_DECODER_PACKED16 floats for that precision) -- a buffer packed for
// one family must not be handed to the kernels of the other
int mipsf_decoder_pack16(const mipsf_decoder_weights* w, float* packed16, int precision, void* stream) {
    MIPSF_REQUIRE(w && packed16, "null pointer");
    MIPSF_REQUIRE(precision == MIPSF_PREC_F16X3 || precision == MIPSF_PREC_F16 || precision == MIPSF_PREC_BF16X6,
                  "precision must be f16x3, f16 or bf16x6");
    static_assert(IMG16H_HALVES >= HEAD16_HALVES && IMG16H_HALVES >= IMG16B_HALVES, "one thread per hi-image element covers all");
    static_assert(f16_lds_bytes<2>() <= 160 * 1024, "tail + both image sets must fit the 160 KB of LDS of a CU");
    if (precision == MIPSF_PREC_BF16X6)
        hipLaunchKernelGGL(decoder_pack16_kernel<3>, dim3((IMG16H_HALVES + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           to_w16(*w), packed16);
    else
        hipLaunchKernelGGL(decoder_pack16_kernel<2>, dim3((IMG16H_HALVES + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           to_w16(*w), packed16);
    return check_launch("decoder_pack16");
}

// tile_live_clear (optional): the live-tile buffer the backward chain of THIS forward will fill
// (mipsf_decoder_bwd_chain16 with MIPSF_CHAIN_HEADER_CLEAR): its counters are cleared by this launch.
int mipsf_decoder_fwd16(const mipsf_decoder_fwd16_args* a, void* stream) {
    MIPSF_REQUIRE(a != nullptr, "null argument block");
    MIPSF_REQUIRE(a->struct_size == sizeof(mipsf_decoder_fwd16_args), "mipsf_decoder_fwd16_args: struct_size %u, this library expects %u",
                  a->struct_size, (unsigned)sizeof(mipsf_decoder_fwd16_args));
    const float* packed16 = a->packed16; const float* feat = a->feat; const int feat_layout = a->feat_layout; const float* x = a->x;
    float* out = a->out; float* saved = a->saved; const int sdf_only = a->sdf_only, precision = a->precision, lean_record = a->lean_record;
    const uint32_t M = a->M;
    uint32_t* clear_hdr = a->tile_live_clear;
    MIPSF_REQUIRE(a->packed16_floats == 0u || a->packed16_floats == decoder_packed16_floats(precision),
                  "packed16 holds %u floats, precision %d needs %u: packed for the other family?", a->packed16_floats, precision,
                  (unsigned)decoder_packed16_floats(precision));
    if (M == 0) return 0;
    MIPSF_REQUIRE(packed16 && feat && x && out, "null pointer");
    MIPSF_REQUIRE(!lean_record || (saved && (precision == MIPSF_PREC_F16X3 || precision == MIPSF_PREC_BF16X6)),
                  "the lean record belongs to the f16x3 / bf16x6 training forward");
    MIPSF_REQUIRE(feat_layout == MIPSF_FEAT_AOS || feat_layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout");
    MIPSF_REQUIRE(precision == MIPSF_PREC_F16X3 || precision == MIPSF_PREC_F16 || precision == MIPSF_PREC_BF16X6,
                  "precision must be f16x3, f16 or bf16x6");
    MIPSF_REQUIRE(!(sdf_only && saved), "the SDF-only forward keeps no activations");
    MIPSF_REQUIRE(precision != MIPSF_PREC_BF16X6 || M < (1u << 24), "bf16x6: M = %u, at most 2^24 - 1 samples per call", M);
    const uint32_t n_tiles = (uint32_t)(((uint64_t)M + 31) / 32);
    const uint32_t blocks = (n_tiles + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
    const int cus = device_cus();
    if (cus <= 0) return 3;
    const bool persistent = n_tiles >= (uint32_t)cus * persist_min_tiles_per_cu();
    const char* st_env = getenv("MIPSF_F16_STAGGER");
    const int stagger = st_env ? atoi(st_env) : 0;
#define F16(LAY, SV, SDF, SPL)                                                                                     \
    do {                                                                                                           \
        if (persistent) {                                                                                          \
            static bool attr_set_dev[MAX_DEVICES] = {false};                                                       \
            bool& attr_set = attr_set_dev[device_slot()];                                                          \
            if (!attr_set) {                                                                                       \
                if (hipFuncSetAttribute((const void*)decoder16_fwd_lds_kernel<LAY, SV, SDF, SPL>,                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, f16_lds_bytes<SPL>()) !=       \
                    hipSuccess) {                                                                                  \
                    set_error("cannot raise dynamic LDS to %d bytes", f16_lds_bytes<SPL>());                       \
                    return 4;                                                                                      \
                }                                                                                                  \
                attr_set = true;                                                                                   \
            }                                                                                                      \
            hipLaunchKernelGGL((decoder16_fwd_lds_kernel<LAY, SV, SDF, SPL>), dim3(cus), dim3(F16_LDS_BLOCK),      \
                               f16_lds_bytes<SPL>(), s, packed16, feat, x, out, saved, M, stagger << 8, n_tiles,   \
                               clear_hdr);                                                                         \
        } else {                                                                                                   \
            hipLaunchKernelGGL((decoder16_fwd_kernel<LAY, SV, SDF, SPL>), dim3(blocks), dim3(DEC_BLOCK), 0, s,     \
                               packed16, feat, x, out, saved, M, 0, clear_hdr);                                    \
        }                                                                                                          \
    } while (0)
#define F16_MODE(LAY, SPL, SPLS)                            \
    do {                                                    \
        if (sdf_only) F16(LAY, 0, true, SPL);               \
        else if (saved != nullptr && lean_record == 2) F16(LAY, 3, false, SPLS); \
        else if (saved != nullptr && lean_record) F16(LAY, 2, false, SPLS); \
        else if (saved != nullptr) F16(LAY, 1, false, SPL); \
        else F16(LAY, 0, false, SPL);                       \
    } while (0)
    if (feat_layout == MIPSF_FEAT_AOS) {
        if (precision == MIPSF_PREC_BF16X6) F16_MODE(MIPSF_FEAT_AOS, 3, 3);
        else if (precision == MIPSF_PREC_F16X3) F16_MODE(MIPSF_FEAT_AOS, 2, 2);
        else F16_MODE(MIPSF_FEAT_AOS, 1, 2);
    } else {
        if (precision == MIPSF_PREC_BF16X6) F16_MODE(MIPSF_FEAT_LEVEL_MAJOR, 3, 3);
        else if (precision == MIPSF_PREC_F16X3) F16_MODE(MIPSF_FEAT_LEVEL_MAJOR, 2, 2);
        else F16_MODE(MIPSF_FEAT_LEVEL_MAJOR, 1, 2);
    }
#undef F16_MODE
#undef F16
    return check_launch("decoder_fwd16");
}


// tile_live (optional, MIPSF_SIZE_DECODER_TILE_WORDS words): receives the lists of the 32-sample tiles whose incoming
// gradient is not zero throughout (layout above); the other tiles get d(features) = d(x) = 0 and NO entry in `dact` --
// hand the same buffer to mipsf_decoder_wgrad16.
// flags: MIPSF_CHAIN_HEADER_CLEAR = the counters of tile_live were cleared by the forward (tile_live_clear) and not
// used since; MIPSF_CHAIN_LEAN_DACT = the lean gradient record (see decoder16_bwd_tile)
int mipsf_decoder_bwd_chain16(const mipsf_decoder_chain16_args* a, void* stream) {
    MIPSF_REQUIRE(a != nullptr, "null argument block");
    MIPSF_REQUIRE(a->struct_size == sizeof(mipsf_decoder_chain16_args), "mipsf_decoder_chain16_args: struct_size %u, this library expects %u",
                  a->struct_size, (unsigned)sizeof(mipsf_decoder_chain16_args));
    const float* packed16 = a->packed16; const int feat_layout = a->feat_layout; const float* x = a->x; const float* out = a->out;
    const float* dout = a->dout; const float* saved = a->saved; float* dfeat = a->dfeat; float* dx = a->dx; float* dact = a->dact;
    uint32_t* tile_live = a->tile_live; const int flags = a->flags; const uint32_t M = a->M;
    {
        const int fam = (flags & MIPSF_CHAIN_BF16X6) ? MIPSF_PREC_BF16X6 : MIPSF_PREC_F16X3;
        MIPSF_REQUIRE(a->packed16_floats == 0u || a->packed16_floats == decoder_packed16_floats(fam),
                      "packed16 holds %u floats, this chain needs %u: packed for the other family?", a->packed16_floats,
                      (unsigned)decoder_packed16_floats(fam));
    }
    if (M == 0) return 0;
    MIPSF_REQUIRE((flags & ~(MIPSF_CHAIN_HEADER_CLEAR | MIPSF_CHAIN_LEAN_DACT | MIPSF_CHAIN_BF16X6)) == 0, "unknown flags 0x%x", flags);
    const bool bf = (flags & MIPSF_CHAIN_BF16X6) != 0;
    const int header_is_clear = flags & MIPSF_CHAIN_HEADER_CLEAR;
    const uint32_t lean_dact = (flags & MIPSF_CHAIN_LEAN_DACT) ? 1u : 0u;
    MIPSF_REQUIRE(packed16 && x && out && dout && saved && dfeat && dx, "null pointer");      // (dact may be NULL: see the header)
    MIPSF_REQUIRE(feat_layout == MIPSF_FEAT_AOS || feat_layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout");
    const uint64_t n_bt = ((uint64_t)M + 127) / 128;
    const uint32_t blocks = (uint32_t)((((uint64_t)M + 31) / 32 + 3) / 4);
    float* dsmall = dact ? dact + n_bt * 4 * ACT_TILE_FLOATS : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n_tiles = (uint32_t)(((uint64_t)M + 31) / 32);
    const int cus = device_cus();
    if (cus <= 0) return 3;
    const bool persistent = n_tiles >= (uint32_t)cus * persist_min_tiles_per_cu() && !getenv("MIPSF_B16_NO_LDS");
    if (tile_live != nullptr && !header_is_clear && hipMemsetAsync(tile_live, 0, TL_HEADER * sizeof(uint32_t), s) != hipSuccess) {
        set_error("cannot clear the tile counters");
        return 4;
    }
#define B16(LAY, NPL)                                                                                              \
    do {                                                                                                           \
        if (persistent) {                                                                                          \
            static bool attr_set_dev[MAX_DEVICES] = {false};                                                       \
            bool& attr_set = attr_set_dev[device_slot()];                                                          \
            if (!attr_set) {                                                                                       \
                if (hipFuncSetAttribute((const void*)decoder16_bwd_lds_kernel<LAY, NPL>,                           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, B16_LDS_BYTES) != hipSuccess) { \
                    set_error("cannot raise dynamic LDS to %d bytes", B16_LDS_BYTES);                              \
                    return 4;                                                                                      \
                }                                                                                                  \
                attr_set = true;                                                                                   \
            }                                                                                                      \
            hipLaunchKernelGGL((decoder16_bwd_lds_kernel<LAY, NPL>), dim3(cus), dim3(F16_LDS_BLOCK), B16_LDS_BYTES, s, \
                               packed16, x, out, dout, saved, dfeat, dx, dact, dsmall, M, n_tiles, tile_live,      \
                               lean_dact);                                                                         \
        } else {                                                                                                   \
            hipLaunchKernelGGL((decoder16_bwd_kernel<LAY, NPL>), dim3(blocks), dim3(DEC_BLOCK), 0, s, packed16, x, out, \
                               dout, saved, dfeat, dx, dact, dsmall, M, tile_live, lean_dact);                     \
        }                                                                                                          \
    } while (0)
    if (feat_layout == MIPSF_FEAT_AOS) {
        if (bf) B16(MIPSF_FEAT_AOS, 3); else B16(MIPSF_FEAT_AOS, 2);
    } else {
        if (bf) B16(MIPSF_FEAT_LEVEL_MAJOR, 3); else B16(MIPSF_FEAT_LEVEL_MAJOR, 2);
    }
#undef B16
    return check_launch("decoder_bwd_chain16");
}

}  // extern "C"
