// Decoder weight gradients on the 16-bit matrix cores.  THREE forms live in this file; which one a launch takes:
//   * behind the lean records (packed16 given: H1 recomputed; the default of the Python modules), f16x3 and bf16x6:
//       the TRANSPOSE-READ form (round 6, w16t_role_a / w16t_role_b, far below): operands several waves need are prepared
//       once per tile and handed over through LDS, every transposition is an LDS write + ds_read_b64_tr_b16, the small-row
//       products run on v_mfma_f32_16x16x32.  -DW16_TR=0 builds its predecessor, the EXCHANGE form with matrix-core
//       transposes (rounds 3-5, w16x_role_a / w16x_role_b) -- kept as the A/B baseline of DESIGN.md 4.7;
//   * full records (packed16 null) and the two-plane bf16 arithmetic: the STREAMING form described next (round 2,
//       w16_role_a / w16_role_b): no LDS, no barriers, the matrix cores do the transposes.
//
// The streaming form:
//
//   dW[out][in] = sum over samples s of dOut[s][out] * In[s][in]
// has the SAMPLE as reduction index, while every activation / gradient record of the decoder (`saved`, `dact`:
// accumulator images of the transposed evaluation, decoder_layout.h) has the sample in the LANE and the feature in the
// register -- the wrong way round for an MFMA operand, whose reduction index must sit inside a lane.  decoder.hip
// transposes through LDS (4 phases x 2 barriers per 128 samples, one wave per SIMD, 0.46 of its fp32-MFMA peak and
// 3.4 TB/s).  Here the matrix core transposes:
//
//     T[sample][c] = sum_k X[sample][feature k] * I[k][c]          (I = 0/1 selection matrix)
//
// is an ordinary 16-bit v_mfma_f32_32x32x16 whose A operand is the record AS LOADED (lane = sample, 8 features per
// k-step) and whose result -- lane = feature c, registers = 16 samples -- is exactly the operand layout the
// weight-gradient product wants.  It is EXACT: every product is a 16-bit value times 1.0, every sum adds zeros.  An
// fp32 record is cut into P planes first, each plane is transposed on its own (2 MFMAs per 32 x 32 block and plane), and
// the weight-gradient product keeps the plane pairs with pa + pb <= P - 1.  Three arithmetics:
//   f16, P = 2   v = hi + lo, 22 bits: 3 products per k-step, dropped remainder 2^-22 (fp32 class) -- the default.  f16
//                has no exponent range to spare (loss gradients are 1e-3 .. 1e-9), and a per-sample scale does not
//                factor out of a sum over samples.  So every GRADIENT block (32 samples x 32 features) is multiplied by
//                a power of two 2^k that brings its largest entry below 2^15, and the accumulators fed by that block
//                are kept in the same scaled domain: when a tile needs another k they are multiplied by 2^(k' - k)
//                first -- exact -- which, with a hysteresis of 2^7, happens rarely; the flush multiplies by 2^-k.  A
//                block's entries 2^10 below its maximum still carry all 22 bits (f16's subnormal floor, 2^-25 absolute,
//                sits 2^40 below the maximum).  A single launch-wide scale is not enough: free-space samples with
//                gradients 2^-17 below the batch maximum are the bulk of a sum and lose the lo plane (measured: 3e-5
//                in the sequence test).  Grid features ride at 2^12 like in the forward.
//   bf16, P = 3  v = hi + mid + lo, 24 bits with fp32's exponent range, no scale: 6 products per k-step, 2^-24.
//   bf16, P = 2  3 products, 2^-16 (for comparison).
//
// A workgroup of 8 waves walks the 32-sample tiles of its share of the batch; every wave owns a few 32 x 32 output
// tiles for the whole launch (<= 6 accumulator tiles = 96 registers -> two waves per SIMD) and loads, cuts and
// transposes only the feature rows and columns those tiles need:
//     waves 0..3 (rt = w):      d w_pts2[rt][0..3] = dH2[rt]^T H1            + (small rows)^T H3[col tile w]
//     waves 4..7 (rt = w - 4):  d w_sdf0[rt][0..2] = dG3[rt]^T [sdf_emb | grid],  d w_pts0[rt][0..1] = dG1[rt]^T e,
//                               (small rows)^T {rgb_emb col tile 0 | 1 | e col tile 0 | 1}[w - 4]
// (small rows = d logits (5) and d rgb (3) of `dsmall`: d w_sdf2, d w_rgb0).  Bias gradients are register sums of the
// transposed tiles.  Per-block partial records + the reduce kernel of decoder.hip (no atomics on weights).
#include "decoder_dev.h"

namespace mipsf {
using namespace dl;

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

struct ArF16 { typedef h8 v8; typedef _Float16 elt; static constexpr int P = 2; static constexpr bool SCALED = true; };
struct ArBF3 { typedef bf8 v8; typedef __bf16 elt; static constexpr int P = 3; static constexpr bool SCALED = false; };
struct ArBF2 { typedef bf8 v8; typedef __bf16 elt; static constexpr int P = 2; static constexpr bool SCALED = false; };
constexpr float W16_GRID_SHIFT = 4096.0f;      // 2^G16_SHIFT
// the bf16 planes ride unscaled (decoder_layout.h, w16_scale): both factors are 1 for the unscaled arithmetics
template <typename A> constexpr float w16_acc_unscale() { return A::SCALED ? 1.0f / (float)(1 << W16_SHIFT) : 1.0f; }
template <typename A> constexpr float w16_grid_shift() { return A::SCALED ? W16_GRID_SHIFT : 1.0f; }

__device__ __forceinline__ f32x16 mfma16(bf8 a, bf8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// next 16-bit plane of 8 fp32 residuals: pl = narrow(r), r -= pl (exact) unless it is the last plane
template <typename A, bool LAST>
__device__ __forceinline__ typename A::v8 next_plane(f32x8& r) {
    typename A::v8 pl;
    if constexpr (A::SCALED && !LAST) {
        // f16: r - (float)half in ONE instruction per value -- v_fma_mix_f32 widens the selected half of the packed pair
        // itself (hipcc emits cvt_f32_f16 + cvt_f32_f16_sdwa + pk_add: 4 instead of 3 instructions per pair)
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            h2_t t;
            t[0] = (_Float16)r[i], t[1] = (_Float16)r[i + 1];
            pl[i] = t[0], pl[i + 1] = t[1];
            float lo, hi;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(t), "v"(r[i]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(t), "v"(r[i + 1]));
            r[i] = lo, r[i + 1] = hi;
        }
    } else if constexpr (std::is_same<typename A::elt, __bf16>::value) {
        // bf16: a PAIR per v_cvt_pk_bf16_f32, widened by a shift and a mask, residual by one v_pk_add_f32
        typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
        typedef float f2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f2_t x = {r[i], r[i + 1]};
            const bf2_t t = __builtin_convertvector(x, bf2_t);
            pl[i] = t.x, pl[i + 1] = t.y;
            if (!LAST) {
                const f2_t q = x - __builtin_convertvector(t, f2_t);
                r[i] = q.x, r[i + 1] = q.y;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const typename A::elt t = (typename A::elt)r[i];
            pl[i] = t;
            if (!LAST) r[i] = r[i] - (float)t;
        }
    }
    return pl;
}

// 16 fp32 accumulator registers holding 16-bit values -> the two k-step operands (samples 8m..8m+7 of this half); exact
template <typename A>
__device__ __forceinline__ void pack_T(const f32x16& T, typename A::v8 (&op)[2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < 8; ++u) op[m][u] = (typename A::elt)T[8 * m + u];
}

// Ordering pins.  The matrix instructions and conversions are pure values to the compiler, which is free to compute all
// planes' cuts first and to sink every transposing MFMA down to its use -- three planes of 16-register transposes alive
// at once, and the kernel spills (a sched_barrier only binds the machine scheduler, not the IR passes before it).  An
// empty asm that "rewrites" the residuals and the finished operands makes plane p + 1 depend on plane p being packed.
#define W16_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define W16_PIN5(a, b, c, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))

// One 32-feature block: A operands vals[q][0..7] (q = 0, 1: the 8 values this lane contributes to MFMA q) ->
// transposed operands ops[p][m] (plane p, k-step m = samples 8m..8m+7 of this half) of lane = column c, where the value
// of element u of MFMA q, contributed by half h, lands in column c = 16 q + 8 (u >> 2) + 4 h + (u & 3).
// For the accumulator images (pieces g = 2q + (u >> 2), element u & 3) that is column = feature - 32 * row tile.
// NQ = 1: only vals[0] is non-zero (the small rows), placed at columns 16 q0 + ...   SUM: rowsum += this lane's 16
// samples of the block (bias gradients; plane by plane, each plane's sum is exact to fp32 rounding).  vals is consumed.
#ifdef W16_ABL_FREE_CUT     // ablation (wrong results): the 16-bit planes of a block cost nothing -- what a record that arrives as
// ready operand planes would save, as an upper bound (the recomputed blocks' cuts are free here as well)
template <typename A>
__device__ __forceinline__ typename A::v8 abl_plane(const f32x8& r, int p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const int o = (2 * p) & 4;
    const u4 w = {__float_as_uint(r[o]), __float_as_uint(r[o + 1]), __float_as_uint(r[o + 2]), __float_as_uint(r[o + 3])};
    return __builtin_bit_cast(typename A::v8, w);
}
#define W16_PLANE(A, LASTP, V, P) abl_plane<A>(V, P)
#else
#define W16_PLANE(A, LASTP, V, P) ((LASTP) ? next_plane<A, true>(V) : next_plane<A, false>(V))
#endif
template <typename A, bool SUM, int NQ = 2>
__device__ __forceinline__ void transpose_block(f32x8 (&vals)[2], const typename A::v8 (&I)[2],
                                                typename A::v8 (&ops)[A::P][2], float& rowsum, int q0 = 0) {
    constexpr int P = A::P;
#ifdef W16_DBG_NO_COMPUTE       // diagnosis builds (tools/micro): loads only
    asm volatile("" ::"v"(vals[0]), "v"(vals[1]));
    return;
#endif
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        f32x16 T = mfma16(W16_PLANE(A, p == P - 1, vals[0], p), I[q0], zero);
        if (NQ == 2) T = mfma16(W16_PLANE(A, p == P - 1, vals[1], p), I[1], T);
        pack_T<A>(T, ops[p]);
        if (SUM) {
            float sp = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sp = sp + T[r];
            s = s + sp;
        }
        if (p + 1 < P) W16_PIN5(vals[0], vals[1], ops[p][0], ops[p][1], s);
    }
    if (SUM) rowsum = rowsum + s;
}

// The right-hand side of a product, plane by plane: plane pb of the block is transposed (2 MFMAs) and multiplied at
// once with the planes pa <= P - 1 - pb of X (the dropped pairs are below 2^-(plane bits x P) of the product); only ONE
// transposed plane is alive at a time.  vals is consumed.
template <typename A>
__device__ __forceinline__ void transpose_mac(f32x8 (&vals)[2], const typename A::v8 (&I)[2],
                                              const typename A::v8 (&X)[A::P][2], f32x16& acc) {
    constexpr int P = A::P;
#ifdef W16_DBG_NO_COMPUTE
    asm volatile("" ::"v"(vals[0]), "v"(vals[1]));
    return;
#endif
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pb = 0; pb < P; ++pb) {
        f32x16 T = mfma16(W16_PLANE(A, pb == P - 1, vals[0], pb), I[0], zero);
        T = mfma16(W16_PLANE(A, pb == P - 1, vals[1], pb), I[1], T);
        typename A::v8 Y[2];
        pack_T<A>(T, Y);
        if (pb + 1 < P) W16_PIN4(vals[0], vals[1], Y[0], Y[1]);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pa = 0; pa + pb < P; ++pa) acc = mfma16(X[pa][m], Y[m], acc);
    }
}

// Cache policy of the record loads: nt (aux bit 1).  The records are a read-once stream of 0.5 GB; with the default
// policy they displace each other and the hash table in L2 / MALL.  Measured on the headline step (tools/replay.py, live
// tile share 0.65, 514 MB): loads only 142 -> 115 us, the kernel 152 -> 135 us; a hand-written read stream does 6.3 TB/s
// with the default policy and 7.0 with nt (tools/micro/stream.hip).
#ifndef W16_LOAD_AUX
#define W16_LOAD_AUX 2
#endif
// the four 16-byte pieces of row tile `rt` of matrix `mat` of a wave tile's accumulator-image record -> vals[q][u]
__device__ __forceinline__ void load_tile_rows(srd_t rec, int mat, int rt, uint32_t lane16, f32x8 (&vals)[2]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#ifdef W16_DBG_NO_LOADS         // diagnosis builds: compute only
        float4 v = make_float4((float)lane16, 1.0f, (float)mat, (float)rt);
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
#else
        const float4 v = buf_load16_aux<W16_LOAD_AUX>(rec, lane16, (uint32_t)(mat * 16 + rt * 4 + g) * 1024u);
#endif
        vals[g >> 1][4 * (g & 1) + 0] = v.x, vals[g >> 1][4 * (g & 1) + 1] = v.y;
        vals[g >> 1][4 * (g & 1) + 2] = v.z, vals[g >> 1][4 * (g & 1) + 3] = v.w;
    }
#ifdef W16_ABL_MORE_BYTES   // ablation: two more 16-byte pieces per lane (a record of three bf16 planes is 96 bytes where fp32 is 64)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float4 v = buf_load16_aux<W16_LOAD_AUX>(rec, lane16, (uint32_t)(mat * 16 + ((rt + 1) & 3) * 4 + g) * 1024u);
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    }
#endif
}

__device__ __forceinline__ void zero_tile(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.0f;
}

// accumulator tile -> partial record; row_of(i) / col_of(j) map the tile's row (lane of X') and column (lane of Y')
// to indices of the gradient matrix or -1
template <typename RowFn, typename ColFn>
__device__ __forceinline__ void flush_mapped(float* __restrict__ rec, int base, int in_dim, int lane, const f32x16& acc,
                                             float mul, RowFn row_of, ColFn col_of) {
    const int jj = lane & 31, hh = lane >> 5;
    const int col = col_of(jj);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_of(rowmap(r, hh));
        if (row >= 0 && col >= 0) rec[base + row * in_dim + col] = acc[r] * mul;
    }
}

constexpr int W16_BLOCK = 512;
constexpr int W16_MAX_BLOCKS = 256;      // = WG_MAX_BLOCKS of decoder.hip: the partial buffer holds that many records

struct W16Args {
    const float* __restrict__ feat;
    const float* __restrict__ x;
    const float* __restrict__ saved;
    const float* __restrict__ dact;
    const float* __restrict__ dsmall;
    float* __restrict__ rec;            // this block's partial record
    uint32_t M, n_tiles;                // n_tiles: tiles to visit (= all list entries when there are lists)
    const uint32_t* live;               // the live-tile buffer of the chain kernel (decoder16.hip), or null = every tile
    uint32_t live_cap;                  // capacity of one of its eight lists
    uint32_t live_start[8];             // first visit index of each list
    const h8* w1_hi;                    // LDS copies of the forward's layer-1 operand images (recompute variant: P planes of
    const h8* w1_lo;                    // RT_F1 x T16_F1 x 64 entries each, plane 0 first), else null
    // lean gradient record (MIPSF_WGRAD_LEAN_DACT, exchange form only): dG3 and the rgb_emb half of dH2 are recomputed
    const h8* gimg;                     // LDS: [S2T hi: 4 row tiles][S2T lo: 4][RGBT hi: 2][RGBT lo: 2] x 64 operands, or null
    const uint2* masks;                 // the ReLU mask part of `saved`
};

// the it-th tile of a pass, last first: the records the chain kernel wrote last are still in the 256 MB Infinity Cache
__device__ __forceinline__ uint32_t w16_tile(const W16Args& a, uint32_t it) {
    const uint32_t k = a.n_tiles - 1 - it;
    if (!a.live) return k;
    uint32_t q = 0, first = 0;
#pragma unroll
    for (uint32_t j = 1; j < 8; ++j)
        if (k >= a.live_start[j]) q = j, first = a.live_start[j];
    return a.live[TL_HEADER + q * a.live_cap + (k - first)];
}

// column of the e products: slot t = 16 ct + 8 (c >> 4) + 4 ((c >> 3) & 1) + (c & 3), half (c >> 2) & 1
__device__ __forceinline__ int w16_e_col(int ct, int c) {
    return eidx(16 * ct + 8 * (c >> 4) + 4 * ((c >> 3) & 1) + (c & 3), (c >> 2) & 1);
}
// small row (0..4 d logits, 5..7 d rgb) held by X' lane i: i in {0..3, 8..11} -> 0..7
__device__ __forceinline__ int w16_small_row(int i) { return (i < 4) ? i : ((i >= 8 && i < 12) ? i - 4 : -1); }

// The 16 e slots 16 ct .. 16 ct + 15 of this lane's half (load_e<true>'s values: slot 8 d + k = sin(2^k pi x_d [+ pi/2]),
// slots 24, 25 the raw coordinates, the rest padding) as the two A operands of a transposing MFMA pair
template <int CT>
__device__ __forceinline__ void w16_e_tile(float x0, float x1, float x2, int h, f32x8 (&v)[2]) {
    const float ph = h ? HALF_PI_F : 0.0f;
    if (CT == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[0][k] = sin_reduced(fmaf(ldexpf(x0, k), PI_F, ph));
            v[1][k] = sin_reduced(fmaf(ldexpf(x1, k), PI_F, ph));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[0][k] = sin_reduced(fmaf(ldexpf(x2, k), PI_F, ph)), v[1][k] = 0.0f;
        v[1][0] = h ? x1 : x0;
        v[1][1] = h ? 0.0f : x2;
    }
}

#define W16_FENCE() __builtin_amdgcn_sched_barrier(0)

// largest |value| of a wave's 16-register block, as the bit pattern of a non-negative float in a scalar register
__device__ __forceinline__ uint32_t w16_block_max_bits(const f32x8 (&v)[2], int nq) {
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (q < nq) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) m = fmaxf(fmaxf(m, fabsf(v[q][u])), fabsf(v[q][u + 1]));
        }
    int x = __float_as_int(m);          // non-negative floats order like their bit patterns
    // DPP reduction: rows of 16 lanes (shift right by 1, 2, 4, 8), then lane 15 of a row into the next row, then lane 31
    // into rows 2 and 3: lane 63 ends up with the maximum of all 64
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// The power-of-two exponent k a gradient block is multiplied by (f16 arithmetic): kept while the block's largest entry
// x 2^k stays in [2^8, 2^15), else moved so that it lands in [2^14, 2^15); `rescale` = 2^(k_new - k_old) for the
// accumulators that live in the block's scaled domain (1.0f: nothing to do).  All in scalar registers.
__device__ __forceinline__ float w16_pick_scale(uint32_t max_bits, int& k, float& rescale) {
    rescale = 1.0f;
    const int eb = (int)(max_bits >> 23);                  // biased exponent: max in [2^(eb - 127), 2^(eb - 126))
    if (eb != 0 && eb != 255) {
        const int top = eb - 126 + k;                      // max x 2^k < 2^top
        if (top > 15 || top < 9) {
            int kn = 15 - (eb - 126);
            kn = kn > 60 ? 60 : (kn < -60 ? -60 : kn);
            rescale = __uint_as_float((uint32_t)(127 + kn - k) << 23);
            k = kn;
        }
    }
    return __uint_as_float((uint32_t)(127 + k) << 23);
}
__device__ __forceinline__ float w16_unscale(int k) { return __uint_as_float((uint32_t)(127 - k) << 23); }

// Both roles are software pipelined by hand, TWO stages deep: the loads of stage k + 2 are issued before stage k
// computes (three 16-register buffers in rotation; the stage counts, 9 and 6, are multiples of three so that a tile ends
// with the first two stages of the wave's next tile in the buffers the loop expects them in), the fences keep the
// compiler from sinking the loads back to their use.  One stage ahead leaves 32 KB per CU in flight -- 4 TB/s at the
// loaded HBM latency, which is what the kernel then ran at; without any prefetch a wave waits out a full latency per stage.
//
// waves 0..3 (row tile w): d w_pts2[w][0..3] = dH2[w]^T H1, d b_pts2;  one more accumulator tile with
//   rows 0..15 : (small rows)^T H3[col tile w]                                          -> d w_sdf2
//   rows 16..31: (small rows)^T {rgb_emb col tile 0 | 1 | e col tile 0 | 1}[w]          -> d w_rgb0
template <int LAYOUT, typename A>
__device__ __forceinline__ void w16_role_a(const W16Args& a, const typename A::v8 (&I)[2], int w, int lane) {
    constexpr int P = A::P;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) zero_tile(acc[t]);
    float bsum = 0.f, bsmall = 0.f, dummy = 0.f;
    int k_main = 0, k_small = 0;         // f16: exponents of the scaled domains of acc[0..3] + bsum / acc[4] + bsmall
    f32x8 buf[3][2];
    auto act_srd = [&](const float* recs, uint32_t tile) {
        return make_srd(recs + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);
    };
    // small rows: lane (j, h = 0) contributes dsmall[s][0..7] to MFMA 0 -> columns {0..3, 8..11}; the other half and
    // the samples past M read zeros through the buffer's bounds check (no branch)
    const srd_t small_srd = make_srd(a.dsmall, a.M * 32u), x_srd = make_srd(a.x, a.M * 12u);
    auto load_small = [&](uint32_t tile, f32x8 (&v)[2]) {
        const uint32_t off = h == 0 ? (tile * 32u + (uint32_t)j) * 32u : 0xfffffff0u;
        const float4 p = buf_load16(small_srd, off, 0), q = buf_load16(small_srd, off, 16);
        v[0][0] = p.x, v[0][1] = p.y, v[0][2] = p.z, v[0][3] = p.w, v[0][4] = q.x, v[0][5] = q.y, v[0][6] = q.z, v[0][7] = q.w;
    };
    auto load_x = [&](uint32_t tile, f32x8 (&v)[2]) {
        const uint32_t s_raw = tile * 32u + (uint32_t)j;
        const uint32_t off = (s_raw < a.M ? s_raw : a.M - 1) * 12u;
#pragma unroll
        for (int d = 0; d < 3; ++d) v[0][d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_srd, off, 4 * d, 0));
    };
    // tiles in reverse: the records the chain kernel wrote last are still in the 256 MB Infinity Cache
    uint32_t it = blockIdx.x;
    if (it < a.n_tiles) {
        load_tile_rows(act_srd(a.dact, w16_tile(a, it)), 1, w, lane16, buf[0]);
        load_tile_rows(act_srd(a.saved, w16_tile(a, it)), 0, 0, lane16, buf[1]);
    }
#pragma clang loop unroll(disable)
    for (; it < a.n_tiles; it += gridDim.x) {
        const uint32_t tile = w16_tile(a, it);
#ifndef W16_NO_TILE_BARRIER
        // The 8 waves read each other's records (H1 by all of waves 0..3, H2 by waves 4..7 and 0, 1): kept within one tile
        // of each other, the second to fourth reader hits in L2; free-running, they drift apart by whole tiles and the
        // re-reads go back to memory.
        __builtin_amdgcn_s_barrier();
#endif
        const srd_t sa = act_srd(a.saved, tile);
        const uint32_t nt = it + gridDim.x < a.n_tiles ? w16_tile(a, it + gridDim.x) : tile;
        typename A::v8 X[P][2];
        load_tile_rows(sa, 0, 1, lane16, buf[2]);
        W16_FENCE();
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(buf[0], 2), k_main, rs);
            if (rs != 1.0f) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] *= rs;
                bsum *= rs;
            }
            buf[0][0] *= sx, buf[0][1] *= sx;
        }
        transpose_block<A, true>(buf[0], I, X, bsum);                   // stage 0: X = dH2[w]
        W16_FENCE();
        load_tile_rows(sa, 0, 2, lane16, buf[0]);
        W16_FENCE();
        transpose_mac<A>(buf[1], I, X, acc[0]);                         // stages 1..4: H1 column tiles
        W16_FENCE();
        load_tile_rows(sa, 0, 3, lane16, buf[1]);
        W16_FENCE();
        transpose_mac<A>(buf[2], I, X, acc[1]);
        W16_FENCE();
        load_small(tile, buf[2]);
        W16_FENCE();
        transpose_mac<A>(buf[0], I, X, acc[2]);
        W16_FENCE();
        load_tile_rows(sa, 2, w, lane16, buf[0]);
        W16_FENCE();
        transpose_mac<A>(buf[1], I, X, acc[3]);
        W16_FENCE();
        load_tile_rows(sa, 1, 2 + (w & 1), lane16, buf[1]);             // rgb_emb = H2 row tiles 2, 3 (read by waves 0, 1)
        W16_FENCE();
        f32x8 sv1[2], sv2[2];                                           // stage 5: X = small rows (columns 0..11)
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(buf[2], 1), k_small, rs);
            if (rs != 1.0f) acc[4] *= rs, bsmall *= rs;
            buf[2][0] *= sx;
        }
        sv1[0] = buf[2][0], sv2[0] = buf[2][0], sv1[1] = buf[2][0], sv2[1] = buf[2][0];      // ([1] is not read: NQ = 1)
        transpose_block<A, true, 1>(sv1, I, X, bsmall);                 // (its row sums are wave 0's to write)
        W16_FENCE();
        load_x(tile, buf[2]);                                           // coordinates (used by waves 2, 3)
        W16_FENCE();
        transpose_mac<A>(buf[0], I, X, acc[4]);                         // stage 6: H3[w]
        transpose_block<A, false, 1>(sv2, I, X, dummy, 1);              // the same small rows at columns 16..27
        W16_FENCE();
        load_tile_rows(act_srd(a.dact, nt), 1, w, lane16, buf[0]);      // next tile's stage 0
        W16_FENCE();
        if (w < 2) transpose_mac<A>(buf[1], I, X, acc[4]);              // stage 7: rgb_emb (waves 0, 1)
        W16_FENCE();
        load_tile_rows(act_srd(a.saved, nt), 0, 0, lane16, buf[1]);     // next tile's stage 1
        W16_FENCE();
        if (w >= 2) {                                                   // stage 8: e (waves 2, 3)
            const float x0 = buf[2][0][0], x1 = buf[2][0][1], x2 = buf[2][0][2];
            if (w == 2) w16_e_tile<0>(x0, x1, x2, h, buf[2]);
            else w16_e_tile<1>(x0, x1, x2, h, buf[2]);
            transpose_mac<A>(buf[2], I, X, acc[4]);
        }
        W16_FENCE();
    }
    float* rec = a.rec;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
        flush_mapped(rec, G_W_PTS2, HID, lane, acc[ct], w16_unscale(k_main), [&](int i) { return 32 * w + i; }, [&](int c) { return 32 * ct + c; });
    flush_mapped(rec, G_W_SDF2, HID, lane, acc[4], w16_unscale(k_small),
                 [&](int i) { const int r = w16_small_row(i); return r < N_CLASS ? r : -1; }, [&](int c) { return 32 * w + c; });
    auto rgb_row = [&](int i) { const int r = i >= 16 ? w16_small_row(i - 16) : -1; return r >= N_CLASS ? r - N_CLASS : -1; };
    if (w < 2)
        flush_mapped(rec, G_W_RGB0, N_RGB_IN, lane, acc[4], w16_unscale(k_small), rgb_row, [&](int c) { return 32 * w + c; });
    else
        flush_mapped(rec, G_W_RGB0, N_RGB_IN, lane, acc[4], w16_unscale(k_small), rgb_row,
                     [&](int c) { const int e = w16_e_col(w - 2, c); return e >= 0 ? N_EMB + e : -1; });
    // bias partials: this lane = feature (lane & 31) of the row tile; the two halves hold different samples
    const float b2 = (bsum + __shfl_xor(bsum, 32, 64)) * w16_unscale(k_main);
    if (h == 0) rec[G_B_PTS2 + 32 * w + j] = b2;
    if (w == 0) {
        const float bs = (bsmall + __shfl_xor(bsmall, 32, 64)) * w16_unscale(k_small);
        const int r = w16_small_row(j);
        if (h == 0 && r >= 0 && r < N_CLASS) rec[G_B_SDF2 + r] = bs;
        if (h == 0 && r >= N_CLASS) rec[G_B_RGB0 + r - N_CLASS] = bs;
    }
}

// Role A of the LEAN record (mipsf_decoder_fwd16: H1 is not stored).  H1 is RECOMPUTED, directly in the layout the product
// wants: the forward evaluates H1^T = W1 e^T with the weight image as A operand; with the operands swapped the same
// instruction yields H1 = e W1^T -- lane = feature, registers = 16 samples -- from the SAME image (an A-operand image of
// W1's rows is a B-operand image of W1^T's columns) and the same e operands, products in the same order: the forward's H1,
// bit for bit, 12 MFMAs per 32 features instead of a 4 KB load + 4 transposing MFMAs.  The images (hi + lo, 32 KB) sit in
// LDS.  Every load of a tile has its own buffer and is issued a whole tile ahead.
template <int LAYOUT, typename A>
__device__ __forceinline__ void w16_role_a_recompute(const W16Args& a, const typename A::v8 (&I)[2], int w, int lane) {
    static_assert(A::SCALED && A::P == 2, "f16 hi/lo arithmetic only");
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) zero_tile(acc[t]);
    float bsum = 0.f, bsmall = 0.f, dummy = 0.f;
    int k_main = 0, k_small = 0;
    auto act_srd = [&](const float* recs, uint32_t tile) {
        return make_srd(recs + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);
    };
    const srd_t small_srd = make_srd(a.dsmall, a.M * 32u), x_srd = make_srd(a.x, a.M * 12u);
    auto load_small = [&](uint32_t tile, f32x8 (&v)[2]) {
        const uint32_t off = h == 0 ? (tile * 32u + (uint32_t)j) * 32u : 0xfffffff0u;
        const float4 p = buf_load16(small_srd, off, 0), q = buf_load16(small_srd, off, 16);
        v[0][0] = p.x, v[0][1] = p.y, v[0][2] = p.z, v[0][3] = p.w, v[0][4] = q.x, v[0][5] = q.y, v[0][6] = q.z, v[0][7] = q.w;
    };
    auto load_x = [&](uint32_t tile, float (&v)[3]) {
        const uint32_t s_raw = tile * 32u + (uint32_t)j;
        const uint32_t off = (s_raw < a.M ? s_raw : a.M - 1) * 12u;
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_srd, off, 4 * d, 0));
    };
    f32x8 bX[2], bS[2], bH3[2], bE[2];
    float xv[3];
    uint32_t it = blockIdx.x;
    if (it < a.n_tiles) {
        const uint32_t t0 = w16_tile(a, it);
        load_tile_rows(act_srd(a.dact, t0), 1, w, lane16, bX);
        load_x(t0, xv);
        load_small(t0, bS);
        load_tile_rows(act_srd(a.saved, t0), 2, w, lane16, bH3);
        load_tile_rows(act_srd(a.saved, t0), 1, 2 + (w & 1), lane16, bE);
    }
#pragma clang loop unroll(disable)
    for (; it < a.n_tiles; it += gridDim.x) {
        const uint32_t tile = w16_tile(a, it);
#ifndef W16_NO_TILE_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        const uint32_t nt = it + gridDim.x < a.n_tiles ? w16_tile(a, it + gridDim.x) : tile;
        const srd_t nsa = act_srd(a.saved, nt);
        const float x0 = xv[0], x1 = xv[1], x2 = xv[2];
        typename A::v8 X[2][2];
        // ---- the small-row products first: their three buffers are free again before the long H1 phase starts
        f32x8 sv1[2], sv2[2];
        {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bS, 1), k_small, rs);
            if (rs != 1.0f) acc[4] *= rs, bsmall *= rs;
            bS[0] *= sx;
        }
        sv1[0] = bS[0], sv2[0] = bS[0], sv1[1] = bS[0], sv2[1] = bS[0];
        transpose_block<A, true, 1>(sv1, I, X, bsmall);                             // X = small rows (columns 0..11)
        W16_FENCE();
        transpose_mac<A>(bH3, I, X, acc[4]);                                        // H3[w] -> rows 0..15
        transpose_block<A, false, 1>(sv2, I, X, dummy, 1);                          // the small rows at columns 16..27
        W16_FENCE();
        if (w < 2) {
            transpose_mac<A>(bE, I, X, acc[4]);                                     // rgb_emb (waves 0, 1)
        } else {
            f32x8 ev[2];
            if (w == 2) w16_e_tile<0>(x0, x1, x2, h, ev);
            else w16_e_tile<1>(x0, x1, x2, h, ev);
            transpose_mac<A>(ev, I, X, acc[4]);                                     // e (waves 2, 3)
        }
        W16_FENCE();
        // ---- X = dH2[w]
        {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bX, 2), k_main, rs);
            if (rs != 1.0f) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] *= rs;
                bsum *= rs;
            }
            bX[0] *= sx, bX[1] *= sx;
        }
        transpose_block<A, true>(bX, I, X, bsum);
        W16_FENCE();
        // ---- e as the forward's layer-1 operand: 4 k-steps of 8 slots per half, bias ones in slots 26, 27
        typename A::v8 eh[4], el[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {            // one k-step at a time (pinned: 24 interleaved sines need 60 temporaries)
            f32x8 ev;
            if (t < 3) {
                const float xd = t == 0 ? x0 : (t == 1 ? x1 : x2);
#pragma unroll
                for (int k = 0; k < 8; ++k) ev[k] = sin_reduced(fmaf(ldexpf(xd, k), PI_F, h ? HALF_PI_F : 0.0f));
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) ev[k] = 0.0f;
                ev[0] = h ? x1 : x0, ev[1] = h ? 0.0f : x2;                  // slots 24, 25: the raw coordinates
                ev[BIAS16_U] = 1.0f, ev[BIAS16_U + 1] = 1.0f;                // slots 26, 27 meet the bias halves of the image
            }
            eh[t] = next_plane<A, false>(ev), el[t] = next_plane<A, true>(ev);
            asm volatile("" : "+v"(eh[t]), "+v"(el[t]));
            W16_FENCE();
        }
        // the next tile's small-row operands have the whole H1 phase to arrive
        load_small(nt, bS);
        load_tile_rows(nsa, 2, w, lane16, bH3);
        load_tile_rows(nsa, 1, 2 + (w & 1), lane16, bE);
        W16_FENCE();
        // ---- H1 column tiles, recomputed, multiplied at once
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            f32x16 hacc = zero;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const h8 wh = a.w1_hi[(ct * T16H_F1 + t) * 64 + lane], wl = a.w1_lo[(ct * T16_F1 + t) * 64 + lane];
                hacc = mfma16(eh[t], wh, hacc);
                hacc = mfma16(el[t], wh, hacc);
                hacc = mfma16(eh[t], wl, hacc);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {                   // (k-step by k-step: one pair of operand planes alive)
                f32x8 r;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    r[u] = __builtin_amdgcn_fmed3f(hacc[8 * m + u] * (1.0f / (float)(1 << W16_SHIFT)), 0.0f, __builtin_inff());
                const typename A::v8 yh = next_plane<A, false>(r);
                const typename A::v8 yl = next_plane<A, true>(r);
                acc[ct] = mfma16(X[0][m], yh, acc[ct]);
                acc[ct] = mfma16(X[1][m], yh, acc[ct]);
                acc[ct] = mfma16(X[0][m], yl, acc[ct]);
            }
            W16_FENCE();
        }
        // the next tile's dH2 and coordinates: needed after its small-row stages
        load_tile_rows(act_srd(a.dact, nt), 1, w, lane16, bX);
        load_x(nt, xv);
        W16_FENCE();
    }
    float* rec = a.rec;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
        flush_mapped(rec, G_W_PTS2, HID, lane, acc[ct], w16_unscale(k_main), [&](int i) { return 32 * w + i; }, [&](int c) { return 32 * ct + c; });
    flush_mapped(rec, G_W_SDF2, HID, lane, acc[4], w16_unscale(k_small),
                 [&](int i) { const int r = w16_small_row(i); return r < N_CLASS ? r : -1; }, [&](int c) { return 32 * w + c; });
    auto rgb_row = [&](int i) { const int r = i >= 16 ? w16_small_row(i - 16) : -1; return r >= N_CLASS ? r - N_CLASS : -1; };
    if (w < 2)
        flush_mapped(rec, G_W_RGB0, N_RGB_IN, lane, acc[4], w16_unscale(k_small), rgb_row, [&](int c) { return 32 * w + c; });
    else
        flush_mapped(rec, G_W_RGB0, N_RGB_IN, lane, acc[4], w16_unscale(k_small), rgb_row,
                     [&](int c) { const int e = w16_e_col(w - 2, c); return e >= 0 ? N_EMB + e : -1; });
    const float b2 = (bsum + __shfl_xor(bsum, 32, 64)) * w16_unscale(k_main);
    if (h == 0) rec[G_B_PTS2 + 32 * w + j] = b2;
    if (w == 0) {
        const float bs = (bsmall + __shfl_xor(bsmall, 32, 64)) * w16_unscale(k_small);
        const int r = w16_small_row(j);
        if (h == 0 && r >= 0 && r < N_CLASS) rec[G_B_SDF2 + r] = bs;
        if (h == 0 && r >= N_CLASS) rec[G_B_RGB0 + r - N_CLASS] = bs;
    }
}

// waves 4..7 (row tile rt): d w_sdf0[rt][0..2] = dG3[rt]^T [sdf_emb | grid], d b_sdf0;  d w_pts0[rt][0..1] = dG1[rt]^T e,
// d b_pts0 (e recomputed from x)
template <int LAYOUT, typename A>
__device__ __forceinline__ void w16_role_b(const W16Args& a, const typename A::v8 (&I)[2], int rt, int lane) {
    constexpr int P = A::P;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) zero_tile(acc[t]);
    float bsum0 = 0.f, bsum1 = 0.f;
    int k3 = 0, k1 = 0;                  // f16: exponents of the scaled domains of acc[0..2] + bsum0 / acc[3..4] + bsum1
    f32x8 buf[3][2];
    auto act_srd = [&](const float* recs, uint32_t tile) {
        return make_srd(recs + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);
    };
    // grid features and coordinates through buffer resources: one 32-bit lane offset instead of 16 address pairs
    const uint64_t feat_bytes = (uint64_t)a.M * N_GRID * 4;
    const srd_t feat_srd = make_srd(a.feat, feat_bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)feat_bytes);
    const srd_t x_srd = make_srd(a.x, a.M * 12u);
    auto sample_of = [&](uint32_t tile) {
        const uint32_t s_raw = tile * 32u + (uint32_t)j;
        return s_raw < a.M ? s_raw : a.M - 1;
    };
    // grid features: this lane holds feature h of level 8 q + u  ->  column 16 q + 8 (u >> 2) + 4 h + (u & 3)
    auto load_grid = [&](uint32_t tile, f32x8 (&v)[2]) {
        const uint32_t s_c = sample_of(tile);
        const uint32_t voff = LAYOUT == MIPSF_FEAT_AOS ? s_c * (uint32_t)(N_GRID * 4) + 4u * (uint32_t)h : (s_c * 2u + (uint32_t)h) * 4u;
        const uint32_t lstride = LAYOUT == MIPSF_FEAT_AOS ? 8u : a.M * 8u;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[q][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(feat_srd, voff, (uint32_t)(8 * q + u) * lstride, 0));
    };
    auto load_x = [&](uint32_t tile, f32x8 (&v)[2]) {
        const uint32_t off = sample_of(tile) * 12u;
#pragma unroll
        for (int d = 0; d < 3; ++d) v[0][d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_srd, off, 4 * d, 0));
    };
    uint32_t it = blockIdx.x;
    if (it < a.n_tiles) {
        load_tile_rows(act_srd(a.dact, w16_tile(a, it)), 2, rt, lane16, buf[0]);
        load_tile_rows(act_srd(a.saved, w16_tile(a, it)), 1, 0, lane16, buf[1]);
    }
#pragma clang loop unroll(disable)
    for (; it < a.n_tiles; it += gridDim.x) {
        const uint32_t tile = w16_tile(a, it);
#ifndef W16_NO_TILE_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        const srd_t sa = act_srd(a.saved, tile), da = act_srd(a.dact, tile);
        const uint32_t nt = it + gridDim.x < a.n_tiles ? w16_tile(a, it + gridDim.x) : tile;
        typename A::v8 X[P][2];
        load_tile_rows(sa, 1, 1, lane16, buf[2]);
        W16_FENCE();
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(buf[0], 2), k3, rs);
            if (rs != 1.0f) {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] *= rs;
                bsum0 *= rs;
            }
            buf[0][0] *= sx, buf[0][1] *= sx;
        }
        transpose_block<A, true>(buf[0], I, X, bsum0);                  // stage 0: X = dG3[rt]
        W16_FENCE();
        load_grid(tile, buf[0]);
        W16_FENCE();
        transpose_mac<A>(buf[1], I, X, acc[0]);                         // stages 1, 2: sdf_emb = H2 row tiles 0, 1
        W16_FENCE();
        load_tile_rows(da, 0, rt, lane16, buf[1]);
        W16_FENCE();
        transpose_mac<A>(buf[2], I, X, acc[1]);
        W16_FENCE();
        load_x(tile, buf[2]);
        W16_FENCE();
        if (A::SCALED) buf[0][0] *= W16_GRID_SHIFT, buf[0][1] *= W16_GRID_SHIFT;
        transpose_mac<A>(buf[0], I, X, acc[2]);                         // stage 3: grid
        W16_FENCE();
        load_tile_rows(act_srd(a.dact, nt), 2, rt, lane16, buf[0]);     // next tile's stage 0
        W16_FENCE();
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(buf[1], 2), k1, rs);
            if (rs != 1.0f) acc[3] *= rs, acc[4] *= rs, bsum1 *= rs;
            buf[1][0] *= sx, buf[1][1] *= sx;
        }
        transpose_block<A, true>(buf[1], I, X, bsum1);                  // stage 4: X = dG1[rt]
        W16_FENCE();
        load_tile_rows(act_srd(a.saved, nt), 1, 0, lane16, buf[1]);     // next tile's stage 1
        W16_FENCE();
        {                                                               // stage 5: e, both column tiles
            const float x0 = buf[2][0][0], x1 = buf[2][0][1], x2 = buf[2][0][2];
            w16_e_tile<0>(x0, x1, x2, h, buf[2]);
            transpose_mac<A>(buf[2], I, X, acc[3]);
            W16_FENCE();
            w16_e_tile<1>(x0, x1, x2, h, buf[2]);
            transpose_mac<A>(buf[2], I, X, acc[4]);
        }
        W16_FENCE();
    }
    float* rec = a.rec;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        flush_mapped(rec, G_W_SDF0, N_SDF_IN, lane, acc[ct], w16_unscale(k3), [&](int i) { return 32 * rt + i; }, [&](int c) { return 32 * ct + c; });
    // grid columns: level = 8 (c >> 4) + 4 ((c >> 3) & 1) + (c & 3), feature (c >> 2) & 1
    flush_mapped(rec, G_W_SDF0, N_SDF_IN, lane, acc[2], A::SCALED ? w16_unscale(k3) / W16_GRID_SHIFT : 1.0f, [&](int i) { return 32 * rt + i; },
                 [&](int c) { return N_EMB + 2 * (8 * (c >> 4) + 4 * ((c >> 3) & 1) + (c & 3)) + ((c >> 2) & 1); });
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        flush_mapped(rec, G_W_PTS0, N_E, lane, acc[3 + ct], w16_unscale(k1), [&](int i) { return 32 * rt + i; },
                     [&](int c) { return w16_e_col(ct, c); });
    const float b3 = (bsum0 + __shfl_xor(bsum0, 32, 64)) * w16_unscale(k3), b1 = (bsum1 + __shfl_xor(bsum1, 32, 64)) * w16_unscale(k1);
    if (h == 0) rec[G_B_SDF0 + 32 * rt + j] = b3, rec[G_B_PTS0 + 32 * rt + j] = b1;
}


// ============================================================================ the EXCHANGE form (lean record, f16 hi/lo)
// In the roles above every wave prepares all the operands its products need, and most of them are needed by four waves:
// the four H1 column tiles (each of waves 0..3 recomputed all four: 48 of its ~100 MFMAs), the [sdf_emb | grid] and e
// column tiles (transposed by each of waves 4..7), and the positional encoding itself (24 sines per lane, evaluated by all
// eight waves).  Per SIMD and tile that came to 1504 vector + 158 matrix instructions; the kernel was bound by instruction
// issue (~110 us at the 1.5 GHz the device sustains under this load, 141 us measured), not by its 0.4 GB of reads.
// Here every shared operand is prepared ONCE per tile, by one wave, and handed to the others through LDS as ready MFMA
// operand planes (16-byte pieces, lane-linear: conflict-free writes and reads), two buffers in rotation, ONE barrier per tile:
//     XE  e as the forward's layer-1 operand: k-step t (hi, lo) by wave t          (for tile i + 1, written during tile i)
//     XA  H1 column tile ct as the product's right-hand planes (yh0, yh1, yl0, yl1) by wave ct
//     XB  transposed column tiles: sdf_emb 0, sdf_emb 1 (waves 4, 5), grid (wave 6), e 0, e 1 (wave 7, from XE's planes)
// before the barrier a wave produces, transposes its own gradient blocks (unique to it) and reads nothing of this tile's
// XA / XB; after it, it multiplies.
#ifdef W16_TRACE     // diagnosis builds (tools/micro/wgrad_probe.py): cycles between the marks of a tile, summed per wave
__device__ unsigned long long w16_trace[2048 * 16];
#define W16_MARK(k) do { __builtin_amdgcn_sched_barrier(0); tr_t[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W16_TRACE_DECL unsigned long long tr_t[8]
#define W16_TRACE_SUM(n, wave)                                                                                    \
    do {                                                                                                          \
        if (lane == 0) {                                                                                          \
            const unsigned wi = (blockIdx.x * 8u + (unsigned)(wave)) & 2047u;                                     \
            for (int k = 0; k < (n); ++k) w16_trace[wi * 16 + k] += tr_t[k + 1] - tr_t[k];                        \
            w16_trace[wi * 16 + 15] += 1ull;                                                                      \
        }                                                                                                         \
    } while (0)
#else
#define W16_MARK(k) do { } while (0)
#define W16_TRACE_DECL do { } while (0)
#define W16_TRACE_SUM(n, wave) do { } while (0)
#endif
// ---- the lean gradient record: what the chain kernel (decoder16.hip, decoder16_bwd_tile) computed and did not store, with its
// own operations in its own order -- bit for bit what it would have stored.
// sm: (d logit 0..4, d rgb 0..2) of this lane's sample.  The chain scales every sample's gradients by a power of two `up` that
// brings the largest of them to [0.5, 1) and stores results multiplied by `down` = 1 / up.
__device__ __forceinline__ void w16x_updown(const f32x8& sm, float& up, float& down) {
    float mx = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) mx = fmaxf(mx, fabsf(sm[c]));
    up = 1.0f, down = 1.0f;
    if (mx > 0.0f && mx < 3.0e38f) {
        const int e = max(__builtin_amdgcn_frexp_expf(mx), -126);      // (subnormal maximum: decoder16.hip, RANGE)
        up = ldexpf(1.0f, -e), down = ldexpf(1.0f, e);
    }
}
// the narrow product's B operand: half 0 carries the n values sm[first .. first + n) x up in elements 0 .. n - 1, half 1 zeros
template <typename A>
__device__ __forceinline__ void w16x_small_operand(const f32x8& sm, int first, int n, float up, int h, typename A::v8 (&b)[A::P]) {
    f32x8 v;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (h == 0 && u < n) ? sm[(first + u) & 7] * up : 0.0f;
#pragma unroll
    for (int p = 0; p < A::P; ++p) b[p] = p == A::P - 1 ? next_plane<A, true>(v) : next_plane<A, false>(v);
}
// one row tile of a one-k-step product, in the CHAIN kernel's order (decoder16.hip, mfma16_layer): a = weight planes (LDS),
// b = the small operand's planes.  P = 2: a0 b0 + a0 b1 + a1 b0;  P = 3: a0 b0, a0 b1, a0 b2, a1 b1, a1 b0, a2 b0.
template <typename A>
__device__ __forceinline__ f32x16 w16x_narrow(const typename A::v8* img, int entry, int plane_stride, const typename A::v8 (&b)[A::P]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const typename A::v8 a0 = img[entry], a1 = img[plane_stride + entry];
    f32x16 acc = mfma16(a0, b[0], zero);
    acc = mfma16(a0, b[1], acc);
    if constexpr (A::P == 3) {
        const typename A::v8 a2 = img[2 * plane_stride + entry];
        acc = mfma16(a0, b[2], acc);
        acc = mfma16(a1, b[1], acc);
        acc = mfma16(a1, b[0], acc);
        acc = mfma16(a2, b[0], acc);
    } else {
        acc = mfma16(a1, b[0], acc);
    }
    return acc;
}
// LDS layout of the exchange form, in 16-byte entries (P planes):
//   gimg   the chain's two narrow products' operand images: S2T plane p, row tile rt at (4 p + rt) x 64; RGBT plane p, row tile
//          q at (4 P + 2 p + q) x 64
//   XE     e as the forward's layer-1 operand: k-step t, plane p at (t P + p) x 64
//   XA/XB  column tile ct: plane p, k-step m at (ct 2 P + 2 p + m) x 64
// P = 2 (f16 hi / lo): every hand-over buffer exists TWICE (tile parity) and a tile needs ONE barrier; 134 KB.  P = 3 (bf16):
// 201 KB that way -- single buffers and a SECOND barrier at the end of a tile instead (132 KB): XE for the next tile is then
// written in the multiply phase, behind the first barrier.
template <typename A>
struct W16XL {
    static constexpr int P = A::P;
    static constexpr int G_ENTRIES = (4 * P + 2 * P) * 64;
    static constexpr int G_PLANE_S2T = 4 * 64, G_RGBT = 4 * P * 64, G_PLANE_RGBT = 2 * 64;
    static constexpr int XE = 4 * P * 64, XA = 4 * 2 * P * 64, XB = 5 * 2 * P * 64, CT = 2 * P * 64;
    static constexpr int NBUF = P == 3 ? 1 : 2;
};

template <typename A>
struct W16X {
    typename A::v8* xe;      // [NBUF][4 k-steps][P planes][64 lanes]
    typename A::v8* xa;      // [NBUF][4 column tiles][2 P][64]
    typename A::v8* xb;      // [NBUF][5 blocks][2 P][64]
};

// all LDS writes of this wave done, then the workgroup barrier (NOT __syncthreads: that also waits for every outstanding
// global load, and the next tile's records are in flight here on purpose)
__device__ __forceinline__ void w16x_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// k-step t of e (the forward's layer-1 B operand: 8 slots per half; t = 3: raw coordinates + the bias ones) as P planes
template <typename A>
__device__ __forceinline__ void w16x_e_step(int t, float x0, float x1, float x2, int h, typename A::v8 (&e)[A::P]) {
#ifdef W16_DBG_NO_COMPUTE
    {
        f32x8 c;
        for (int k = 0; k < 8; ++k) c[k] = x0 + x1 + x2;
        for (int p = 0; p < A::P; ++p) e[p] = next_plane<A, true>(c);
        return;
    }
#endif
    f32x8 ev;
    if (t < 3) {
        const float xd = t == 0 ? x0 : (t == 1 ? x1 : x2);
#pragma unroll
        for (int k = 0; k < 8; ++k) ev[k] = sin_reduced(fmaf(ldexpf(xd, k), PI_F, h ? HALF_PI_F : 0.0f));
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) ev[k] = 0.0f;
        ev[0] = h ? x1 : x0, ev[1] = h ? 0.0f : x2;                  // slots 24, 25: the raw coordinates
#pragma unroll
        for (int k = 0; k < A::P; ++k) ev[BIAS16_U + k] = 1.0f;      // slots 26.. meet the bias pieces of the image (2 or 3)
    }
#pragma unroll
    for (int p = 0; p < A::P; ++p) e[p] = p == A::P - 1 ? next_plane<A, true>(ev) : next_plane<A, false>(ev);
}

// acc += X^T Y for ready planes: y = one column tile of XA / XB (plane pb, k-step m at (2 pb + m) x 64); the plane pairs with
// pa + pb <= P - 1, k-step by k-step
template <typename A>
__device__ __forceinline__ void w16x_mac(const typename A::v8 (&X)[A::P][2], const typename A::v8* y, int lane, f32x16& acc) {
#ifdef W16_DBG_NO_COMPUTE
    return;
#endif
    constexpr int P = A::P;
    typename A::v8 Y[P][2];
#pragma unroll
    for (int pb = 0; pb < P; ++pb)
#pragma unroll
        for (int m = 0; m < 2; ++m) Y[pb][m] = y[(2 * pb + m) * 64 + lane];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pb = 0; pb < P; ++pb)
#pragma unroll
            for (int pa = 0; pa + pb < P; ++pa) acc = mfma16(X[pa][m], Y[pb][m], acc);
}
// two such products side by side (acc_a += Xa^T Ya, acc_b += Xb^T Yb), their MFMAs in turn: a chain of MFMAs on ONE
// accumulator issues every ~82 cycles instead of every 32 (each waits for the one in front of it); with two chains per wave
// and two waves per SIMD the matrix pipe always finds an independent instruction
template <typename A>
__device__ __forceinline__ void w16x_mac2(const typename A::v8 (&Xa)[A::P][2], const typename A::v8* ya, f32x16& acc_a,
                                          const typename A::v8 (&Xb)[A::P][2], const typename A::v8* yb, f32x16& acc_b, int lane) {
#ifdef W16_DBG_NO_COMPUTE
    return;
#endif
    constexpr int P = A::P;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pb = 0; pb < P; ++pb) {
            const typename A::v8 Ya = ya[(2 * pb + m) * 64 + lane], Yb = yb[(2 * pb + m) * 64 + lane];
#pragma unroll
            for (int pa = 0; pa + pb < P; ++pa) {
                acc_a = mfma16(Xa[pa][m], Ya, acc_a);
                acc_b = mfma16(Xb[pa][m], Yb, acc_b);
            }
        }
}
template <typename A>
__device__ __forceinline__ void w16x_put(typename A::v8* y, int lane, const typename A::v8 (&Y)[A::P][2]) {
#pragma unroll
    for (int p = 0; p < A::P; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m) y[(2 * p + m) * 64 + lane] = Y[p][m];
}

// waves 0..3 (w): d w_pts2[w][0..3] = dH2[w]^T H1, d b_pts2; rows 0..15 of the fifth tile (small rows)^T H3[w] -> d w_sdf2,
// rows 16..31 (small rows)^T {rgb_emb 0 | rgb_emb 1 | e 0 | e 1}[w] -> d w_rgb0.  Produces e k-step w and H1 column tile w.
template <int LAYOUT, typename A>
__device__ __forceinline__ void w16x_role_a(const W16Args& a, const W16X<A>& lx, const typename A::v8 (&I)[2], int w, int lane) {
    typedef W16XL<A> L;
    typedef typename A::v8 v8;
    constexpr int P = A::P;
    constexpr bool TWO_BARRIERS = L::NBUF == 1;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) zero_tile(acc[t]);
    float bsum = 0.f, bsmall = 0.f, dummy = 0.f;
    int k_main = 0, k_small = 0;
    auto act_srd = [&](const float* recs, uint32_t tile) {
        return make_srd(recs + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);
    };
    const srd_t small_srd = make_srd(a.dsmall, a.M * 32u), x_srd = make_srd(a.x, a.M * 12u);
    // small rows: BOTH halves read their sample's 8 values (the recomputation below scales by the sample); only half 0's copy
    // enters the small-row products (the other half's lanes contribute zeros)
    auto load_small = [&](uint32_t tile, f32x8 (&v)[2]) {
        const uint32_t off = (tile * 32u + (uint32_t)j) * 32u;
        const float4 p = buf_load16(small_srd, off, 0), q = buf_load16(small_srd, off, 16);
        v[0][0] = p.x, v[0][1] = p.y, v[0][2] = p.z, v[0][3] = p.w, v[0][4] = q.x, v[0][5] = q.y, v[0][6] = q.z, v[0][7] = q.w;
    };
    const v8* gimg = reinterpret_cast<const v8*>(a.gimg);
    const v8* w1 = reinterpret_cast<const v8*>(a.w1_hi);     // [P planes][RT_F1 x T16_F1 x 64]
    constexpr int W1_PLANE = RT_F1 * T16H_F1 * 64;
    const bool recompute_x = a.gimg != nullptr && w >= 2;       // lean gradient record: dH2[2], dH2[3] = Wrgb^T drgb are not stored
    auto dh2_srd = [&](uint32_t tile) {
        return make_srd(a.dact + (size_t)tile * ACT_TILE_FLOATS, recompute_x ? 0 : ACT_TILE_FLOATS * 4);
    };
    auto load_x = [&](uint32_t tile, float (&v)[3]) {
        const uint32_t s_raw = tile * 32u + (uint32_t)j;
        const uint32_t off = (s_raw < a.M ? s_raw : a.M - 1) * 12u;
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_srd, off, 4 * d, 0));
    };
    // Every wave issues the SAME loads in the same order, whether it needs them or not (waves 2, 3 have no rgb_emb tile: their
    // resource is empty, the loads return zeros without touching memory): the compiler counts outstanding loads per program
    // path, and where paths with different counts meet it waits for the shortest one's count -- a wave on a longer path then
    // waits for loads it has just issued (measured: 2700 instead of 900 cycles in the phase behind such a join).
    auto e_srd = [&](uint32_t tile) {
        return make_srd(a.saved + (size_t)tile * ACT_TILE_FLOATS, w < 2 ? ACT_TILE_FLOATS * 4 : 0);
    };
    auto put_e = [&](v8* xe, const float (&xv)[3]) {
        v8 e[P];
        w16x_e_step<A>(w, xv[0], xv[1], xv[2], h, e);
#pragma unroll
        for (int p = 0; p < P; ++p) xe[(w * P + p) * 64 + lane] = e[p];
    };
    f32x8 bX[2], bS[2], bH3[2], bE[2];
    float xn[3] = {0.f, 0.f, 0.f};           // coordinates of the NEXT tile (its e is produced during this one)
    uint32_t it = blockIdx.x, par = 0;
    if (it < a.n_tiles) {
        const uint32_t t0 = w16_tile(a, it);
        load_x(t0, xn);
        load_tile_rows(dh2_srd(t0), 1, w, lane16, bX);
        load_small(t0, bS);
        load_tile_rows(act_srd(a.saved, t0), 2, w, lane16, bH3);
        load_tile_rows(e_srd(t0), 1, 2 + (w & 1), lane16, bE);
        put_e(lx.xe, xn);
        load_x(w16_tile(a, it + gridDim.x < a.n_tiles ? it + gridDim.x : it), xn);
    }
    w16x_barrier();
#pragma clang loop unroll(disable)
    for (; it < a.n_tiles; it += gridDim.x, par ^= (L::NBUF == 2 ? 1u : 0u)) {
        const bool more = it + gridDim.x < a.n_tiles;
        const uint32_t nt = more ? w16_tile(a, it + gridDim.x) : w16_tile(a, it);
        v8* xe = lx.xe + par * L::XE;
        v8* xa = lx.xa + par * L::XA;
        const v8* xb = lx.xb + par * L::XB;
        v8 X[P][2];
        W16_TRACE_DECL;
        W16_MARK(0);
        // ---- H1 column tile w = e W1[w]^T (the operands swapped: lane = feature, registers = 16 samples; the forward's H1
        //      bit for bit: the forward's products in the forward's order), ReLU, planes -> XA
        {
            f32x16 hacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#ifdef W16_DBG_NO_COMPUTE
                break;
#endif
                v8 e[P], wp[P];
#pragma unroll
                for (int p = 0; p < P; ++p) e[p] = xe[(t * P + p) * 64 + lane], wp[p] = w1[p * W1_PLANE + (w * T16H_F1 + t) * 64 + lane];
                hacc = mfma16(e[0], wp[0], hacc);
                hacc = mfma16(e[1], wp[0], hacc);
                if constexpr (P == 3) {
                    hacc = mfma16(e[2], wp[0], hacc);
                    hacc = mfma16(e[1], wp[1], hacc);
                    hacc = mfma16(e[0], wp[1], hacc);
                    hacc = mfma16(e[0], wp[2], hacc);
                } else {
                    hacc = mfma16(e[0], wp[1], hacc);
                }
            }
            v8 Y[P][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x8 r;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    r[u] = __builtin_amdgcn_fmed3f(hacc[8 * m + u] * w16_acc_unscale<A>(), 0.0f, __builtin_inff());
#pragma unroll
                for (int p = 0; p < P; ++p) Y[p][m] = p == P - 1 ? next_plane<A, true>(r) : next_plane<A, false>(r);
            }
            w16x_put<A>(xa + w * L::CT, lane, Y);
        }
        W16_FENCE();
        W16_MARK(1);
        // ---- e k-step w of the NEXT tile.  Two buffers: into the other one, now (it is read by everybody before the next
        //      barrier).  One buffer: behind this tile's first barrier (below).
        if (!TWO_BARRIERS) {
            if (more) put_e(lx.xe + (par ^ 1u) * L::XE, xn);
            load_x(w16_tile(a, it + 2 * gridDim.x < a.n_tiles ? it + 2 * gridDim.x : it), xn);
        }
        W16_FENCE();
        W16_MARK(2);
        // ---- X = dH2[w]
        if (recompute_x) {          // = (Wrgb[:, :64]^T drgb)[row tile w - 2], as the chain kernel computed it
            float up, down;
            w16x_updown(bS[0], up, down);
            v8 rp[P];
            w16x_small_operand<A>(bS[0], N_CLASS, 3, up, h, rp);
            const f32x16 ac = w16x_narrow<A>(gimg, L::G_RGBT + (w - 2) * 64 + lane, L::G_PLANE_RGBT, rp);
            const float unscale = w16_acc_unscale<A>();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) bX[q][u] = (ac[8 * q + u] * unscale) * down;
        }
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bX, 2), k_main, rs);
            if (rs != 1.0f) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] *= rs;
                bsum *= rs;
            }
            bX[0] *= sx, bX[1] *= sx;
        }
        transpose_block<A, true>(bX, I, X, bsum);
        W16_FENCE();
        load_tile_rows(dh2_srd(nt), 1, w, lane16, bX);                  // the next tile's dH2
        W16_FENCE();
        W16_MARK(3);
        w16x_barrier();
        W16_MARK(4);
        if (TWO_BARRIERS) {         // everybody has read this tile's e: the next tile's goes into the same buffer
            if (more) put_e(lx.xe, xn);
            load_x(w16_tile(a, it + 2 * gridDim.x < a.n_tiles ? it + 2 * gridDim.x : it), xn);
        }
        // ---- the four H1 column tiles
        if constexpr (P == 3) {
            w16x_mac2<A>(X, xa, acc[0], X, xa + L::CT, acc[1], lane);
            w16x_mac2<A>(X, xa + 2 * L::CT, acc[2], X, xa + 3 * L::CT, acc[3], lane);
        } else {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) w16x_mac<A>(X, xa + ct * L::CT, lane, acc[ct]);
        }
        W16_FENCE();
        W16_MARK(5);
        // ---- small rows: (d logits, d rgb)^T H3[w] (rows 0..15), ^T {rgb_emb | e} (rows 16..31)
        f32x8 sv1[2], sv2[2];
        if (h != 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bS[0][u] = 0.0f;
        }
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bS, 1), k_small, rs);
            if (rs != 1.0f) acc[4] *= rs, bsmall *= rs;
            bS[0] *= sx;
        }
        sv1[0] = bS[0], sv2[0] = bS[0], sv1[1] = bS[0], sv2[1] = bS[0];
        transpose_block<A, true, 1>(sv1, I, X, bsmall);                             // X = small rows (columns 0..11)
        W16_FENCE();
        transpose_mac<A>(bH3, I, X, acc[4]);                                        // H3[w] -> rows 0..15
        transpose_block<A, false, 1>(sv2, I, X, dummy, 1);                          // the small rows at columns 16..27
        W16_FENCE();
        if (w < 2) transpose_mac<A>(bE, I, X, acc[4]);                              // rgb_emb (waves 0, 1)
        else w16x_mac<A>(X, xb + (3 + (w - 2)) * L::CT, lane, acc[4]);              // e column tile w - 2 (waves 2, 3)
        W16_FENCE();
        load_small(nt, bS);
        load_tile_rows(act_srd(a.saved, nt), 2, w, lane16, bH3);
        load_tile_rows(e_srd(nt), 1, 2 + (w & 1), lane16, bE);
        W16_FENCE();
        W16_MARK(6);
        if (TWO_BARRIERS) w16x_barrier();      // this tile's XA / XB have been read: the next tile may overwrite them
        W16_MARK(7);
        W16_TRACE_SUM(7, w);
    }
    float* rec = a.rec;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
        flush_mapped(rec, G_W_PTS2, HID, lane, acc[ct], w16_unscale(k_main), [&](int i) { return 32 * w + i; }, [&](int c) { return 32 * ct + c; });
    flush_mapped(rec, G_W_SDF2, HID, lane, acc[4], w16_unscale(k_small),
                 [&](int i) { const int r = w16_small_row(i); return r < N_CLASS ? r : -1; }, [&](int c) { return 32 * w + c; });
    auto rgb_row = [&](int i) { const int r = i >= 16 ? w16_small_row(i - 16) : -1; return r >= N_CLASS ? r - N_CLASS : -1; };
    if (w < 2)
        flush_mapped(rec, G_W_RGB0, N_RGB_IN, lane, acc[4], w16_unscale(k_small), rgb_row, [&](int c) { return 32 * w + c; });
    else
        flush_mapped(rec, G_W_RGB0, N_RGB_IN, lane, acc[4], w16_unscale(k_small), rgb_row,
                     [&](int c) { const int e = w16_e_col(w - 2, c); return e >= 0 ? N_EMB + e : -1; });
    const float b2 = (bsum + __shfl_xor(bsum, 32, 64)) * w16_unscale(k_main);
    if (h == 0) rec[G_B_PTS2 + 32 * w + j] = b2;
    if (w == 0) {
        const float bs = (bsmall + __shfl_xor(bsmall, 32, 64)) * w16_unscale(k_small);
        const int r = w16_small_row(j);
        if (h == 0 && r >= 0 && r < N_CLASS) rec[G_B_SDF2 + r] = bs;
        if (h == 0 && r >= N_CLASS) rec[G_B_RGB0 + r - N_CLASS] = bs;
    }
}

// waves 4..7 (rt): d w_sdf0[rt][0..2] = dG3[rt]^T [sdf_emb | grid], d b_sdf0;  d w_pts0[rt][0..1] = dG1[rt]^T e, d b_pts0.
// Produces the transposed column tiles: rt 0, 1 -> sdf_emb 0, 1; rt 2 -> grid; rt 3 -> e 0 and e 1 (from XE's planes).
template <int LAYOUT, typename A>
__device__ __forceinline__ void w16x_role_b(const W16Args& a, const W16X<A>& lx, const typename A::v8 (&I)[2], int rt, int lane) {
    typedef W16XL<A> L;
    typedef typename A::v8 v8;
    constexpr int P = A::P;
    constexpr bool TWO_BARRIERS = L::NBUF == 1;
#ifndef W16_B_EARLY_X
#define W16_B_EARLY_X 1      // experiments: 0 = both transposes between the second barrier and the first one (round 4's first form)
#endif
    constexpr bool EARLY_X = TWO_BARRIERS && W16_B_EARLY_X;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) zero_tile(acc[t]);
    float bsum0 = 0.f, bsum1 = 0.f, dummy = 0.f;
    int k3 = 0, k1 = 0;
    f32x8 bG3[2], bG1[2], bY[2];
    auto act_srd = [&](const float* recs, uint32_t tile) {
        return make_srd(recs + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);
    };
    const uint64_t feat_bytes = (uint64_t)a.M * N_GRID * 4;
    const srd_t feat_srd = make_srd(a.feat, feat_bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)feat_bytes);
    // what this wave transposes for everybody: sdf_emb row tile rt for rt < 2 (4 loads of 16 bytes), the grid features for
    // rt == 2 (16 loads of 4 bytes), both into bY -- the branch is wave-uniform, and the wave with the e column tiles (rt == 3)
    // loads nothing here
    auto load_mine = [&](uint32_t tile) {
        if (rt < 2) {
            load_tile_rows(make_srd(a.saved + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4), 1, rt & 1, lane16, bY);
        } else if (rt == 2) {
            const uint32_t s_raw = tile * 32u + (uint32_t)j;
            const uint32_t s_c = s_raw < a.M ? s_raw : a.M - 1;
            const uint32_t voff = LAYOUT == MIPSF_FEAT_AOS ? s_c * (uint32_t)(N_GRID * 4) + 4u * (uint32_t)h : (s_c * 2u + (uint32_t)h) * 4u;
            const uint32_t lstride = LAYOUT == MIPSF_FEAT_AOS ? 8u : a.M * 8u;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    bY[q][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(feat_srd, voff, (uint32_t)(8 * q + u) * lstride, 0));
        }
    };
    // lean gradient record: dG3[rt] = relu'(H3[rt]) (Ws2^T dlogits)[rt] is recomputed from the sample's 8 small-row values and
    // its mask bits (loaded one tile ahead like everything else); the record's dG3 pieces are then read through an empty resource
    const bool lean = a.gimg != nullptr;
    const v8* gimg = reinterpret_cast<const v8*>(a.gimg);
    const srd_t small_srd = make_srd(a.dsmall, a.M * 32u);
    f32x8 bSm;
    uint2 bMk = make_uint2(0u, 0u);
    auto load_lean = [&](uint32_t tile) {
        const uint32_t off = (tile * 32u + (uint32_t)j) * 32u;
        const float4 p = buf_load16(small_srd, off, 0), q = buf_load16(small_srd, off, 16);
        bSm[0] = p.x, bSm[1] = p.y, bSm[2] = p.z, bSm[3] = p.w, bSm[4] = q.x, bSm[5] = q.y, bSm[6] = q.z, bSm[7] = q.w;
        if (lean) bMk = a.masks[(size_t)tile * (MASK_TILE_WORDS / 2) + 64 + lane];
    };
    auto g3_srd = [&](uint32_t tile) {
        return make_srd(a.dact + (size_t)tile * ACT_TILE_FLOATS, lean ? 0 : ACT_TILE_FLOATS * 4);
    };
    uint32_t it = blockIdx.x, par = 0;
    if (it < a.n_tiles) {
        const uint32_t t0 = w16_tile(a, it);
        load_mine(t0);
        load_lean(t0);
        load_tile_rows(g3_srd(t0), 2, rt, lane16, bG3);
        load_tile_rows(act_srd(a.dact, t0), 0, rt, lane16, bG1);
    }
    w16x_barrier();
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    v8 X3[P][2], X1[P][2];
    // X3 = dG3[rt], X1 = dG1[rt] of the tile whose records are in bSm / bMk / bG3 / bG1, then the loads of tile `nt` into them
    auto make_x = [&](uint32_t nt, bool with_loads) {
        if (lean) {
            float up, down;
            w16x_updown(bSm, up, down);
            v8 lp[P];
            w16x_small_operand<A>(bSm, 0, N_CLASS, up, h, lp);
            const f32x16 ac = w16x_narrow<A>(gimg, rt * 64 + lane, L::G_PLANE_S2T, lp);
            const uint32_t m3[2] = {bMk.x, bMk.y};
            const float unscale = w16_acc_unscale<A>();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) bG3[q][u] = mask_apply(m3, rt, 8 * q + u, ac[8 * q + u] * unscale) * down;
        }
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bG3, 2), k3, rs);
            if (rs != 1.0f) {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] *= rs;
                bsum0 *= rs;
            }
            bG3[0] *= sx, bG3[1] *= sx;
        }
        transpose_block<A, true>(bG3, I, X3, bsum0);
        W16_FENCE();
        if (with_loads) {
            load_lean(nt);
            load_tile_rows(g3_srd(nt), 2, rt, lane16, bG3);
        }
        W16_FENCE();
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bG1, 2), k1, rs);
            if (rs != 1.0f) acc[3] *= rs, acc[4] *= rs, bsum1 *= rs;
            bG1[0] *= sx, bG1[1] *= sx;
        }
        transpose_block<A, true>(bG1, I, X1, bsum1);
        W16_FENCE();
        if (with_loads) load_tile_rows(act_srd(a.dact, nt), 0, rt, lane16, bG1);
        W16_FENCE();
    };
    auto load_x = [&](uint32_t nt) {       // the records make_x reads, of tile nt
        load_lean(nt);
        load_tile_rows(g3_srd(nt), 2, rt, lane16, bG3);
        load_tile_rows(act_srd(a.dact, nt), 0, rt, lane16, bG1);
    };
    // With ONE set of exchange buffers (P = 3: two sets do not fit the CU's LDS) a tile has two barriers, and everything a wave
    // does between the second one and the first one of the next tile is on the workgroup's critical path (the role-a waves wait
    // at the first barrier with their products still to do).  The two transposes of this role are private to the wave, so there
    // they are done for the NEXT tile right behind this tile's products -- beside the role-a waves' products -- and only the
    // column tile everybody waits for is left between the barriers (phase trace, tools/replay.py w16trace: the role-a waves
    // waited 6 350 of their 15 960 cycles per tile at the first barrier).
    if (EARLY_X && it < a.n_tiles) make_x(0u, false);
#pragma clang loop unroll(disable)
    for (; it < a.n_tiles; it += gridDim.x, par ^= (L::NBUF == 2 ? 1u : 0u)) {
        const uint32_t nt = it + gridDim.x < a.n_tiles ? w16_tile(a, it + gridDim.x) : w16_tile(a, it);
        const v8* xe = lx.xe + par * L::XE;
        v8* xb = lx.xb + par * L::XB;
        W16_TRACE_DECL;
        W16_MARK(0);
        // ---- this wave's column tile(s) -> XB
        if (rt < 3) {
            if (rt == 2) bY[0] = bY[0] * w16_grid_shift<A>(), bY[1] = bY[1] * w16_grid_shift<A>();
            v8 Y[P][2];
            transpose_block<A, false>(bY, I, Y, dummy);
            w16x_put<A>(xb + rt * L::CT, lane, Y);
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {           // e column tile blk = k-steps 2 blk, 2 blk + 1, plane by plane
                v8 Y[P][2];
#pragma unroll
                for (int pb = 0; pb < P; ++pb) {
#ifdef W16_DBG_NO_COMPUTE
                    Y[pb][0] = I[0], Y[pb][1] = I[1];
                    continue;
#endif
                    f32x16 T = mfma16(xe[((2 * blk) * P + pb) * 64 + lane], I[0], zero);
                    T = mfma16(xe[((2 * blk + 1) * P + pb) * 64 + lane], I[1], T);
                    pack_T<A>(T, Y[pb]);
                }
                w16x_put<A>(xb + (3 + blk) * L::CT, lane, Y);
            }
        }
        W16_FENCE();
        load_mine(nt);
        W16_FENCE();
        W16_MARK(1);
#ifndef W16_B_LOADS_LATE
#define W16_B_LOADS_LATE 1           // experiments: 0 = the next records' loads in front of the first barrier
#endif
        if (!EARLY_X) make_x(nt, true);       // ---- X3 = dG3[rt], X1 = dG1[rt]
        else if (!W16_B_LOADS_LATE) load_x(nt);
        W16_MARK(2);
        W16_MARK(3);
        w16x_barrier();
        W16_MARK(4);
        // (the column tile's temporaries are gone, and issuing these loads is not in the other waves' way any more: the
        // records of the next tile's X3, X1; the products below cover their latency)
        if (EARLY_X && W16_B_LOADS_LATE) load_x(nt);
        if constexpr (P == 3) {
            w16x_mac2<A>(X3, xb, acc[0], X3, xb + L::CT, acc[1], lane);
            w16x_mac2<A>(X3, xb + 2 * L::CT, acc[2], X1, xb + 3 * L::CT, acc[3], lane);
            w16x_mac<A>(X1, xb + 4 * L::CT, lane, acc[4]);
        } else {
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) w16x_mac<A>(X3, xb + ct * L::CT, lane, acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) w16x_mac<A>(X1, xb + (3 + ct) * L::CT, lane, acc[3 + ct]);
        }
        W16_FENCE();
        W16_MARK(5);
        if (EARLY_X && it + gridDim.x < a.n_tiles) make_x(0u, false);      // the next tile's X3, X1
        W16_MARK(6);
        if (TWO_BARRIERS) w16x_barrier();
        W16_MARK(7);
        W16_TRACE_SUM(7, 4 + rt);
    }
    float* rec = a.rec;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        flush_mapped(rec, G_W_SDF0, N_SDF_IN, lane, acc[ct], w16_unscale(k3), [&](int i) { return 32 * rt + i; }, [&](int c) { return 32 * ct + c; });
    flush_mapped(rec, G_W_SDF0, N_SDF_IN, lane, acc[2], w16_unscale(k3) / w16_grid_shift<A>(), [&](int i) { return 32 * rt + i; },
                 [&](int c) { return N_EMB + 2 * (8 * (c >> 4) + 4 * ((c >> 3) & 1) + (c & 3)) + ((c >> 2) & 1); });
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        flush_mapped(rec, G_W_PTS0, N_E, lane, acc[3 + ct], w16_unscale(k1), [&](int i) { return 32 * rt + i; },
                     [&](int c) { return w16_e_col(ct, c); });
    const float b3 = (bsum0 + __shfl_xor(bsum0, 32, 64)) * w16_unscale(k3), b1 = (bsum1 + __shfl_xor(bsum1, 32, 64)) * w16_unscale(k1);
    if (h == 0) rec[G_B_SDF0 + 32 * rt + j] = b3, rec[G_B_PTS0 + 32 * rt + j] = b1;
}


// ============================================================ the TRANSPOSE-READ form (round 6): the exchange form without
// matrix-core transposes.  What round 2 gave to the matrix pipe -- T = X I, 2 MFMAs + 16 converts back to 16 bits per plane of
// a 32 x 32 block: 162 of the exchange form's 830 MFMAs per tile -- is what gfx950's LDS does on the way out:
// ds_read_b64_tr_b16 hands lane i of a 16-lane group element (i & 3) of the four 8-byte chunks that lanes 4 k + (i >> 2) of
// the group address, k = 0..3 (tools/micro/tr_probe.hip checks the mapping on the device).  A block is cut into its 16-bit
// planes in the records' LOAD layout (lane = sample, 16 features per lane as four chunks of four consecutive features), every
// plane is written to LDS as it is cut (2 x ds_write_b128 per lane) and read back as the MFMA operand: lane = feature, 8
// consecutive samples per lane = two transpose reads of 4 samples each.
//   plane of a block (2 KB): row = sample (64 B), four 16-byte slots; the lane (j, h) of the load layout owns slots
//       (2 q + h) ^ sw(j), q = 0, 1 (features 16 q + 4 h + {0..3} and 16 q + 8 + 4 h + {0..3}: the two chunks of a slot),
//       sw(j) = bit 1 of j | (bit 2 ^ bit 3 of j) << 1:  the 8 lanes of a ds_write_b128 group hit 8 different bank quads, the
//       32 lanes of a transpose read cover 4 whole rows (32-wide operand) or complementary halves of 8 rows (16-wide operand)
//       = all 64 banks once.
// The small-row products (d w_sdf2: 5 rows, d w_rgb0: 3 rows; 96 MFMAs of 32 x 32 x 16 + 60 transposing ones per tile in the
// exchange form, for 0.7 % of the arithmetic) run on v_mfma_f32_16x16x32: its k = 32 is the whole tile's samples, 16 output
// rows hold the 8 small rows, and a product is one instruction of half the cycles per 16 columns.
// Bias gradients: d b_pts0 and the small rows' are columns of products that exist anyway (the e operand carries constant
// ones against the layer-1 bias pieces: column 18 of its second column tile); d b_pts2 / d b_sdf0 are per-lane sums in the
// load layout, reduced across lanes once per launch.
// Per tile: 4 x (24 + 48 + 24 half-size [+ 6]) + 4 x (60 [+ 6] [+ 12]) = 684 MFMA instructions (830), 612 in 32 x 32 units.
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define W16_LDS(T, addr) ((__attribute__((address_space(3))) T*)(addr))

__device__ __forceinline__ f32x4 mfma16s(bf8 a, bf8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16s(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ uint32_t w16t_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// two transpose reads = one 8-element operand (whole-vector bit casts only: an element-wise __builtin_bit_cast of the
// result's elements compiled to element 0 four times, tools/micro/tr_probe.hip)
template <typename A>
__device__ __forceinline__ typename A::v8 w16t_tr2(uint32_t a0, uint32_t a1) {
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(W16_LDS(s4v, a0));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(W16_LDS(s4v, a1));
    return __builtin_bit_cast(typename A::v8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <typename A>
__device__ __forceinline__ void w16t_st(uint32_t addr, const typename A::v8& v) { *W16_LDS(typename A::v8, addr) = v; }
template <typename A>
__device__ __forceinline__ typename A::v8 w16t_ld(uint32_t addr) { return *W16_LDS(typename A::v8, addr); }

constexpr uint32_t W16T_PLANE = 2048;       // bytes of one 16-bit plane of a 32 x 32 block

// this lane's byte offsets inside a plane
struct W16TAddr {
    uint32_t wr0, wr1;      // its two slots (q = 0, 1)
    uint32_t rd0, rd1;      // 32-wide operand (v_mfma_32x32x16: lane = feature l & 31, half l >> 5): the two transpose reads of
                            //   k-step 0; k-step 1 at + 1024
    uint32_t rsa, rsb;      // 16-wide operand (v_mfma_16x16x32: lane = feature 16 cg + (l & 15), k group l >> 4): read r of
                            //   column group cg at (cg ^ r ? rsb : rsa) + 512 r
    uint32_t ry;            // 16-wide operand out of READY 32-wide operand planes ([k-step m][64 lanes] x 16 B): column group cg of
                            //   plane entry e0 (its k-step 0) at (e0 * 64 + 16 cg) * 16 + ry
};
__device__ __forceinline__ W16TAddr w16t_addr(int lane) {
    W16TAddr ad;
    const uint32_t j = lane & 31, h = lane >> 5;
    const uint32_t sw = ((j >> 1) & 1u) | ((((j >> 2) ^ (j >> 3)) & 1u) << 1);
    ad.wr0 = j * 64u + ((h ^ sw) * 16u);
    ad.wr1 = j * 64u + (((2u + h) ^ sw) * 16u);
    const uint32_t g = lane >> 4, ii = lane & 15, hh = ii & 1u, aa = (ii >> 1) & 1u, s1 = (ii >> 3) & 1u;
    // Which sample is element u of an operand: the READY planes (H1 out of its product, the e column tiles out of pack_T) hold the
    // accumulator rows of a 32 x 32 tile, k-step m, half kg, element u = sample 16 m + 8 (u >> 2) + 4 kg + (u & 3); the transpose
    // reads follow that order (read r = u >> 2 fetches rows 8 r + 4 kg + {0..3}), a product pairs the same samples on both sides.
    {   // 32-wide: q = g & 1, kg = g >> 1; row 16 m + 8 r + 4 kg + (ii >> 2), slot 2 (q ^ r ^ kg) + (hh ^ s1)
        const uint32_t q = g & 1u, kg = g >> 1;
        const uint32_t row0 = 4u * kg + (ii >> 2), lo = (hh ^ s1) * 16u + 8u * aa;
        ad.rd0 = row0 * 64u + 32u * (q ^ kg) + lo;
        ad.rd1 = (row0 + 8u) * 64u + 32u * (q ^ 1u ^ kg) + lo;
    }
    {   // 16-wide: group g = k-step g >> 1, half g & 1 of the same order: row 16 (g >> 1) + 8 r + 4 (g & 1) + (ii >> 2),
        // slot 2 (cg ^ r ^ (g & 1)) + (hh ^ s1)
        const uint32_t base = (16u * (g >> 1) + 4u * (g & 1u) + (ii >> 2)) * 64u + (hh ^ s1) * 16u + 8u * aa;
        ad.rsa = base + 32u * (g & 1u);
        ad.rsb = base + 32u * (1u - (g & 1u));
    }
    ad.ry = (((g >> 1) * 64u) + (g & 1u) * 32u + ii) * 16u;
    return ad;
}

// plane by plane: cut, write, read back transposed -- X[p][m] = the 32-wide operand (A or B alike) of k-step m.  One plane of
// scratch per wave: the LDS executes a wave's instructions in order, the next plane's writes cannot pass this plane's reads.
template <typename A>
__device__ __forceinline__ void w16t_x32(f32x8 (&vals)[2], uint32_t scr, const W16TAddr& ad, typename A::v8 (&X)[A::P][2]) {
    constexpr int P = A::P;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        w16t_st<A>(scr + ad.wr0, W16_PLANE(A, p == P - 1, vals[0], p));
        w16t_st<A>(scr + ad.wr1, W16_PLANE(A, p == P - 1, vals[1], p));
        X[p][0] = w16t_tr2<A>(scr + ad.rd0, scr + ad.rd1);
        X[p][1] = w16t_tr2<A>(scr + ad.rd0 + 1024u, scr + ad.rd1 + 1024u);
    }
}
// a block as a column tile for everybody: its planes in load layout at `tile` (P planes)
template <typename A>
__device__ __forceinline__ void w16t_put_block(f32x8 (&vals)[2], uint32_t tile, const W16TAddr& ad) {
    constexpr int P = A::P;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        w16t_st<A>(tile + p * W16T_PLANE + ad.wr0, W16_PLANE(A, p == P - 1, vals[0], p));
        w16t_st<A>(tile + p * W16T_PLANE + ad.wr1, W16_PLANE(A, p == P - 1, vals[1], p));
    }
}
// where a product's right-hand operand (plane pb, k-step m) comes from: ready planes, or load-layout planes read transposed
template <typename A>
struct W16YReady {
    uint32_t base;      // tile + 16 lane
    __device__ __forceinline__ typename A::v8 operator()(int pb, int m) const { return w16t_ld<A>(base + (uint32_t)(2 * pb + m) * 1024u); }
};
template <typename A>
struct W16YTr {
    uint32_t b0, b1;    // tile + rd0, tile + rd1
    __device__ __forceinline__ typename A::v8 operator()(int pb, int m) const {
        const uint32_t o = (uint32_t)pb * W16T_PLANE + (uint32_t)m * 1024u;
        return w16t_tr2<A>(b0 + o, b1 + o);
    }
};
// acc_a += Xa^T Ya, acc_b += Xb^T Yb, the two chains' MFMAs in turn (w16x_mac2)
template <typename A, typename YA, typename YB>
__device__ __forceinline__ void w16t_mac2(const typename A::v8 (&Xa)[A::P][2], const YA& ya, f32x16& acc_a,
                                          const typename A::v8 (&Xb)[A::P][2], const YB& yb, f32x16& acc_b) {
    constexpr int P = A::P;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pb = 0; pb < P; ++pb) {
            const typename A::v8 Ya = ya(pb, m), Yb = yb(pb, m);
#pragma unroll
            for (int pa = 0; pa + pb < P; ++pa) {
                acc_a = mfma16(Xa[pa][m], Ya, acc_a);
                acc_b = mfma16(Xb[pa][m], Yb, acc_b);
            }
        }
}
template <typename A, typename YA>
__device__ __forceinline__ void w16t_mac(const typename A::v8 (&Xa)[A::P][2], const YA& ya, f32x16& acc_a) {
    constexpr int P = A::P;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pb = 0; pb < P; ++pb) {
            const typename A::v8 Ya = ya(pb, m);
#pragma unroll
            for (int pa = 0; pa + pb < P; ++pa) acc_a = mfma16(Xa[pa][m], Ya, acc_a);
        }
}
// the small rows times a private block (its planes through the wave's scratch): acc[cg] += S^T Y[:, 16 cg ..]
template <typename A>
__device__ __forceinline__ void w16t_small_block(f32x8 (&vals)[2], uint32_t scr, const W16TAddr& ad, const typename A::v8 (&S)[A::P],
                                                 f32x4 (&acc)[2]) {
    constexpr int P = A::P;
#pragma unroll
    for (int pb = 0; pb < P; ++pb) {
        w16t_st<A>(scr + ad.wr0, W16_PLANE(A, pb == P - 1, vals[0], pb));
        w16t_st<A>(scr + ad.wr1, W16_PLANE(A, pb == P - 1, vals[1], pb));
        const typename A::v8 Y0 = w16t_tr2<A>(scr + ad.rsa, scr + ad.rsb + 512u);       // cg 0: r = 0 -> rsa, r = 1 -> rsb
        const typename A::v8 Y1 = w16t_tr2<A>(scr + ad.rsb, scr + ad.rsa + 512u);       // cg 1
#pragma unroll
        for (int pa = 0; pa + pb < P; ++pa) {
            acc[0] = mfma16s(S[pa], Y0, acc[0]);
            acc[1] = mfma16s(S[pa], Y1, acc[1]);
        }
    }
}
// the same against a column tile of READY 32-wide operand planes (the e column tiles)
template <typename A>
__device__ __forceinline__ void w16t_small_ready(uint32_t tile, const W16TAddr& ad, const typename A::v8 (&S)[A::P], f32x4 (&acc)[2]) {
    constexpr int P = A::P;
#pragma unroll
    for (int pb = 0; pb < P; ++pb) {
        const typename A::v8 Y0 = w16t_ld<A>(tile + (uint32_t)(2 * pb) * 1024u + ad.ry);
        const typename A::v8 Y1 = w16t_ld<A>(tile + (uint32_t)(2 * pb) * 1024u + 256u + ad.ry);
#pragma unroll
        for (int pa = 0; pa + pb < P; ++pa) {
            acc[0] = mfma16s(S[pa], Y0, acc[0]);
            acc[1] = mfma16s(S[pa], Y1, acc[1]);
        }
    }
}
// sum over the 32 lanes of a half (the samples of a tile): once per launch
__device__ __forceinline__ float w16t_half_sum(float v) {
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d, 64);
    return v;
}
constexpr int W16T_ONES_COL = 16 + BIAS16_U;      // column of the e column tile 1 that carries the constant one of half 0

template <typename A>
struct W16T {
    uint32_t xe;        // LDS byte addresses: e as the forward's layer-1 operand [NBUF][4 k-steps][P][64 lanes] x 16 B
    uint32_t xa;        // H1 column tiles, ready planes [NBUF][4][2 P][64] x 16 B
    uint32_t xb;        // [NBUF][5 column tiles]: sdf_emb 0, sdf_emb 1, grid in load layout (P planes of 2 KB), e 0, e 1 ready
    uint32_t scr;       // this wave's plane of scratch
};

// waves 0..3 (w): d w_pts2[w][0..3] = dH2[w]^T H1, d b_pts2; d w_sdf2[:, 32 w ..] = (d logits)^T H3[w]; d w_rgb0 columns
// {rgb_emb 0 | rgb_emb 1 | e 0 | e 1}[w] = (d rgb)^T ...; wave 3: the small rows' bias gradients.  Produces e k-step w and
// H1 column tile w.
template <int LAYOUT, typename A>
__device__ __forceinline__ void w16t_role_a(const W16Args& a, const W16T<A>& lx, int w, int lane) {
    typedef W16XL<A> L;
    typedef typename A::v8 v8;
    constexpr int P = A::P;
    constexpr bool TWO_BARRIERS = L::NBUF == 1;
    constexpr uint32_t XE_B = L::XE * 16u, XA_B = L::XA * 16u, XB_B = L::XB * 16u, CT_B = L::CT * 16u;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const W16TAddr ad = w16t_addr(lane);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zero_tile(acc[t]);
    f32x4 acc_s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acc_r[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x8 bacc[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) bacc[0][u] = 0.f, bacc[1][u] = 0.f;
    int k_main = 0, k_small = 0;
    auto act_srd = [&](const float* recs, uint32_t tile) {
        return make_srd(recs + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);
    };
    const srd_t small_srd = make_srd(a.dsmall, a.M * 32u), x_srd = make_srd(a.x, a.M * 12u);
    auto load_small = [&](uint32_t tile, f32x8 (&v)[2]) {
        const uint32_t off = (tile * 32u + (uint32_t)j) * 32u;
        const float4 p = buf_load16(small_srd, off, 0), q = buf_load16(small_srd, off, 16);
        v[0][0] = p.x, v[0][1] = p.y, v[0][2] = p.z, v[0][3] = p.w, v[0][4] = q.x, v[0][5] = q.y, v[0][6] = q.z, v[0][7] = q.w;
    };
    const v8* gimg = reinterpret_cast<const v8*>(a.gimg);
    const v8* w1 = reinterpret_cast<const v8*>(a.w1_hi);
    constexpr int W1_PLANE = RT_F1 * T16H_F1 * 64;
    const bool recompute_x = a.gimg != nullptr && w >= 2;
    auto dh2_srd = [&](uint32_t tile) {
        return make_srd(a.dact + (size_t)tile * ACT_TILE_FLOATS, recompute_x ? 0 : ACT_TILE_FLOATS * 4);
    };
    auto load_x = [&](uint32_t tile, float (&v)[3]) {
        const uint32_t s_raw = tile * 32u + (uint32_t)j;
        const uint32_t off = (s_raw < a.M ? s_raw : a.M - 1) * 12u;
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_srd, off, 4 * d, 0));
    };
    auto e_srd = [&](uint32_t tile) {       // (every wave issues the same loads: see w16x_role_a)
        return make_srd(a.saved + (size_t)tile * ACT_TILE_FLOATS, w < 2 ? ACT_TILE_FLOATS * 4 : 0);
    };
    auto put_e = [&](uint32_t xe, const float (&xv)[3]) {
        v8 e[P];
        w16x_e_step<A>(w, xv[0], xv[1], xv[2], h, e);
#pragma unroll
        for (int p = 0; p < P; ++p) w16t_st<A>(xe + (uint32_t)((w * P + p) * 64) * 16u + lane16, e[p]);
    };
    f32x8 bX[2], bS[2], bH3[2], bE[2];
    float xn[3] = {0.f, 0.f, 0.f};
    uint32_t it = blockIdx.x, par = 0;
    if (it < a.n_tiles) {
        const uint32_t t0 = w16_tile(a, it);
        load_x(t0, xn);
        load_tile_rows(dh2_srd(t0), 1, w, lane16, bX);
        load_small(t0, bS);
        load_tile_rows(act_srd(a.saved, t0), 2, w, lane16, bH3);
        load_tile_rows(e_srd(t0), 1, 2 + (w & 1), lane16, bE);
        put_e(lx.xe, xn);
        load_x(w16_tile(a, it + gridDim.x < a.n_tiles ? it + gridDim.x : it), xn);
    }
    w16x_barrier();
#pragma clang loop unroll(disable)
    for (; it < a.n_tiles; it += gridDim.x, par ^= (L::NBUF == 2 ? 1u : 0u)) {
        const bool more = it + gridDim.x < a.n_tiles;
        const uint32_t nt = more ? w16_tile(a, it + gridDim.x) : w16_tile(a, it);
        const uint32_t xe = lx.xe + par * XE_B, xa = lx.xa + par * XA_B, xb = lx.xb + par * XB_B;
        v8 X[P][2];
        W16_TRACE_DECL;
        W16_MARK(0);
        // ---- H1 column tile w = e W1[w]^T, ReLU, planes -> XA (comes out of the product in operand layout: w16x_role_a)
        {
            f32x16 hacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v8 e[P], wp[P];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    e[p] = w16t_ld<A>(xe + (uint32_t)((t * P + p) * 64) * 16u + lane16), wp[p] = w1[p * W1_PLANE + (w * T16H_F1 + t) * 64 + lane];
                hacc = mfma16(e[0], wp[0], hacc);
                hacc = mfma16(e[1], wp[0], hacc);
                if constexpr (P == 3) {
                    hacc = mfma16(e[2], wp[0], hacc);
                    hacc = mfma16(e[1], wp[1], hacc);
                    hacc = mfma16(e[0], wp[1], hacc);
                    hacc = mfma16(e[0], wp[2], hacc);
                } else {
                    hacc = mfma16(e[0], wp[1], hacc);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x8 r;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    r[u] = __builtin_amdgcn_fmed3f(hacc[8 * m + u] * w16_acc_unscale<A>(), 0.0f, __builtin_inff());
#pragma unroll
                for (int p = 0; p < P; ++p)
                    w16t_st<A>(xa + (uint32_t)w * CT_B + (uint32_t)(2 * p + m) * 1024u + lane16,
                               p == P - 1 ? next_plane<A, true>(r) : next_plane<A, false>(r));
            }
        }
        W16_FENCE();
        W16_MARK(1);
        if (!TWO_BARRIERS) {
            if (more) put_e(lx.xe + (par ^ 1u) * XE_B, xn);
            load_x(w16_tile(a, it + 2 * gridDim.x < a.n_tiles ? it + 2 * gridDim.x : it), xn);
        }
        W16_FENCE();
        W16_MARK(2);
        // ---- X = dH2[w]
        if (recompute_x) {
            float up, down;
            w16x_updown(bS[0], up, down);
            v8 rp[P];
            w16x_small_operand<A>(bS[0], N_CLASS, 3, up, h, rp);
            const f32x16 ac = w16x_narrow<A>(gimg, L::G_RGBT + (w - 2) * 64 + lane, L::G_PLANE_RGBT, rp);
            const float unscale = w16_acc_unscale<A>();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) bX[q][u] = (ac[8 * q + u] * unscale) * down;
        }
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bX, 2), k_main, rs);
            if (rs != 1.0f) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] *= rs;
                bacc[0] *= rs, bacc[1] *= rs;
            }
            bX[0] *= sx, bX[1] *= sx;
        }
        bacc[0] = bacc[0] + bX[0], bacc[1] = bacc[1] + bX[1];
        w16t_x32<A>(bX, lx.scr, ad, X);
        W16_FENCE();
        load_tile_rows(dh2_srd(nt), 1, w, lane16, bX);
        W16_FENCE();
        W16_MARK(3);
        w16x_barrier();
        W16_MARK(4);
        if (TWO_BARRIERS) {
            if (more) put_e(lx.xe, xn);
            load_x(w16_tile(a, it + 2 * gridDim.x < a.n_tiles ? it + 2 * gridDim.x : it), xn);
        }
        // ---- the four H1 column tiles (ready planes)
        {
            const W16YReady<A> y0 = {xa + lane16}, y1 = {xa + CT_B + lane16}, y2 = {xa + 2 * CT_B + lane16}, y3 = {xa + 3 * CT_B + lane16};
            w16t_mac2<A>(X, y0, acc[0], X, y1, acc[1]);
            w16t_mac2<A>(X, y2, acc[2], X, y3, acc[3]);
        }
        W16_FENCE();
        W16_MARK(5);
        // ---- small rows: rows {0..3, 8..11} of a 16-row operand = (d logits 0..4, d rgb 0..2)
        if (h != 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bS[0][u] = 0.0f;
        }
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bS, 1), k_small, rs);
            if (rs != 1.0f) acc_s[0] *= rs, acc_s[1] *= rs, acc_r[0] *= rs, acc_r[1] *= rs;
            bS[0] *= sx;
        }
        v8 S[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {        // slot 0 of every row: features {0..3, 8..11} by half 0, zeros by half 1
            w16t_st<A>(lx.scr + ad.wr0, W16_PLANE(A, p == P - 1, bS[0], p));
            S[p] = w16t_tr2<A>(lx.scr + ad.rsa, lx.scr + ad.rsb + 512u);
        }
        W16_FENCE();
        w16t_small_block<A>(bH3, lx.scr, ad, S, acc_s);                                  // H3[w]
        W16_FENCE();
        if (w < 2) w16t_small_block<A>(bE, lx.scr, ad, S, acc_r);                        // rgb_emb (waves 0, 1)
        else w16t_small_ready<A>(xb + (uint32_t)(3 + (w - 2)) * CT_B, ad, S, acc_r);     // e column tile w - 2 (waves 2, 3)
        W16_FENCE();
        load_small(nt, bS);
        load_tile_rows(act_srd(a.saved, nt), 2, w, lane16, bH3);
        load_tile_rows(e_srd(nt), 1, 2 + (w & 1), lane16, bE);
        W16_FENCE();
        W16_MARK(6);
        if (TWO_BARRIERS) w16x_barrier();
        W16_MARK(7);
        W16_TRACE_SUM(7, w);
    }
    float* rec = a.rec;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
        flush_mapped(rec, G_W_PTS2, HID, lane, acc[ct], w16_unscale(k_main), [&](int i) { return 32 * w + i; }, [&](int c) { return 32 * ct + c; });
    // 16 x 16 tiles: lane = column n of column group cg, rows 4 (lane >> 4) + r; the small rows sit in rows {0..3, 8..11}
    {
        const int n = lane & 15, rg = lane >> 4;
        const float us = w16_unscale(k_small);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = w16_small_row(4 * rg + r);         // 0..7 or -1
                if (v >= 0 && v < N_CLASS) rec[G_W_SDF2 + v * HID + 32 * w + 16 * cg + n] = acc_s[cg][r] * us;
                if (v >= N_CLASS) {
                    const int col = w < 2 ? 32 * w + 16 * cg + n : (w16_e_col(w - 2, 16 * cg + n) >= 0 ? N_EMB + w16_e_col(w - 2, 16 * cg + n) : -1);
                    if (col >= 0) rec[G_W_RGB0 + (v - N_CLASS) * N_RGB_IN + col] = acc_r[cg][r] * us;
                }
                // the ones column of e's second column tile: sum over the samples of every small row
                if (w == 3 && 16 * cg + n == W16T_ONES_COL && v >= 0) {
                    if (v < N_CLASS) rec[G_B_SDF2 + v] = acc_r[cg][r] * us;
                    else rec[G_B_RGB0 + v - N_CLASS] = acc_r[cg][r] * us;
                }
            }
    }
    // d b_pts2: this lane's 16 features of row tile w, summed over the 32 samples of its half
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float t = w16t_half_sum(bacc[q][u]) * w16_unscale(k_main);
            if (j == 0) rec[G_B_PTS2 + 32 * w + 16 * q + 8 * (u >> 2) + 4 * h + (u & 3)] = t;
        }
}

// waves 4..7 (rt): d w_sdf0[rt][0..2] = dG3[rt]^T [sdf_emb | grid], d b_sdf0;  d w_pts0[rt][0..1] = dG1[rt]^T e, d b_pts0 (the
// ones column).  Produces column tiles: rt 0, 1 -> sdf_emb 0, 1 and rt 2 -> grid, as planes in load layout; rt 3 -> e 0 and
// e 1 from XE's operand planes (the one transposition left on the matrix pipe: 12 MFMAs per tile).
template <int LAYOUT, typename A>
__device__ __forceinline__ void w16t_role_b(const W16Args& a, const W16T<A>& lx, const typename A::v8 (&I)[2], int rt, int lane) {
    typedef W16XL<A> L;
    typedef typename A::v8 v8;
    constexpr int P = A::P;
    constexpr bool TWO_BARRIERS = L::NBUF == 1;
    constexpr bool EARLY_X = TWO_BARRIERS;
    constexpr uint32_t XE_B = L::XE * 16u, XB_B = L::XB * 16u, CT_B = L::CT * 16u;
    static_assert(L::CT * 16 == P * (int)W16T_PLANE, "a column tile is P planes in either layout");
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const W16TAddr ad = w16t_addr(lane);
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) zero_tile(acc[t]);
    f32x8 bacc[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) bacc[0][u] = 0.f, bacc[1][u] = 0.f;
    int k3 = 0, k1 = 0;
    f32x8 bG3[2], bG1[2], bY[2];
    auto act_srd = [&](const float* recs, uint32_t tile) {
        return make_srd(recs + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);
    };
    const uint64_t feat_bytes = (uint64_t)a.M * N_GRID * 4;
    const srd_t feat_srd = make_srd(a.feat, feat_bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)feat_bytes);
    auto load_mine = [&](uint32_t tile) {
        if (rt < 2) {
            load_tile_rows(make_srd(a.saved + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4), 1, rt & 1, lane16, bY);
        } else if (rt == 2) {
            const uint32_t s_raw = tile * 32u + (uint32_t)j;
            const uint32_t s_c = s_raw < a.M ? s_raw : a.M - 1;
            const uint32_t voff = LAYOUT == MIPSF_FEAT_AOS ? s_c * (uint32_t)(N_GRID * 4) + 4u * (uint32_t)h : (s_c * 2u + (uint32_t)h) * 4u;
            const uint32_t lstride = LAYOUT == MIPSF_FEAT_AOS ? 8u : a.M * 8u;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    bY[q][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(feat_srd, voff, (uint32_t)(8 * q + u) * lstride, 0));
        }
    };
    const bool lean = a.gimg != nullptr;
    const v8* gimg = reinterpret_cast<const v8*>(a.gimg);
    const srd_t small_srd = make_srd(a.dsmall, a.M * 32u);
    f32x8 bSm;
    uint2 bMk = make_uint2(0u, 0u);
    auto load_lean = [&](uint32_t tile) {
        const uint32_t off = (tile * 32u + (uint32_t)j) * 32u;
        const float4 p = buf_load16(small_srd, off, 0), q = buf_load16(small_srd, off, 16);
        bSm[0] = p.x, bSm[1] = p.y, bSm[2] = p.z, bSm[3] = p.w, bSm[4] = q.x, bSm[5] = q.y, bSm[6] = q.z, bSm[7] = q.w;
        if (lean) bMk = a.masks[(size_t)tile * (MASK_TILE_WORDS / 2) + 64 + lane];
    };
    auto g3_srd = [&](uint32_t tile) {
        return make_srd(a.dact + (size_t)tile * ACT_TILE_FLOATS, lean ? 0 : ACT_TILE_FLOATS * 4);
    };
    uint32_t it = blockIdx.x, par = 0;
    if (it < a.n_tiles) {
        const uint32_t t0 = w16_tile(a, it);
        load_mine(t0);
        load_lean(t0);
        load_tile_rows(g3_srd(t0), 2, rt, lane16, bG3);
        load_tile_rows(act_srd(a.dact, t0), 0, rt, lane16, bG1);
    }
    w16x_barrier();
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    v8 X3[P][2], X1[P][2];
    auto make_x = [&](uint32_t nt, bool with_loads) {
        if (lean) {
            float up, down;
            w16x_updown(bSm, up, down);
            v8 lp[P];
            w16x_small_operand<A>(bSm, 0, N_CLASS, up, h, lp);
            const f32x16 ac = w16x_narrow<A>(gimg, rt * 64 + lane, L::G_PLANE_S2T, lp);
            const uint32_t m3[2] = {bMk.x, bMk.y};
            const float unscale = w16_acc_unscale<A>();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) bG3[q][u] = mask_apply(m3, rt, 8 * q + u, ac[8 * q + u] * unscale) * down;
        }
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bG3, 2), k3, rs);
            if (rs != 1.0f) {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] *= rs;
                bacc[0] *= rs, bacc[1] *= rs;
            }
            bG3[0] *= sx, bG3[1] *= sx;
        }
        bacc[0] = bacc[0] + bG3[0], bacc[1] = bacc[1] + bG3[1];
        w16t_x32<A>(bG3, lx.scr, ad, X3);
        W16_FENCE();
        if (with_loads) {
            load_lean(nt);
            load_tile_rows(g3_srd(nt), 2, rt, lane16, bG3);
        }
        W16_FENCE();
        if (A::SCALED) {
            float rs;
            const float sx = w16_pick_scale(w16_block_max_bits(bG1, 2), k1, rs);
            if (rs != 1.0f) acc[3] *= rs, acc[4] *= rs;
            bG1[0] *= sx, bG1[1] *= sx;
        }
        w16t_x32<A>(bG1, lx.scr, ad, X1);
        W16_FENCE();
        if (with_loads) load_tile_rows(act_srd(a.dact, nt), 0, rt, lane16, bG1);
        W16_FENCE();
    };
    auto load_x = [&](uint32_t nt) {
        load_lean(nt);
        load_tile_rows(g3_srd(nt), 2, rt, lane16, bG3);
        load_tile_rows(act_srd(a.dact, nt), 0, rt, lane16, bG1);
    };
    if (EARLY_X && it < a.n_tiles) make_x(0u, false);
#pragma clang loop unroll(disable)
    for (; it < a.n_tiles; it += gridDim.x, par ^= (L::NBUF == 2 ? 1u : 0u)) {
        const uint32_t nt = it + gridDim.x < a.n_tiles ? w16_tile(a, it + gridDim.x) : w16_tile(a, it);
        const uint32_t xe = lx.xe + par * XE_B, xb = lx.xb + par * XB_B;
        W16_TRACE_DECL;
        W16_MARK(0);
        // ---- this wave's column tile(s) -> XB
        if (rt < 3) {
            if (rt == 2) bY[0] = bY[0] * w16_grid_shift<A>(), bY[1] = bY[1] * w16_grid_shift<A>();
            w16t_put_block<A>(bY, xb + (uint32_t)rt * CT_B, ad);
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                v8 Y[P][2];
#pragma unroll
                for (int pb = 0; pb < P; ++pb) {
                    f32x16 T = mfma16(w16t_ld<A>(xe + (uint32_t)(((2 * blk) * P + pb) * 64) * 16u + lane16), I[0], zero);
                    T = mfma16(w16t_ld<A>(xe + (uint32_t)(((2 * blk + 1) * P + pb) * 64) * 16u + lane16), I[1], T);
                    pack_T<A>(T, Y[pb]);
                }
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int m = 0; m < 2; ++m) w16t_st<A>(xb + (uint32_t)(3 + blk) * CT_B + (uint32_t)(2 * p + m) * 1024u + lane16, Y[p][m]);
            }
        }
        W16_FENCE();
        load_mine(nt);
        W16_FENCE();
        W16_MARK(1);
        if (!EARLY_X) make_x(nt, true);
        W16_MARK(2);
        W16_MARK(3);
        w16x_barrier();
        W16_MARK(4);
        if (EARLY_X) load_x(nt);
        {
            const W16YTr<A> y0 = {xb + ad.rd0, xb + ad.rd1}, y1 = {xb + CT_B + ad.rd0, xb + CT_B + ad.rd1},
                            y2 = {xb + 2 * CT_B + ad.rd0, xb + 2 * CT_B + ad.rd1};
            const W16YReady<A> y3 = {xb + 3 * CT_B + lane16}, y4 = {xb + 4 * CT_B + lane16};
            w16t_mac2<A>(X3, y0, acc[0], X3, y1, acc[1]);
            w16t_mac2<A>(X3, y2, acc[2], X1, y3, acc[3]);
            w16t_mac<A>(X1, y4, acc[4]);
        }
        W16_FENCE();
        W16_MARK(5);
        if (EARLY_X && it + gridDim.x < a.n_tiles) make_x(0u, false);
        W16_MARK(6);
        if (TWO_BARRIERS) w16x_barrier();
        W16_MARK(7);
        W16_TRACE_SUM(7, 4 + rt);
    }
    float* rec = a.rec;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        flush_mapped(rec, G_W_SDF0, N_SDF_IN, lane, acc[ct], w16_unscale(k3), [&](int i) { return 32 * rt + i; }, [&](int c) { return 32 * ct + c; });
    flush_mapped(rec, G_W_SDF0, N_SDF_IN, lane, acc[2], w16_unscale(k3) / w16_grid_shift<A>(), [&](int i) { return 32 * rt + i; },
                 [&](int c) { return N_EMB + 2 * (8 * (c >> 4) + 4 * ((c >> 3) & 1) + (c & 3)) + ((c >> 2) & 1); });
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        flush_mapped(rec, G_W_PTS0, N_E, lane, acc[3 + ct], w16_unscale(k1), [&](int i) { return 32 * rt + i; },
                     [&](int c) { return w16_e_col(ct, c); });
    // d b_pts0: the ones column of e's second column tile
    if (j == W16T_ONES_COL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[G_B_PTS0 + 32 * rt + rowmap(r, h)] = acc[4][r] * w16_unscale(k1);
    }
    // d b_sdf0: this lane's 16 features of row tile rt, summed over the 32 samples of its half
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float t = w16t_half_sum(bacc[q][u]) * w16_unscale(k3);
            if (j == 0) rec[G_B_SDF0 + 32 * rt + 16 * q + 8 * (u >> 2) + 4 * h + (u & 3)] = t;
        }
}

#ifndef W16_EXCHANGE
#define W16_EXCHANGE 1      // experiments: 0 = every wave prepares its own operands (the roles above) behind the lean record, too
#endif
#ifndef W16_TR
#define W16_TR 1            // experiments: 0 = the exchange form with matrix-core transposes (rounds 3-5)
#endif

template <int LAYOUT, typename A, bool RECOMP>
__global__ __launch_bounds__(W16_BLOCK, 2) void decoder_wgrad16_kernel(const float* __restrict__ packed16,
                                                                       const float* __restrict__ feat,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ saved,
                                                                       const float* __restrict__ dact,
                                                                       const float* __restrict__ dsmall,
                                                                       float* __restrict__ partial, uint32_t M,
                                                                       uint32_t n_tiles_all,
                                                                       const uint32_t* __restrict__ live, uint32_t lean_dact) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // live: the chain kernel's live-tile buffer (mipsf_decoder_bwd_chain16): eight lists, visited one after the other
    uint32_t n_tiles = n_tiles_all, live_start[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
        n_tiles = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            live_start[q] = n_tiles;
            n_tiles += (uint32_t)__builtin_amdgcn_readfirstlane((int)live[64 * q + 32]);
        }
    }
    const int j = lane & 31, h = lane >> 5;
    // the two selection matrices (B operands of the transposing MFMAs): lane = column c, half hb supplies k = 8 hb + u;
    // I[q][u] = 1 iff column c = 16 q + 8 (u >> 2) + 4 hb + (u & 3)
    typename A::v8 I[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            I[q][u] = (j == 16 * q + 8 * (u >> 2) + 4 * h + (u & 3)) ? (typename A::elt)1.0f : (typename A::elt)0.0f;
    constexpr int W1_ENTRIES = RT_F1 * T16H_F1 * 64;          // 16-byte operands of one layer-1 image (hi; lo has as many)
    static_assert(T16H_F1 == T16_F1, "layer 1 has no separate bias k-step");
    constexpr bool EXCH = RECOMP && W16_EXCHANGE;
    typedef W16XL<A> L;
    constexpr int P = A::P;
    __shared__ h8 w1img[RECOMP ? P * W1_ENTRIES : 1];
    __shared__ h8 xch[EXCH ? L::NBUF * (L::XE + L::XA + L::XB) : 1];
    __shared__ h8 gimg[EXCH ? L::G_ENTRIES : 1];
    constexpr bool TRF = EXCH && W16_TR;
    __shared__ h8 tscr[TRF ? 8 * (W16T_PLANE / 16) : 1];          // transpose-read form: one plane of scratch per wave
    static_assert(!EXCH || sizeof(h8) * (P * W1_ENTRIES + L::NBUF * (L::XE + L::XA + L::XB) + L::G_ENTRIES + (TRF ? 8 * (W16T_PLANE / 16) : 0))
                               <= 160 * 1024,
                  "the exchange form's images and hand-over buffers must fit the LDS of a CU");
    if constexpr (RECOMP) {
        // planes 0 and 1 of an image sit where the f16 layout has hi and lo; plane 2 (bf16 only) in the buffer's extension
        const h8* img = reinterpret_cast<const h8*>(packed16 + TAIL16_FLOATS);
        const h8* ext = reinterpret_cast<const h8*>(packed16 + PACKED16_FLOATS);
        for (int q = tid; q < W1_ENTRIES; q += W16_BLOCK) {
            w1img[q] = img[OFF16H_F1 / 8 + q];
            w1img[W1_ENTRIES + q] = img[(IMG16H_HALVES + OFF16L_F1) / 8 + q];
            if constexpr (P == 3) w1img[2 * W1_ENTRIES + q] = ext[(EXT16_FWD + OFF16L_F1) / 8 + q];
        }
        if constexpr (EXCH) {       // the chain's two narrow products: operand images of the backward sets, plane by plane
            const h8* bimg = img + OFF16_BWD_HALVES / 8;
            for (int q = tid; q < 4 * 64; q += W16_BLOCK) {
                gimg[q] = bimg[OFF16B_S2T / 8 + q];
                gimg[L::G_PLANE_S2T + q] = bimg[(IMG16B_HALVES + OFF16B_S2T) / 8 + q];
                if constexpr (P == 3) gimg[2 * L::G_PLANE_S2T + q] = ext[(EXT16_BWD + OFF16B_S2T) / 8 + q];
            }
            for (int q = tid; q < 2 * 64; q += W16_BLOCK) {
                gimg[L::G_RGBT + q] = bimg[OFF16B_RGBT / 8 + q];
                gimg[L::G_RGBT + L::G_PLANE_RGBT + q] = bimg[(IMG16B_HALVES + OFF16B_RGBT) / 8 + q];
                if constexpr (P == 3) gimg[L::G_RGBT + 2 * L::G_PLANE_RGBT + q] = ext[(EXT16_BWD + OFF16B_RGBT) / 8 + q];
            }
        }
        __syncthreads();
    }
    const W16Args a = {feat, x, saved, dact, dsmall, partial + (size_t)blockIdx.x * G_STRIDE, M, n_tiles,
                       live, tl_cap(n_tiles_all),
                       {live_start[0], live_start[1], live_start[2], live_start[3], live_start[4], live_start[5],
                        live_start[6], live_start[7]},
                       RECOMP ? w1img : nullptr, RECOMP ? w1img + W1_ENTRIES : nullptr,
                       (EXCH && lean_dact) ? gimg : nullptr,
                       reinterpret_cast<const uint2*>(saved + (((size_t)M + 127) / 128) * 4 * ACT_TILE_FLOATS)};
    if constexpr (TRF) {
        const uint32_t xb0 = w16t_lds_addr(xch);
        const W16T<A> lx = {xb0, xb0 + 16u * L::NBUF * L::XE, xb0 + 16u * L::NBUF * (L::XE + L::XA),
                            w16t_lds_addr(tscr) + (uint32_t)w * W16T_PLANE};
        if (w < 4) w16t_role_a<LAYOUT, A>(a, lx, w, lane);
        else w16t_role_b<LAYOUT, A>(a, lx, I, w - 4, lane);
    } else if constexpr (EXCH) {
        typename A::v8* xp = reinterpret_cast<typename A::v8*>(xch);
        const W16X<A> lx = {xp, xp + L::NBUF * L::XE, xp + L::NBUF * (L::XE + L::XA)};
        if (w < 4) w16x_role_a<LAYOUT, A>(a, lx, I, w, lane);
        else w16x_role_b<LAYOUT, A>(a, lx, I, w - 4, lane);
    } else {
        if (w < 4) {
            if constexpr (RECOMP) w16_role_a_recompute<LAYOUT, A>(a, I, w, lane);
            else w16_role_a<LAYOUT, A>(a, I, w, lane);
        } else {
            w16_role_b<LAYOUT, A>(a, I, w - 4, lane);
        }
    }
}

}  // namespace mipsf

using namespace mipsf;

#ifdef W16_TRACE
extern "C" int mipsf_w16_trace_read(unsigned long long* host, int clear) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(w16_trace), sizeof(unsigned long long) * 2048 * 16) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long z[2048 * 16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(w16_trace), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
namespace mipsf {
uint64_t decoder_tile_words(uint32_t M) { return TL_HEADER + 8ull * tl_cap((uint32_t)(((uint64_t)M + 31) / 32)); }
}  // namespace mipsf

extern "C" int mipsf_decoder_wgrad16(const mipsf_decoder_wgrad16_args* a, void* stream) {
    MIPSF_REQUIRE(a != nullptr, "null argument block");
    MIPSF_REQUIRE(a->struct_size == sizeof(mipsf_decoder_wgrad16_args), "mipsf_decoder_wgrad16_args: struct_size %u, this library expects %u",
                  a->struct_size, (unsigned)sizeof(mipsf_decoder_wgrad16_args));
    const float* packed16 = a->packed16; const float* feat = a->feat; const int feat_layout = a->feat_layout; const float* x = a->x;
    const float* saved = a->saved; const float* dact = a->dact; const uint32_t* tile_live = a->tile_live;
    const mipsf_decoder_grads* grads = a->grads; float* partial = a->partial; const int arithmetic = a->arithmetic;
    const uint32_t flags = a->flags, M = a->M;
    MIPSF_REQUIRE(packed16 == nullptr || a->packed16_floats == 0u || a->packed16_floats == decoder_packed16_floats(arithmetic),
                  "packed16 holds %u floats, arithmetic %d needs %u: packed for the other family?", a->packed16_floats, arithmetic,
                  (unsigned)decoder_packed16_floats(arithmetic));
    if (M == 0) return 0;
    MIPSF_REQUIRE((flags & ~(uint32_t)MIPSF_WGRAD_LEAN_DACT) == 0u, "unknown flags 0x%x", flags);
    const uint32_t lean_dact = (flags & MIPSF_WGRAD_LEAN_DACT) ? 1u : 0u;
    MIPSF_REQUIRE(!lean_dact || (packed16 != nullptr && (arithmetic == MIPSF_PREC_F16X3 || arithmetic == MIPSF_PREC_BF16X6) && W16_EXCHANGE),
                  "the lean gradient record is read by the f16x3 / bf16x6 kernel with packed16 only");
    MIPSF_REQUIRE(packed16 == nullptr || arithmetic == MIPSF_PREC_F16X3 || arithmetic == MIPSF_PREC_BF16X6,
                  "H1 is recomputed by the f16x3 / bf16x6 arithmetic only (packed16 of the same family)");
    MIPSF_REQUIRE(feat && x && saved && dact && partial && grads, "null pointer");
    MIPSF_REQUIRE(feat_layout == MIPSF_FEAT_AOS || feat_layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout");
    MIPSF_REQUIRE(arithmetic == MIPSF_PREC_F16X3 || arithmetic == MIPSF_PREC_BF16X6 || arithmetic == MIPSF_PREC_BF16X3,
                  "arithmetic must be f16x3, bf16x6 or bf16x3");
    MIPSF_REQUIRE(M < (1u << 25), "M = %u: the grid features are addressed through one 4 GB buffer resource", M);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n_tiles = (uint32_t)(((uint64_t)M + 31) / 32);
    const uint64_t n_bt = ((uint64_t)M + 127) / 128;
    const float* dsmall = dact + n_bt * 4 * ACT_TILE_FLOATS;
    const int cus = device_cus();
    if (cus <= 0) return 3;
    uint32_t blocks = n_tiles < (uint32_t)cus ? n_tiles : (uint32_t)cus;
    if (blocks > (uint32_t)W16_MAX_BLOCKS) blocks = (uint32_t)W16_MAX_BLOCKS;
    const uint32_t* live = tile_live;
#define W16(LAY, AR, RC) hipLaunchKernelGGL((decoder_wgrad16_kernel<LAY, AR, RC>), dim3(blocks), dim3(W16_BLOCK), 0, s, packed16, \
                                            feat, x, saved, dact, dsmall, partial, M, n_tiles, live, lean_dact)
#define W16_L(LAY) do { if (arithmetic == MIPSF_PREC_F16X3) { if (packed16) W16(LAY, ArF16, true); else W16(LAY, ArF16, false); } \
                        else if (arithmetic == MIPSF_PREC_BF16X6) { if (packed16) W16(LAY, ArBF3, true); else W16(LAY, ArBF3, false); } \
                        else W16(LAY, ArBF2, false); } while (0)
    if (feat_layout == MIPSF_FEAT_AOS) W16_L(MIPSF_FEAT_AOS); else W16_L(MIPSF_FEAT_LEVEL_MAJOR);
#undef W16_L
#undef W16
    if (int e = check_launch("decoder_wgrad16")) return e;
    return wgrad_reduce_launch(partial, blocks, grads, s);
}
