// Index maps of the register-chained MFMA decoder (shared by the device kernels, the device/host weight
// packers and -- through mipsf_decoder_pack_host -- the CPU layout tests).
//
// The decoder is evaluated TRANSPOSED, one wavefront per 32 samples:
//     Out^T[features x 32 samples] = W[features x K] * In^T[K x 32 samples]
// with v_mfma_f32_32x32x2_f32 (exact fp32).  Lane l = (j = l & 31, h = l >> 5).
//   A operand (weights):    lane holds A[row i = j][k = h]
//   B operand (activations): lane holds B[k = h][col = sample j]
//   C/D (16 regs):           reg r, lane (j,h) = D[row = (r&3) + 8*(r>>2) + 4*h][col = sample j]
// Because the reduction index k may be visited in any order, accumulator register r of row-tile q of the
// PREVIOUS layer is used directly as the B operand of k-step t = 16*q + r of the NEXT layer: the lower
// half-wave contributes feature 32q + rowmap(r,0) and the upper half feature 32q + rowmap(r,1).  The weight
// images below are laid out so that the matching A operand is one contiguous 16-byte-per-lane load.
// Activations therefore never leave registers between layers (no LDS, no barriers in the chain kernels).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIPSF_HD __host__ __device__ inline
#else
#define MIPSF_HD inline
#endif

namespace mipsf {
namespace dl {

constexpr int HID = 128;          // n_hidden = n_hidden_branch
constexpr int N_PE = 48;          // 3 dims * 8 freqs * {sin,cos}
constexpr int N_E = 51;           // [x(3), pe(48)]
constexpr int N_GRID = 32;        // 16 levels * 2 features
constexpr int N_EMB = 64;         // sdf_emb / rgb_emb width
constexpr int N_RGB_IN = 115;     // rgb_emb(64) + e(51)
constexpr int N_SDF_IN = 96;      // sdf_emb(64) + grid(32)
constexpr int N_CLASS = 5;
constexpr int E_SLOTS = 26;       // k-steps that carry e (52 half-slots, one pad)

// k-steps (each = 2 reduction indices) per layer, padded to a multiple of 4 for 16-byte operand loads
constexpr int T_F1 = 28, T_F2 = 64, T_F3 = 48;      // forward: pts0, pts2, sdf0
constexpr int T_B3 = 64, T_B2 = 64, T_B1 = 64;      // backward chain: sdf0^T, pts2^T, pts0^T
constexpr int RT_F1 = 4, RT_F2 = 4, RT_F3 = 4, RT_B3 = 3, RT_B2 = 4, RT_B1 = 2;

// live-tile buffer of the backward pass (decoder16.hip: layout; wgrad16.hip: reader): header words, list capacity
constexpr unsigned TL_HEADER = 512;
MIPSF_HD unsigned tl_cap(unsigned n_tiles) { return ((n_tiles + 63u) / 64u) * 8u + 8u; }

MIPSF_HD int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
MIPSF_HD int feat_of(int q, int r, int h) { return 32 * q + rowmap(r, h); }
// feature carried by half h at k-step t when the B operand is accumulator reg (t/16, t%16)
MIPSF_HD int kfeat(int t, int h) { return feat_of(t >> 4, t & 15, h); }

// e index (into [x0,x1,x2, pe0..pe47]) carried by half h at e-slot t; -1 = padding
MIPSF_HD int eidx(int t, int h) {
    if (t < 24) return 3 + (t >> 3) * 16 + 2 * (t & 7) + h;
    if (t == 24) return h;
    if (t == 25) return h == 0 ? 2 : -1;
    return -1;
}
// inverse of rowmap for an A-operand row i in [0,32): which (reg, half) of an accumulator tile owns it
MIPSF_HD void row_owner(int i, int& r, int& h) {
    h = (i >> 2) & 1;
    r = (i & 3) + 4 * (i >> 3);
}

// ------------------------------------------------------------------ packed buffer (floats)
constexpr int img_floats(int rt, int T) { return rt * T * 64; }
constexpr int OFF_F1 = 0;
constexpr int OFF_F2 = OFF_F1 + img_floats(RT_F1, T_F1);
constexpr int OFF_F3 = OFF_F2 + img_floats(RT_F2, T_F2);
constexpr int OFF_B3 = OFF_F3 + img_floats(RT_F3, T_F3);
constexpr int OFF_B2 = OFF_B3 + img_floats(RT_B3, T_B3);
constexpr int OFF_B1 = OFF_B2 + img_floats(RT_B2, T_B2);
constexpr int OFF_TRGB = OFF_B1 + img_floats(RT_B1, T_B1);   // [h][58 slots][4]   (c < 3)
constexpr int TRGB_SLOTS = 32 + E_SLOTS;
constexpr int OFF_TS2 = OFF_TRGB + 2 * TRGB_SLOTS * 4;      // [h][64 slots][8]   (c < 5)
constexpr int OFF_BIAS = OFF_TS2 + 2 * 64 * 8;              // [layer 0..2][64 slots][h]
constexpr int OFF_BSMALL = OFF_BIAS + 3 * 64 * 2;           // b_rgb0[0..2], pad, b_sdf2[0..4], pad*3
constexpr int PACKED_FLOATS = OFF_BSMALL + 12;

// element (rt, t, lane) of an A image; stored as [rt][t/4][lane][t%4]
MIPSF_HD int img_index(int T, int rt, int t, int lane) { return ((rt * (T >> 2) + (t >> 2)) * 64 + lane) * 4 + (t & 3); }

struct W {   // nn.Linear weights, row-major [out][in]
    const float *w_pts0, *b_pts0, *w_pts2, *b_pts2, *w_rgb0, *b_rgb0, *w_sdf0, *b_sdf0, *w_sdf2, *b_sdf2;
};

// value of packed[idx]; every packed float is produced by exactly this function (device kernel: one thread
// per idx; host mirror: a loop), so host tests exercise the very code the GPU runs.
MIPSF_HD float packed_value(const W& w, int idx) {
    if (idx < OFF_TRGB) {
        int base, T, kind;
        if (idx < OFF_F2) { base = OFF_F1; T = T_F1; kind = 0; }
        else if (idx < OFF_F3) { base = OFF_F2; T = T_F2; kind = 1; }
        else if (idx < OFF_B3) { base = OFF_F3; T = T_F3; kind = 2; }
        else if (idx < OFF_B2) { base = OFF_B3; T = T_B3; kind = 3; }
        else if (idx < OFF_B1) { base = OFF_B2; T = T_B2; kind = 4; }
        else { base = OFF_B1; T = T_B1; kind = 5; }
        const int rel = idx - base;
        const int u = rel & 3, lane = (rel >> 2) & 63, g = rel >> 8;   // g = rt * (T/4) + t4
        const int t4n = T >> 2;
        const int rt = g / t4n, t = (g - rt * t4n) * 4 + u;
        const int i = lane & 31, h = lane >> 5;
        const int row = 32 * rt + i;
        switch (kind) {
            case 0: {   // pts0: H1^T = W1 * e^T
                const int e = eidx(t, h);
                return e < 0 ? 0.f : w.w_pts0[row * N_E + e];
            }
            case 1:     // pts2: H2^T = W2 * H1^T
                return w.w_pts2[row * HID + kfeat(t, h)];
            case 2: {   // sdf0: G3^T = Ws1 * [sdf_emb; grid]^T
                const int src = t < 32 ? kfeat(t, h) : 64 + 2 * (t - 32) + h;
                return w.w_sdf0[row * N_SDF_IN + src];
            }
            case 3:     // d[sdf_emb; grid]^T = Ws1^T * dG3^T      (row = input index 0..95)
                return w.w_sdf0[kfeat(t, h) * N_SDF_IN + row];
            case 4:     // dH1^T = W2^T * dH2^T
                return w.w_pts2[kfeat(t, h) * HID + row];
            default: {  // d e^T = W1^T * dG1^T; output row i of tile rt lands in e-slot (16*rt + r', h')
                int r2, h2;
                row_owner(i, r2, h2);
                const int e = eidx(16 * rt + r2, h2);
                return e < 0 ? 0.f : w.w_pts0[kfeat(t, h) * N_E + e];
            }
        }
    }
    if (idx < OFF_TS2) {            // rgb table: slots 0..31 = rgb_emb regs (q=2,3), 32..57 = e slots
        const int rel = idx - OFF_TRGB;
        const int c = rel & 3, slot = (rel >> 2) % TRGB_SLOTS, h = (rel >> 2) / TRGB_SLOTS;
        if (c >= 3) return 0.f;
        if (slot < 32) return w.w_rgb0[c * N_RGB_IN + (32 * (slot >> 4) + rowmap(slot & 15, h))];
        const int e = eidx(slot - 32, h);
        return e < 0 ? 0.f : w.w_rgb0[c * N_RGB_IN + N_EMB + e];
    }
    if (idx < OFF_BIAS) {           // sdf2 table: slot = q*16 + r
        const int rel = idx - OFF_TS2;
        const int c = rel & 7, slot = (rel >> 3) & 63, h = rel >> 9;
        return c < N_CLASS ? w.w_sdf2[c * HID + kfeat(slot, h)] : 0.f;
    }
    if (idx < OFF_BSMALL) {         // hidden-layer biases in accumulator order
        const int rel = idx - OFF_BIAS;
        const int h = rel & 1, slot = (rel >> 1) & 63, layer = rel >> 7;
        const float* b = layer == 0 ? w.b_pts0 : (layer == 1 ? w.b_pts2 : w.b_sdf0);
        return b[kfeat(slot, h)];
    }
    const int rel = idx - OFF_BSMALL;
    if (rel < 3) return w.b_rgb0[rel];
    if (rel >= 4 && rel < 4 + N_CLASS) return w.b_sdf2[rel - 4];
    return 0.f;
}

// ------------------------------------------------------------------ f16 operand images (decoder16.hip)
// The same transposed, register-chained evaluation on v_mfma_f32_32x32x16_f16: one k-step carries 16 reduction
// indices, the A/B operands are 8 halves per lane.  Lane (j, h) of the B operand supplies elements u = 0..7 = the
// 8 input features of sample j that k-step t assigns to half h; with the C/D layout above (it does not depend on the
// data type) accumulator registers 8m .. 8m+7 of row tile q of the previous layer are exactly the B operand of
// k-step t = 2q + m of the next one, so the chaining needs no data movement, only a float -> half conversion.
// fp32 values are carried as hi + lo halves (hi = rne(v), lo = rne(v - hi): 22 significant bits); a product is
// hi*hi + hi*lo + lo*hi accumulated in fp32 (the dropped lo*lo term is 2^-22 relative).
//
// Biases ride on the matrix pipe: B-operand elements that are the constant 1.0 meet the bias in the A operand, so the
// accumulators start from the inline constant 0 (no per-register initialisation, no bias loads).  A bias b occupies
// TWO elements, b_hi = rne16(b) and b - b_hi, both against 1.0: exact to 22 bits with ONE MFMA (no lo products).
// Layer 1 has padding to spare (elements u = 2, 3 of half 0 at k-step 3); layers 2 and 3 get a bias k-step of their
// own in FRONT (t = 0: half 0, u = 0, 1) that exists in the hi image set only.
// RANGE of the f16 halves.  A weight of 0.1 has lo = 0.1 * 2^-12 = 2e-5, below f16's smallest normal (6.1e-5): its lo
// half would keep a handful of bits; a freshly initialised hash grid has FEATURES of 1e-4 (tcnn's U(-1e-4, 1e-4)) whose
// lo half would vanish altogether (measured: the 51-iteration sequence drifted 100x further than with fp32 MFMA).  So
// every operand image stores weight * 2^W16_SHIFT (exact; |w| < 2^(15 - W16_SHIFT) = 32 stays finite), every completed
// accumulator is multiplied by 2^-W16_SHIFT (exact), and the grid features enter layer 3 as feature * 2^G16_SHIFT against
// weight columns stored as weight * 2^(W16_SHIFT - G16_SHIFT).
constexpr int W16_SHIFT = 10, G16_SHIFT = 12;
MIPSF_HD float pow2f(int e) { float r = 1.0f; for (int i = 0; i < (e < 0 ? -e : e); ++i) r *= (e < 0 ? 0.5f : 2.0f); return r; }
// The bf16 mode (np = 3 planes) has fp32's exponent range: its images hold the weights themselves and its kernels neither
// unscale accumulators nor upscale the grid features (a power-of-two scale commutes with the cut into bf16 pieces, so the
// results would be bit-identical with it -- only the multiplications go).
MIPSF_HD float w16_scale(int np) { return np == 3 ? 1.0f : pow2f(W16_SHIFT); }
MIPSF_HD float w16_grid_col_scale(int np) { return np == 3 ? 1.0f : pow2f(W16_SHIFT - G16_SHIFT); }
constexpr int T16_F1 = 4, T16_F2 = 8, T16_F3 = 6;          // data k-steps (x16 inputs): e (52 -> 64), H1 (128), [sdf_emb | grid] (96)
constexpr int T16H_F1 = 4, T16H_F2 = 9, T16H_F3 = 7;       // k-steps of the hi images (bias k-step first for layers 2, 3)
constexpr int BIAS16_T = 3, BIAS16_U = 2;                  // layer 1: where the two bias elements sit (half 0)
// e index carried by element (t, h, u) of layer 1's B operand (-1 = padding): the 26 e-slots of the fp32 kernel, 8 per k-step
MIPSF_HD int e16(int t, int h, int u) { const int s = 8 * t + u; return s < E_SLOTS ? eidx(s, h) : -1; }
// feature carried by element (t, h, u) when the B operand is accumulator registers 8*(t&1)+u of row tile t>>1
MIPSF_HD int kfeat16(int t, int h, int u) { return 32 * (t >> 1) + rowmap(8 * (t & 1) + u, h); }
// input index of sdf_linear.0 ([sdf_emb(64) | grid(32)]) carried by element (t, h, u): grid level 8*(t-4)+u, feature h
MIPSF_HD int src16_f3(int t, int h, int u) { return t < 4 ? kfeat16(t, h, u) : N_EMB + 2 * (8 * (t - 4) + u) + h; }
// images [layer][rt][t][lane][u]: halves; all hi images first, then all lo images (data k-steps only)
constexpr int img16_halves(int rt, int T) { return rt * T * 64 * 8; }
constexpr int OFF16H_F1 = 0;
constexpr int OFF16H_F2 = OFF16H_F1 + img16_halves(RT_F1, T16H_F1);
constexpr int OFF16H_F3 = OFF16H_F2 + img16_halves(RT_F2, T16H_F2);
constexpr int IMG16H_HALVES = OFF16H_F3 + img16_halves(RT_F3, T16H_F3);   // 40 960 halves = 80 KB
constexpr int OFF16L_F1 = 0;
constexpr int OFF16L_F2 = OFF16L_F1 + img16_halves(RT_F1, T16_F1);
constexpr int OFF16L_F3 = OFF16L_F2 + img16_halves(RT_F2, T16_F2);
constexpr int IMG16L_HALVES = OFF16L_F3 + img16_halves(RT_F3, T16_F3);    // 36 864 halves = 72 KB
constexpr int TAIL_FLOATS = PACKED_FLOATS - OFF_TRGB;                      // head tables (+ fp32 biases), as in `packed`
// ---- the two narrow heads (rgb_linear.0: 3 outputs from [rgb_emb | e]; sdf_linear.2: 5 logits from H3) ride on the matrix
// pipe as well: one accumulator tile each, 8 k-steps, f16x3.  (On the vector ALU -- ~500 fp32 fmas + 187 table reads from
// LDS per tile -- they were 30 of the evaluation forward's 72 us: tools/micro/fwd_probe.py.)  Only a few of the 32 A-operand
// rows carry weights, so the images are stored COMPACT: per (k-step, plane) `slots` 16-byte operands -- one per used
// (row, half) -- plus one of zeros that every other lane reads (a broadcast).
//   sdf head   rows 0..3 = classes 0..3 and rows 4..7 = classes 0..3 AGAIN, row 8 = row 12 = class 4: with the C/D layout
//              (row = (r & 3) + 8 (r >> 2) + 4 h) BOTH halves of a lane pair end up with logit c in register c -- no exchange
//   rgb head   rows 0..2 = colours (half 0 registers 0..2; half 1 does not need them)
// [sdf: t][plane hi, lo][21 slots][8 halves], then [rgb: t][plane][7 slots][8], then 12 floats of head biases (fp32).
constexpr int T16_HEAD = 8;
constexpr int HEAD16_SDF_SLOTS = 21, HEAD16_RGB_SLOTS = 7;
constexpr int HEAD16_SDF_HALVES = T16_HEAD * 2 * HEAD16_SDF_SLOTS * 8;    // 2688
constexpr int HEAD16_RGB_HALVES = T16_HEAD * 2 * HEAD16_RGB_SLOTS * 8;    // 896
constexpr int HEAD16_HALVES = HEAD16_SDF_HALVES + HEAD16_RGB_HALVES;
constexpr int OFF16_BSMALL = HEAD16_HALVES / 2;                            // floats: b_rgb0[0..2], pad, b_sdf2[0..4], pad x 3
constexpr int TAIL16_FLOATS = OFF16_BSMALL + 12;                           // the front of `packed16` (7216 bytes, LDS-resident)
static_assert(TAIL16_FLOATS % 4 == 0 && HEAD16_HALVES % 8 == 0, "16-byte pieces");
// operand slot of A-operand lane (row i, half hA) in the compact images
MIPSF_HD int head16_sdf_slot(int i, int hA) {
    const int ri = i < 8 ? i : (i == 8 ? 8 : (i == 12 ? 9 : 10));
    return ri < 10 ? 2 * ri + hA : HEAD16_SDF_SLOTS - 1;
}
MIPSF_HD int head16_rgb_slot(int i, int hA) { return i < 3 ? 2 * i + hA : HEAD16_RGB_SLOTS - 1; }
// ---- backward-chain images (decoder16.hip: decoder16_bwd_kernel), same element order [rt][t][lane][u]; hi set, then lo set
//   S2T  dH3^T  = Ws2^T  dlogits^T      4 row tiles x 1 k-step  (k = class c: half 0, u = c < 5)
//   B3   d[sdf_emb|grid]^T = Ws1^T dG3^T 3 x 8                  (k = hidden feature kfeat16)
//   RGBT d rgb_emb^T = Wrgb[:, :64]^T drgb^T  2 x 1             (k = colour c: half 0, u = c < 3; rows = rgb_emb 32q + i)
//   B2   dH1^T  = W2^T   dH2^T          4 x 8
//   B1   d e^T  = W1^T   dG1^T          2 x 9   (k-steps 0..7: hidden features; k-step 8: the rgb head's share, k = c;
//                                                output row i of tile rt lands in e-slot (16 rt + r', h') = row_owner(i))
constexpr int T16_S2T = 1, T16_B3 = 8, T16_RGBT = 1, T16_B2 = 8, T16_B1 = 9;
constexpr int RT16_S2T = 4, RT16_B3 = 3, RT16_RGBT = 2, RT16_B2 = 4, RT16_B1 = 2;
constexpr int OFF16B_S2T = 0;
constexpr int OFF16B_B3 = OFF16B_S2T + img16_halves(RT16_S2T, T16_S2T);
constexpr int OFF16B_RGBT = OFF16B_B3 + img16_halves(RT16_B3, T16_B3);
constexpr int OFF16B_B2 = OFF16B_RGBT + img16_halves(RT16_RGBT, T16_RGBT);
constexpr int OFF16B_B1 = OFF16B_B2 + img16_halves(RT16_B2, T16_B2);
constexpr int IMG16B_HALVES = OFF16B_B1 + img16_halves(RT16_B1, T16_B1);   // 40 960 halves = 80 KB per set
MIPSF_HD float img16b_weight(const W& w, int idx, int np = 2) {
    int base, T, kind;
    if (idx < OFF16B_B3) { base = OFF16B_S2T; T = T16_S2T; kind = 0; }
    else if (idx < OFF16B_RGBT) { base = OFF16B_B3; T = T16_B3; kind = 1; }
    else if (idx < OFF16B_B2) { base = OFF16B_RGBT; T = T16_RGBT; kind = 2; }
    else if (idx < OFF16B_B1) { base = OFF16B_B2; T = T16_B2; kind = 3; }
    else { base = OFF16B_B1; T = T16_B1; kind = 4; }
    const int rel = idx - base;
    const int u = rel & 7, lane = (rel >> 3) & 63, g = rel >> 9;
    const int rt = g / T, t = g - rt * T;
    const int i = lane & 31, h = lane >> 5;
    const int row = 32 * rt + i;
    const float sc = w16_scale(np);                    // every backward image is scaled alike (see RANGE above)
    switch (kind) {
        case 0: return (h == 0 && u < N_CLASS) ? w.w_sdf2[u * HID + row] * sc : 0.f;
        case 1: return w.w_sdf0[kfeat16(t, h, u) * N_SDF_IN + row] * sc;
        case 2: return (h == 0 && u < 3) ? w.w_rgb0[u * N_RGB_IN + row] * sc : 0.f;
        case 3: return w.w_pts2[kfeat16(t, h, u) * HID + row] * sc;
        default: {
            int r2, h2;
            row_owner(i, r2, h2);
            const int e = eidx(16 * rt + r2, h2);
            if (e < 0) return 0.f;
            if (t == 8) return (h == 0 && u < 3) ? w.w_rgb0[u * N_RGB_IN + N_EMB + e] * sc : 0.f;
            return w.w_pts0[kfeat16(t, h, u) * N_E + e] * sc;
        }
    }
}
constexpr int OFF16_BWD_HALVES = IMG16H_HALVES + IMG16L_HALVES;            // where the backward sets start (in halves)
// [head images + head biases (TAIL16) | fwd hi | fwd lo | bwd hi | bwd lo]
constexpr int PACKED16_FLOATS = TAIL16_FLOATS + (IMG16H_HALVES + IMG16L_HALVES + 2 * IMG16B_HALVES) / 2;
// ---- the bf16 mode ("bf16x6", NP = 3 planes): every fp32 value is carried EXACTLY as three bf16 pieces p0 = rne(v),
// p1 = rne(v - p0), p2 = v - p0 - p1 (8 + 8 + 8 significant bits).  Planes 0 and 1 of every image sit where the f16 layout
// has its hi and lo halves (same offsets, same sizes: the kernels address both alike); plane 2 follows the f16-sized buffer
// as an EXTENSION that stays in L2 (it feeds one of the six products of a k-step):
//     [head sdf p2: t][slots][8] [head rgb p2: t][slots][8] [forward p2: the lo image's shape] [backward p2: one image set]
constexpr int EXT16_HEAD_SDF = 0;
constexpr int EXT16_HEAD_RGB = EXT16_HEAD_SDF + T16_HEAD * HEAD16_SDF_SLOTS * 8;
constexpr int EXT16_FWD = EXT16_HEAD_RGB + T16_HEAD * HEAD16_RGB_SLOTS * 8;
constexpr int EXT16_BWD = EXT16_FWD + IMG16L_HALVES;
constexpr int EXT16_HALVES = EXT16_BWD + IMG16B_HALVES;
static_assert(EXT16_FWD % 8 == 0 && EXT16_HALVES % 8 == 0, "16-byte pieces");
constexpr int PACKED16X_FLOATS = PACKED16_FLOATS + EXT16_HALVES / 2;       // size of a bf16x6 `packed16`
// fp32 -> bf16, round to nearest even, as the upper 16 bits (finite inputs)
MIPSF_HD uint16_t bf16_bits(float v) {
    union { float f; uint32_t u; } c;
    c.f = v;
    return (uint16_t)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}
MIPSF_HD float bf16_value(uint16_t b) {
    union { float f; uint32_t u; } c;
    c.u = (uint32_t)b << 16;
    return c.f;
}
// piece `which` (0, 1, 2) of the exact three-piece cut
MIPSF_HD float bf16_piece(float v, int which) {
    const float p0 = bf16_value(bf16_bits(v));
    if (which == 0) return p0;
    const float r = v - p0;
    const float p1 = bf16_value(bf16_bits(r));
    return which == 1 ? p1 : r - p1;
}
// the pieces a bias contributes (each meets a constant 1.0 of the B operand): f16 modes two, which = 0 -> rne16(b) as a
// float, 1 -> b - rne16(b); bf16 mode (np = 3) the three bf16 pieces
MIPSF_HD float bias16_part(float b, int which, int np = 2) {
    if (np == 3) return bf16_piece(b, which);
    const float hi = (float)(_Float16)b;
    return which == 0 ? hi : b - hi;
}
// fp32 value behind element idx of the HI image set (the packer stores rne16 of it there and, for data k-steps, the
// residual in the lo set at img16_lo_index)
MIPSF_HD float img16_weight(const W& w, int idx, int np = 2) {
    int base, T, kind;
    if (idx < OFF16H_F2) { base = OFF16H_F1; T = T16H_F1; kind = 0; }
    else if (idx < OFF16H_F3) { base = OFF16H_F2; T = T16H_F2; kind = 1; }
    else { base = OFF16H_F3; T = T16H_F3; kind = 2; }
    const int rel = idx - base;
    const int u = rel & 7, lane = (rel >> 3) & 63, g = rel >> 9;       // g = rt * T + t
    const int rt = g / T, t = g - rt * T;
    const int i = lane & 31, h = lane >> 5;
    const int row = 32 * rt + i;
    const float sc = w16_scale(np);
    if (kind == 0) {
        if (t == BIAS16_T && h == 0 && u >= BIAS16_U && u < BIAS16_U + np) return bias16_part(w.b_pts0[row] * sc, u - BIAS16_U, np);
        const int e = e16(t, h, u);
        return e < 0 ? 0.f : w.w_pts0[row * N_E + e] * sc;
    }
    if (t == 0) {                                                       // bias k-step
        const float* b = kind == 1 ? w.b_pts2 : w.b_sdf0;
        return (h == 0 && u < np) ? bias16_part(b[row] * sc, u, np) : 0.f;
    }
    if (kind == 1) return w.w_pts2[row * HID + kfeat16(t - 1, h, u)] * sc;
    const int src = src16_f3(t - 1, h, u);
    return w.w_sdf0[row * N_SDF_IN + src] * (src < N_EMB ? sc : w16_grid_col_scale(np));
}
// fp32 value behind half idx of the compact head images (both planes hold the same value here: the packer stores rne16(v)
// in plane 0 and rne16(v - rne16(v)) in plane 1); plane = which plane idx belongs to
// index in the plane-2 extension (EXT16_HEAD_*) of the element behind half idx of the compact head images
MIPSF_HD int head16_ext_index(int idx) {
    const bool sdf = idx < HEAD16_SDF_HALVES;
    const int rel = sdf ? idx : idx - HEAD16_SDF_HALVES;
    const int slots = sdf ? HEAD16_SDF_SLOTS : HEAD16_RGB_SLOTS;
    const int u = rel & 7, g = rel >> 3;
    const int slot = g % slots, t = (g / slots) >> 1;
    return (sdf ? EXT16_HEAD_SDF : EXT16_HEAD_RGB) + (t * slots + slot) * 8 + u;
}
MIPSF_HD float head16_weight(const W& w, int idx, int& plane, int np = 2) {
    const float sc = w16_scale(np);
    const bool sdf = idx < HEAD16_SDF_HALVES;
    const int rel = sdf ? idx : idx - HEAD16_SDF_HALVES;
    const int slots = sdf ? HEAD16_SDF_SLOTS : HEAD16_RGB_SLOTS;
    const int u = rel & 7, g = rel >> 3;
    const int slot = g % slots, tp = g / slots;
    plane = tp & 1;
    const int t = tp >> 1;
    if (slot == slots - 1) return 0.f;
    const int ri = slot >> 1, hA = slot & 1;
    if (sdf) {
        const int c = ri < 8 ? (ri & 3) : 4;
        return w.w_sdf2[c * HID + kfeat16(t, hA, u)] * sc;
    }
    if (t < 4) return w.w_rgb0[ri * N_RGB_IN + kfeat16(t, hA, u)] * sc;          // rgb_emb = H2 row tiles 2, 3
    const int e = e16(t - 4, hA, u);
    return e < 0 ? 0.f : w.w_rgb0[ri * N_RGB_IN + N_EMB + e] * sc;
}
// index in the LO image set of hi-image element idx, or -1 (bias k-steps have no lo part)
MIPSF_HD int img16_lo_index(int idx) {
    int base, T, Tl, lbase;
    if (idx < OFF16H_F2) { base = OFF16H_F1; T = T16H_F1; Tl = T16_F1; lbase = OFF16L_F1; }
    else if (idx < OFF16H_F3) { base = OFF16H_F2; T = T16H_F2; Tl = T16_F2; lbase = OFF16L_F2; }
    else { base = OFF16H_F3; T = T16H_F3; Tl = T16_F3; lbase = OFF16L_F3; }
    const int rel = idx - base;
    const int within = rel & 511, g = rel >> 9;
    const int rt = g / T, t = g - rt * T;
    const int tl = t - (T - Tl);                                         // data k-step number
    if (tl < 0) return -1;
    return lbase + ((rt * Tl + tl) << 9) + within;
}

// --------------------------------------------------------- activations kept between kernels
// one 32-sample wave tile = 192 accumulator registers x 64 lanes, stored [tile][g = slot/4][lane][4]
constexpr int ACT_SLOTS = 192;                 // 3 matrices x 4 row tiles x 16 regs
constexpr int ACT_TILE_FLOATS = ACT_SLOTS * 64;
MIPSF_HD int64_t act_index(int64_t tile, int mat, int rt, int r, int lane) {
    const int slot = mat * 64 + rt * 16 + r;
    return (tile * (ACT_SLOTS / 4) + (slot >> 2)) * 256 + lane * 4 + (slot & 3);
}

// ---------------------------------------------- natural-layout gradient record (partials + reduce)
constexpr int G_W_PTS0 = 0;
constexpr int G_B_PTS0 = G_W_PTS0 + HID * N_E;
constexpr int G_W_PTS2 = G_B_PTS0 + HID;
constexpr int G_B_PTS2 = G_W_PTS2 + HID * HID;
constexpr int G_W_RGB0 = G_B_PTS2 + HID;
constexpr int G_B_RGB0 = G_W_RGB0 + 3 * N_RGB_IN;
constexpr int G_W_SDF0 = G_B_RGB0 + 3;
constexpr int G_B_SDF0 = G_W_SDF0 + HID * N_SDF_IN;
constexpr int G_W_SDF2 = G_B_SDF0 + HID;
constexpr int G_B_SDF2 = G_W_SDF2 + N_CLASS * HID;
constexpr int G_TOTAL = G_B_SDF2 + N_CLASS;      // 36577 parameters
constexpr int G_STRIDE = (G_TOTAL + 63) / 64 * 64;

}  // namespace dl
}  // namespace mipsf
