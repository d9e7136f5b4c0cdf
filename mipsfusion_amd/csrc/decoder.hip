// SDF/colour decoder MLP (model/decoder.py:32-75 of the reference) for gfx950, forward and backward,
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, bitwise an fmaf chain).
//
// Kernels
//   decoder_pack_kernel    nn.Linear weights -> MFMA A-operand images (decoder_layout.h)
//   decoder_fwd_kernel     one wave = 32 samples; the 51->128->128->{115->3, 96->128->5} chain stays in
//                          accumulator registers from the positional encoding to the softmax (no LDS).
//   decoder_bwd_lds_kernel activation-gradient chain; each gradient tile is staged in a per-wave LDS buffer that
//                          feeds the next layer's B operand; emits d(feat), d(x) and the pre-activation
//                          gradients the weight-gradient kernel needs.
//   decoder_wgrad_kernel   dW = dOut^T * In as MFMA GEMMs whose reduction index is the SAMPLE: a block of
//                          4 waves transposes 128 samples through LDS, each wave owns 11 of the 44 output
//                          tiles and keeps them in registers across its whole share of the batch.
//   decoder_wgrad_reduce   sum of the per-block partials, accumulated into the .grad tensors.
//   (decoder_fwd_lds_kernel: persistent forward with the weight images in LDS for large batches; both forward
//    kernels also exist in an SDF-only form (JointEncoding.query_sdf, scene_rep.py:106-107), see decoder_fwd_tile.)
//
// What these kernels are written around (DESIGN_NOTES.md 4b): a wave's vector instructions are NOT hidden behind its own
// MFMAs (5.7 cycles of kernel time each at one wave per SIMD, 2.5 at two), so everything between the MFMAs is kept
// to as few instructions as possible: scalar tile bases, buffer addressing, ReLU masks as bits, hardware sin/cos
// after a two-constant range reduction, packed fp32 fma in the narrow heads, bias gradients from a ones row.
//
// Roofline: MFMA-bound.  72 370 FLOP/sample forward, 217 110 forward+backward; the three big layers
// are 138 k-steps x 4 row tiles = 552 MFMAs (64 cycles each) per 32 samples forward.
#include "decoder_dev.h"

namespace mipsf {
using namespace dl;


// ================================================================================ forward
// one wave, one tile of 32 samples; img1/2/3 = A-operand images of the three big layers (global or LDS)
// SDF_ONLY: what JointEncoding.query_sdf keeps of the decoder output (column 3, scene_rep.py:106-107; decoder.py:53-75):
// layer 2 only produces its sdf_emb half (2 of 4 row tiles), no rgb head, no entropy; out = sdf [M].  This is what
// RandomOptimizer.get_fitness and the mesher's SDF grid queries call -- 23 % fewer MFMAs, a tenth of the output.
template <bool PE_INTERNAL, int LAYOUT, bool SAVE, bool SDF_ONLY = false>
__device__ __forceinline__ void decoder_fwd_tile(const float* tail, const float4* img1,
                                                 const float4* img2, const float4* img3,
                                                 const float* __restrict__ feat, const float* __restrict__ x,
                                                 const float* __restrict__ embed_pos, float* __restrict__ out,
                                                 float* __restrict__ saved, uint32_t M, int pin, int64_t tile,
                                                 int lane) {
    const int j = lane & 31, h = lane >> 5;
    const uint32_t s_raw = (uint32_t)(tile * 32 + j);
    const bool live = s_raw < M;
    const uint32_t s = live ? s_raw : M - 1;   // tail lanes recompute the last sample (finite values, no stores)
    // `tile` is wave-uniform (the callers pass it through readfirstlane): the saved-activation tile gets its own
    // buffer resource, its 48 stores need no vector address arithmetic
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const srd_t sv = make_srd(SAVE ? saved + (size_t)tile * ACT_TILE_FLOATS : saved, SAVE ? ACT_TILE_FLOATS * 4 : 0);

    float ev[E_SLOTS];
    load_e<PE_INTERNAL>(x, embed_pos, s, h, ev);

    // ---- layer 1: pts_linear.0 + ReLU
    f32x16 H1[4];
    load_bias(tail, 0, h, H1);
    mfma_layer<RT_F1, T_F1>(img1, lane, H1,
                            [&](int t) { return t < E_SLOTS ? ev[t] : 0.0f; });
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) H1[rt][r] = fmaxf(H1[rt][r], 0.0f);

    uint32_t m1[2] = {0u, 0u};
    if (SAVE) relu_masks(H1, m1);

    // ---- layer 2: pts_linear.2 -> [sdf_emb | rgb_emb]   (H1 is written out one 16-byte group per k-group)
    constexpr int RT2 = SDF_ONLY ? 2 : RT_F2;
    f32x16 H2[RT2];
    load_bias(tail, 1, h, H2);
    mfma_layer<RT2, T_F2>(img2, lane, H2,
                          [&](int t) { return H1[t >> 4][t & 15]; },
                          [&](int t4) { if constexpr (SAVE) { if (pin == 0) buf_store_act_piece(sv, lane16, 0, H1, t4); } });

    // ---- rgb_linear.0 on the vector ALU (3 outputs): this lane's half of every dot product, then one swap
    // (two of the three / four of the five running sums advance with one packed v_pk_fma_f32 each: same fmaf per
    //  component, a third fewer vector instructions in the two heads)
    float rgb[3] = {0.f, 0.f, 0.f};
    if constexpr (!SDF_ONLY) {
        float pr[3];
        const float4* trgb = reinterpret_cast<const float4*>(tail) + h * TRGB_SLOTS;
        f32x2 p01 = {0.f, 0.f};
        float p2 = 0.f;
#pragma unroll
        for (int slot = 0; slot < 32; ++slot) {
            const float4 wv = trgb[slot];
            const float v = H2[2 + (slot >> 4)][slot & 15];
            p01 = __builtin_elementwise_fma(f32x2{wv.x, wv.y}, f32x2{v, v}, p01);
            p2 = fmaf(wv.z, v, p2);
        }
#pragma unroll
        for (int t = 0; t < E_SLOTS; ++t) {
            const float4 wv = trgb[32 + t];
            p01 = __builtin_elementwise_fma(f32x2{wv.x, wv.y}, f32x2{ev[t], ev[t]}, p01);
            p2 = fmaf(wv.z, ev[t], p2);
        }
        pr[0] = p01.x, pr[1] = p01.y, pr[2] = p2;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = (pr[c] + __shfl_xor(pr[c], 32, 64)) + tail[OFF_BSMALL - OFF_TRGB + c];
    }

    // ---- layer 3: sdf_linear.0 + ReLU on [sdf_emb (regs of H2 tiles 0,1) | grid features (loaded)]
    float gf[16];
    if (LAYOUT == MIPSF_FEAT_LEVEL_MAJOR && M < (1u << 24)) {       // 16 levels x M x 8 B within one 4 GB resource
        const srd_t fs = make_srd(feat, M * 128u);
        const uint32_t voff = (2u * s + (uint32_t)h) * 4u;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            gf[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fs, voff, (uint32_t)u * M * 8u, 0));
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) gf[u] = load_feat<LAYOUT>(feat, s, u, h, M);
    }
    f32x16 H3[4];
    load_bias(tail, 2, h, H3);
    mfma_layer<RT_F3, T_F3>(img3, lane, H3,
                            [&](int t) { return t < 32 ? H2[t >> 4][t & 15] : gf[t - 32]; },
                            [&](int t4) {                       // 16 groups of H2 over 12 k-groups
                                if constexpr (SAVE) {
                                    if (pin == 0) {
                                        buf_store_act_piece(sv, lane16, 1, H2, t4);
                                        if (t4 < 4) buf_store_act_piece(sv, lane16, 1, H2, 12 + t4);
                                    }
                                }
                            });
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) H3[rt][r] = fmaxf(H3[rt][r], 0.0f);
    if (SAVE) {
        buf_store_act(sv, lane16, 2, H3);
        uint32_t m3[2];
        relu_masks(H3, m3);
        uint2* mk = reinterpret_cast<uint2*>(saved + (((size_t)M + 127) / 128) * 4 * ACT_TILE_FLOATS) +
                    (size_t)tile * (MASK_TILE_WORDS / 2) + lane;
        mk[0] = make_uint2(m1[0], m1[1]);
        mk[64] = make_uint2(m3[0], m3[1]);
    }

    // ---- sdf_linear.2 (5 logits) on the vector ALU, softmax, entropy, expected class -> SDF
    float pl[N_CLASS];
    {
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
        }
        pl[0] = q01.x, pl[1] = q01.y, pl[2] = q23.x, pl[3] = q23.y, pl[4] = q4;
    }
    float lg[N_CLASS], mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) {
        lg[c] = (pl[c] + __shfl_xor(pl[c], 32, 64)) + tail[OFF_BSMALL - OFF_TRGB + 4 + c];
        mx = fmaxf(mx, lg[c]);
    }
    float p[N_CLASS], den = 0.f;
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) {
        p[c] = expf(lg[c] - mx);
        den += p[c];
    }
    float ent = 0.f, cls = 0.f;
#pragma unroll
    for (int c = 0; c < N_CLASS; ++c) {
        p[c] = p[c] / den;
        ent += p[c] * log2f(p[c] + 1e-5f);
        cls += p[c] * (float)c;
    }
    const float sdf = (cls / 4.0f - 0.5f) * 2.0f;
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

template <bool PE_INTERNAL, int LAYOUT, bool SAVE, bool SDF_ONLY = false>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(DEC_BLOCK, 2) void decoder_fwd_kernel(const float* __restrict__ packed,
                                                                const float* __restrict__ feat,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ embed_pos,
                                                                float* __restrict__ out, float* __restrict__ saved,
                                                                uint32_t M, int pin) {
    // biases and the two small head tables (7.5 KB): broadcast LDS reads instead of ~380 L2 loads per lane and tile
    __shared__ float4 tailbuf[TAIL_F4];
    for (int q = threadIdx.x; q < TAIL_F4; q += DEC_BLOCK) tailbuf[q] = reinterpret_cast<const float4*>(packed + OFF_TRGB)[q];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (DEC_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile * 32 >= (int64_t)M) return;
    decoder_fwd_tile<PE_INTERNAL, LAYOUT, SAVE, SDF_ONLY>(reinterpret_cast<const float*>(tailbuf), reinterpret_cast<const float4*>(packed + OFF_F1),
                                                reinterpret_cast<const float4*>(packed + OFF_F2),
                                                reinterpret_cast<const float4*>(packed + OFF_F3), feat, x, embed_pos,
                                                out, saved, M, pin, tile, lane);
}

// Persistent variant for large batches: one workgroup of 8 waves per CU keeps the three forward weight images
// (140 KB) in LDS for its whole share of the batch.  The A operands then arrive over the LDS path, so the trickled
// activation stores are alone in the vector-memory queue: on gfx9 loads and stores retire through ONE in-order
// counter (vmcnt), and an L2 weight load queued behind a store to HBM waits for that store.
constexpr int FWD_LDS_FLOATS = OFF_B3 - OFF_F1;
constexpr int FWD_LDS_BYTES = FWD_LDS_FLOATS * 4 + TAIL_F4 * 16;
constexpr int FWD_LDS_BLOCK = 512;
template <bool PE_INTERNAL, int LAYOUT, bool SAVE, bool SDF_ONLY = false>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(FWD_LDS_BLOCK, 1) void decoder_fwd_lds_kernel(const float* __restrict__ packed,
                                                                        const float* __restrict__ feat,
                                                                        const float* __restrict__ x,
                                                                        const float* __restrict__ embed_pos,
                                                                        float* __restrict__ out,
                                                                        float* __restrict__ saved, uint32_t M,
                                                                        int pin, uint32_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float4 wimg[];
    {
        const float4* src = reinterpret_cast<const float4*>(packed + OFF_F1);
        for (int q = threadIdx.x; q < FWD_LDS_FLOATS / 4; q += FWD_LDS_BLOCK) wimg[q] = src[q];
        const float4* tsrc = reinterpret_cast<const float4*>(packed + OFF_TRGB);
        for (int q = threadIdx.x; q < TAIL_F4; q += FWD_LDS_BLOCK) wimg[FWD_LDS_FLOATS / 4 + q] = tsrc[q];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (uint32_t tile = blockIdx.x * (FWD_LDS_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); tile < n_tiles;
         tile += gridDim.x * (FWD_LDS_BLOCK / 64)) {
        // the images are loop invariant: an opaque zero keeps the compiler from hoisting ~550 LDS reads out of the loop
        // (same for the small per-lane tables read from `packed`)
        uint32_t z = 0, zs = 0;
        asm volatile("" : "+v"(z));
        asm volatile("" : "+s"(zs));
        const float4* w4 = wimg + z;
        (void)zs;
        decoder_fwd_tile<PE_INTERNAL, LAYOUT, SAVE, SDF_ONLY>(reinterpret_cast<const float*>(w4 + FWD_LDS_FLOATS / 4),
                                                    w4 + (OFF_F1 - OFF_F1) / 4, w4 + (OFF_F2 - OFF_F1) / 4,
                                                    w4 + (OFF_F3 - OFF_F1) / 4, feat, x, embed_pos, out, saved, M, pin,
                                                    (int64_t)tile, lane);
    }
}

// ============================================================================ backward chain
// dsmall[s*8 + {0..4}] = d logits, {5..7} = d rgb (inputs of the two small weight-gradient GEMMs)
__device__ __forceinline__ void zero_acc3(f32x16 (&a)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[q][r] = 0.0f;
}
__device__ __forceinline__ void zero_acc4(f32x16 (&a)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[q][r] = 0.0f;
}

// ---------------------------------------------------------------------------- backward chain (LDS-staged)
// Every pre-activation gradient tile goes through a per-wave LDS buffer instead of staying in registers (the first
// version chained accumulators register-to-register like the forward kernel and needed 460 registers): the B
// operand of k-step group t4 is ONE ds_read_b128 (rows 8*t4 + 4h + {0..3} of this lane's sample), which is also
// exactly the 16-byte piece the weight-gradient kernel wants in `dact`.  196 registers -> two blocks share a CU and
// one wave's scalar sections (softmax backward, frequency chain, HBM latency at tile start) hide under the other
// wave's matrix work.  Measured on MI355X (4096x64 samples): 297 us register-chained -> 251 us.
constexpr int TAB_F4 = (OFF_BIAS - OFF_TRGB) / 4;      // rgb-head and sdf2-head tables, contiguous in `packed`
constexpr int XB_ENTRIES = 32 * 32;                  // float4 entries per wave: [group 2*t4+h][sample j]

// first k-group of a layer's A image: requested BEFORE the vector section that precedes the layer, so that the
// layer's first MFMA does not start with an L2 round trip
template <int RT, int T>
__device__ __forceinline__ void preload_a(srd_t wsrd, uint32_t img_off, uint32_t lane16, float4 (&a)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = buf_load16(wsrd, lane16, img_off + (rt * (T / 4)) * 1024);
}

template <int RT, int T, typename SideFn>
__device__ __forceinline__ void mfma_layer_b4(srd_t wsrd, uint32_t img_off, uint32_t lane16, f32x16 (&acc)[RT],
                                              const float4* xb, float4 (&a)[RT], SideFn side) {
    constexpr int T4 = T / 4;
    float4 nxt[RT], b, nb;
    b = xb[0];
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
        if (t4 + 1 < T4) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) nxt[rt] = buf_load16(wsrd, lane16, img_off + (rt * T4 + t4 + 1) * 1024);
            nb = xb[(t4 + 1) * 64];
        }
        side(t4, b);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].x, b.x, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].y, b.y, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].z, b.z, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma(a[rt].w, b.w, acc[rt]);
        if (t4 + 1 < T4) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = nxt[rt];
            b = nb;
        }
    }
}

template <bool PE_INTERNAL, int LAYOUT>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(DEC_BLOCK, 2) void decoder_bwd_lds_kernel(
    const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ out,
    const float* __restrict__ dout, const float* __restrict__ saved, float* __restrict__ dfeat,
    float* __restrict__ dx, float* __restrict__ dembed_pos, float* __restrict__ dact, float* __restrict__ dsmall,
    uint32_t M, int pin) {
    __shared__ float4 xb_all[(DEC_BLOCK / 64) * XB_ENTRIES];
    // the two small per-half-wave weight tables (rgb head, sdf2 head: 6 KB) are read 186 x 16 B per lane and tile;
    // from LDS that is a broadcast ds_read_b128 instead of an L2 round trip
    __shared__ float4 tab[TAB_F4];
    for (int q = threadIdx.x; q < TAB_F4; q += DEC_BLOCK) tab[q] = reinterpret_cast<const float4*>(packed + OFF_TRGB)[q];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    // the wave index is made a scalar so that every per-tile base address below is scalar arithmetic: vector
    // instructions are only partly hidden behind the other wave's MFMAs (DESIGN_NOTES.md 4b)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * (DEC_BLOCK / 64) + wv;
    if (tile * 32 >= (int64_t)M) return;
    float4* xb = xb_all + wv * XB_ENTRIES + h * 32 + j;      // piece p of this lane: xb[p * 64]
    const uint32_t s_raw = (uint32_t)(tile * 32 + j);
    const bool live = s_raw < M;
    const uint32_t s = live ? s_raw : M - 1;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const srd_t wsrd = make_srd(packed, PACKED_FLOATS * 4);
    const srd_t da = make_srd(dact + (size_t)tile * ACT_TILE_FLOATS, ACT_TILE_FLOATS * 4);      // ... and gradients

    // The two 40-byte rows of out / dout, the ReLU mask words and the first weight group are requested before anything
    // is computed: this wave has no earlier tile whose matrix work could cover them.
    float2 o2[5], g2[5];
    {
        const float2* o = reinterpret_cast<const float2*>(out + (size_t)s * 10);
        const float2* g = reinterpret_cast<const float2*>(dout + (size_t)s * 10);
#pragma unroll
        for (int c = 0; c < 5; ++c) o2[c] = o[c], g2[c] = g[c];
    }
    __builtin_amdgcn_sched_barrier(0);      // vmcnt retires in order: the softmax below waits for these rows only
    float4 a3[RT_B3];
    const uint2* mk = reinterpret_cast<const uint2*>(saved + (((size_t)M + 127) / 128) * 4 * ACT_TILE_FLOATS) +
                      (size_t)tile * (MASK_TILE_WORDS / 2) + lane;
    const uint2 mk1 = mk[0], mk3 = mk[64];
    const uint32_t m1[2] = {mk1.x, mk1.y}, m3[2] = {mk3.x, mk3.y};
    preload_a<RT_B3, T_B3>(wsrd, OFF_B3 * 4, lane16, a3);
    __builtin_amdgcn_sched_barrier(0);

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
        if (h == 0) {
            float4* d4 = reinterpret_cast<float4*>(dsmall + (size_t)(tile * 32 + j) * 8);
            d4[0] = make_float4(dlg[0], dlg[1], dlg[2], dlg[3]);
            d4[1] = make_float4(dlg[4], drgb[0], drgb[1], drgb[2]);
        }
    }

    // ---- dG3 = relu'(H3) * (Ws2^T dlogits)   (vector ALU, K = 5), one 16-byte piece at a time into LDS
    {
        const float4* ts2 = tab + (OFF_TS2 - OFF_TRGB) / 4 + h * 128;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int slot = 4 * p + i;
                const float4 w0 = ts2[2 * slot], w1 = ts2[2 * slot + 1];
                float t = w0.x * dlg[0];
                t = fmaf(w0.y, dlg[1], t), t = fmaf(w0.z, dlg[2], t), t = fmaf(w0.w, dlg[3], t), t = fmaf(w1.x, dlg[4], t);
                v[i] = t;
            }
            xb[p * 64] = make_float4(mask_apply(m3, p >> 2, 4 * (p & 3) + 0, v[0]), mask_apply(m3, p >> 2, 4 * (p & 3) + 1, v[1]),
                                     mask_apply(m3, p >> 2, 4 * (p & 3) + 2, v[2]), mask_apply(m3, p >> 2, 4 * (p & 3) + 3, v[3]));
        }
    }

    // ---- d[sdf_emb | grid] = Ws1^T dG3   (3 row tiles: 0,1 -> d sdf_emb, 2 -> d grid features)
    f32x16 dIn3[3];
    zero_acc3(dIn3);
    mfma_layer_b4<RT_B3, T_B3>(wsrd, OFF_B3 * 4, lane16, dIn3, xb, a3,
                               [&](int t4, const float4& b) { if (pin == 0) buf_store16(da, lane16, (2 * 16 + t4) * 1024, b); });
    if (live) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int row = rowmap(r, h);
            const float2 v = make_float2(dIn3[2][r], dIn3[2][r + 1]);
            if (LAYOUT == MIPSF_FEAT_AOS)
                *reinterpret_cast<float2*>(dfeat + (size_t)s * N_GRID + row) = v;
            else
                *reinterpret_cast<float2*>(dfeat + ((size_t)(row >> 1) * M + s) * 2) = v;
        }
    }

    // ---- dH2 = [d sdf_emb (from above) | d rgb_emb = Wrgb^T drgb] -> LDS
    float4 a2[RT_B2];
    preload_a<RT_B2, T_B2>(wsrd, OFF_B2 * 4, lane16, a2);
    const float4* trgb = tab + h * TRGB_SLOTS;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int rt = p >> 2, g = p & 3;
        xb[p * 64] = make_float4(dIn3[rt][4 * g], dIn3[rt][4 * g + 1], dIn3[rt][4 * g + 2], dIn3[rt][4 * g + 3]);
    }
#pragma unroll
    for (int p = 8; p < 16; ++p) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 wv = trgb[4 * (p - 8) + i];
            v[i] = fmaf(wv.z, drgb[2], fmaf(wv.y, drgb[1], wv.x * drgb[0]));
        }
        xb[p * 64] = make_float4(v[0], v[1], v[2], v[3]);
    }

    // ---- dG1 = relu'(H1) * (W2^T dH2)   (dH2 goes out, H1 comes in, one 16-byte group per k-group)
    float4 a1[RT_B1];
    {
        f32x16 dG1[4];
        zero_acc4(dG1);
        mfma_layer_b4<RT_B2, T_B2>(wsrd, OFF_B2 * 4, lane16, dG1, xb, a2,
                                   [&](int t4, const float4& b) {
                                       if (pin == 0) buf_store16(da, lane16, (1 * 16 + t4) * 1024, b);
                                   });
        preload_a<RT_B1, T_B1>(wsrd, OFF_B1 * 4, lane16, a1);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int rt = p >> 2, g = p & 3;
            xb[p * 64] = make_float4(mask_apply(m1, rt, 4 * g + 0, dG1[rt][4 * g]), mask_apply(m1, rt, 4 * g + 1, dG1[rt][4 * g + 1]),
                                     mask_apply(m1, rt, 4 * g + 2, dG1[rt][4 * g + 2]),
                                     mask_apply(m1, rt, 4 * g + 3, dG1[rt][4 * g + 3]));
        }
    }

    // ---- d e = W1^T dG1 (+ rgb share); rows are arranged so that e-slot (t, h) lands in THIS lane
    f32x16 dE[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dE[rt][r] = 0.0f;
    mfma_layer_b4<RT_B1, T_B1>(wsrd, OFF_B1 * 4, lane16, dE, xb, a1,
                               [&](int t4, const float4& b) { if (pin == 0) buf_store16(da, lane16, (0 * 16 + t4) * 1024, b); });
    float de[E_SLOTS];
#pragma unroll
    for (int t = 0; t < E_SLOTS; ++t) {
        const float4 wv = trgb[32 + t];
        de[t] = fmaf(wv.z, drgb[2], fmaf(wv.y, drgb[1], wv.x * drgb[0])) + dE[t >> 4][t & 15];
    }

    if (PE_INTERNAL) {
        const float x0 = x[3 * (size_t)s], x1 = x[3 * (size_t)s + 1], x2 = x[3 * (size_t)s + 2];
        float gx[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float xd = d == 0 ? x0 : (d == 1 ? x1 : x2);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // derivative at the argument the forward used; cos_reduced = 6 instructions instead of cosf's ~65
                // (24 of them per lane were a quarter of this kernel's vector instructions)
                const float arg = fmaf(ldexpf(xd, k), PI_F, h ? HALF_PI_F : 0.0f);
                gx[d] = gx[d] + de[d * 8 + k] * ((ldexpf(1.0f, k) * PI_F) * cos_reduced(arg));
            }
        }
        gx[0] += h == 0 ? de[24] : 0.0f;   // slot 24 carries x0 (lower half) / x1 (upper half)
        gx[1] += h == 1 ? de[24] : 0.0f;
        gx[2] += h == 0 ? de[25] : 0.0f;   // slot 25 carries x2 (lower half only)
#pragma unroll
        for (int d = 0; d < 3; ++d) gx[d] = gx[d] + __shfl_xor(gx[d], 32, 64);
        if (live && h == 0) {
            dx[3 * (size_t)s] = gx[0], dx[3 * (size_t)s + 1] = gx[1], dx[3 * (size_t)s + 2] = gx[2];
        }
    } else if (live) {
#pragma unroll
        for (int t = 0; t < 24; ++t) dembed_pos[(size_t)s * N_PE + (t >> 3) * 16 + 2 * (t & 7) + h] = de[t];
        dx[3 * (size_t)s + h] = de[24];
        if (h == 0) dx[3 * (size_t)s + 2] = de[25];
    }
}

// ============================================================================ weight gradients
constexpr int WG_LDW = 132;                         // row stride: 16-byte aligned rows, 4 banks apart (b128 reads conflict-free)
constexpr int WG_ROWS = 128;
constexpr int WG_LDS_FLOATS = (2 * WG_ROWS + 32) * WG_LDW;   // X^T, Y^T and the 8 (padded to 32) small-gradient rows
constexpr int WG_LDS_BYTES = WG_LDS_FLOATS * 4;

// Staging is split in two so that HBM latency hides under matrix work (the kernel runs one wave per SIMD, nothing
// else covers a stall): the global loads of the NEXT phase are issued into registers from inside the MFMA loop of the
// current phase; put_act writes them, transposed, into LDS after the loop's closing barrier.
// act_base = wave-uniform pointer to the tile's (matrix, first row tile) + lane, so that a 16-byte piece costs one
// load with an immediate (or scalar) offset and no vector address arithmetic
__device__ __forceinline__ const char* act_base(const float* __restrict__ src, int tile_u, int mat, int rt0) {
    return reinterpret_cast<const char*>(src) + ((size_t)tile_u * (ACT_SLOTS / 4) + mat * 16 + rt0 * 4) * 1024;
}
// piece k of a staged matrix: scalar base + 32-bit lane offset + immediate
__device__ __forceinline__ float4 act_piece(const char* base_u, int k, uint32_t lane16) {
    return *reinterpret_cast<const float4*>(base_u + (size_t)(k * 1024) + lane16);
}

// accumulator-image registers -> LDS transposed [feature row][sample]; regs 4g..4g+3 of a tile are rows +0..+3
__device__ __forceinline__ void put_act(float* __restrict__ dstT, const float4* __restrict__ regs, int rt0, int nrt,
                                        int row_shift, int w, int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < nrt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = regs[q * 4 + g];
                float* d = dstT + (feat_of(rt0 + q, 4 * g, h) + row_shift) * WG_LDW + 32 * w + j;
                d[0] = v.x, d[WG_LDW] = v.y, d[2 * WG_LDW] = v.z, d[3 * WG_LDW] = v.w;
            }
        }
    }
}

// acc[ct] += X^T[row tile rtile] * Y^T[col tile ct0+ct]^T over the block's 128 samples
struct NoFetch {
    __device__ __forceinline__ void operator()(int) const {}
};

// acc[ct] += X^T[row tile rtile] * Y^T[col tile ct0+ct]^T over the block's 128 samples
template <int NCT, typename SideFn = NoFetch>
__device__ __forceinline__ void wgrad_mma(const float* __restrict__ XT, const float* __restrict__ YT, int rtile,
                                          int ct0, int lane, f32x16 (&acc)[NCT], SideFn side = NoFetch()) {
    // MFMA k-step t multiplies samples t (lanes 0..31) and t + 64 (lanes 32..63) -- any pairing is valid as long as
    // both operands use it -- so that four consecutive k-steps of one operand are ONE ds_read_b128 (256 operand reads per
    // block tile instead of 512 ds_read2_b32; measured neutral, kept for the simpler loop)
    const int i = lane & 31, kk = lane >> 5;
    const float4* xa = reinterpret_cast<const float4*>(XT + (32 * rtile + i) * WG_LDW + 64 * kk);
    const float4* yb = reinterpret_cast<const float4*>(YT + (32 * ct0 + i) * WG_LDW + 64 * kk);
    // operands of k-step group g+1 are read from LDS while the 4*NCT MFMAs of group g execute; side(g), g = 0..15,
    // issues the g-th slice of the NEXT phase's global loads: spread over the loop, the wave never sits in a full
    // memory-instruction queue before its first MFMA (308 -> 297 us)
    float4 a, b[NCT], na, nb[NCT];
    a = xa[0];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) b[ct] = yb[ct * 8 * WG_LDW];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + 1 < 16) {
            na = xa[g + 1];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) nb[ct] = yb[ct * 8 * WG_LDW + g + 1];
        }
        side(g);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma(a.x, b[ct].x, acc[ct]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma(a.y, b[ct].y, acc[ct]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma(a.z, b[ct].z, acc[ct]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma(a.w, b[ct].w, acc[ct]);
        a = na;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) b[ct] = nb[ct];
        // without the fence the scheduler gathers all sixteen slices at the end of the loop
        if (!std::is_same<SideFn, NoFetch>::value) __builtin_amdgcn_sched_barrier(0);
    }
}

// The same product on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: 16x the fp32-MFMA rate).  Every fp32 operand read
// from LDS is split into hi = bf16(v) and lo = bf16(v - hi) (16-17 significant bits, fp32's exponent range: the
// gradients of a mean over N*S samples sit at 1e-8 .. 1e-3 and need no scaling, unlike f16) and a product is
// hi*hi + hi*lo + lo*hi accumulated in fp32.  Per-term error <= 2^-16, zero-mean; a weight gradient is a sum over all
// 262 144 samples, so the error of the SUM is ~1e-7 of its magnitude (tests: 3e-6 of max against the fp32-MFMA
// kernel).  k-step u of the 8 multiplies samples 8u..8u+7 (lower half-wave) and 64+8u..64+8u+7 (upper): two
// ds_read_b128 per operand -- the same LDS reads as the fp32 form.  side(g), g = 0..15, as above (two per k-step).
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8_bf16(const float4& p, const float4& q, bf8& hi, bf8& lo) {
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 t = (__bf16)v[i];
        hi[i] = t;
        lo[i] = (__bf16)(v[i] - (float)t);
    }
}
template <int NCT, typename SideFn = NoFetch>
__device__ __forceinline__ void wgrad_mma_bf16(const float* __restrict__ XT, const float* __restrict__ YT, int rtile,
                                               int ct0, int lane, f32x16 (&acc)[NCT], SideFn side = NoFetch()) {
    const int i = lane & 31, kk = lane >> 5;
    const float4* xa = reinterpret_cast<const float4*>(XT + (32 * rtile + i) * WG_LDW + 64 * kk);
    const float4* yb = reinterpret_cast<const float4*>(YT + (32 * ct0 + i) * WG_LDW + 64 * kk);
    float4 a0 = xa[0], a1 = xa[1], b0[NCT], b1[NCT], na0, na1, nb0[NCT], nb1[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) b0[ct] = yb[ct * 8 * WG_LDW], b1[ct] = yb[ct * 8 * WG_LDW + 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u + 1 < 8) {
            na0 = xa[2 * u + 2], na1 = xa[2 * u + 3];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) nb0[ct] = yb[ct * 8 * WG_LDW + 2 * u + 2], nb1[ct] = yb[ct * 8 * WG_LDW + 2 * u + 3];
        }
        side(2 * u);
        bf8 ah, al;
        split8_bf16(a0, a1, ah, al);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            bf8 bh, bl;
            split8_bf16(b0[ct], b1[ct], bh, bl);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ct], 0, 0, 0);
        }
        side(2 * u + 1);
        a0 = na0, a1 = na1;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) b0[ct] = nb0[ct], b1[ct] = nb1[ct];
        if (!std::is_same<SideFn, NoFetch>::value) __builtin_amdgcn_sched_barrier(0);
    }
}

// The two products whose X operand is the 8 small-gradient rows (rgb0: drgb, sdf2: dlogits) on 16x16x4 tiles: a
// 32x32x2 tile spends 32 output rows on them, a 16x16x4 one 16 -- half the matrix time of these two phases (4096 ->
// 2048 cycles per wave and phase, 9 % of the kernel's MFMA cycles).  acc[ct] += X8[rows 0..15] * Y^T[cols 16*(2w+ct)
// .. +15]^T; MFMA k-step t multiplies samples {t, t+32, t+64, t+96} (lane groups of 16), so four consecutive k-steps
// of an operand are again one ds_read_b128.  side(g), g = 0..15, as in wgrad_mma.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename SideFn = NoFetch>
__device__ __forceinline__ void wgrad_mma_small(const float* __restrict__ X8, const float* __restrict__ YT, int w,
                                                int lane, f32x4 (&acc)[2], SideFn side = NoFetch()) {
    const int i = lane & 15, g4 = lane >> 4;
    const float4* xa = reinterpret_cast<const float4*>(X8 + i * WG_LDW + 32 * g4);
    const float4* yb = reinterpret_cast<const float4*>(YT + (32 * w + i) * WG_LDW + 32 * g4);
    float4 a = xa[0], b0 = yb[0], b1 = yb[4 * WG_LDW], na, nb0, nb1;
#pragma unroll
    for (int t4 = 0; t4 < 8; ++t4) {
        if (t4 + 1 < 8) na = xa[t4 + 1], nb0 = yb[t4 + 1], nb1 = yb[4 * WG_LDW + t4 + 1];
        side(2 * t4);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[1], 0, 0, 0);
        side(2 * t4 + 1);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc[1], 0, 0, 0);
        a = na, b0 = nb0, b1 = nb1;
        if (!std::is_same<SideFn, NoFetch>::value) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NCT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NCT]) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
}

// write one accumulator tile into the block's natural-layout partial record
__device__ __forceinline__ void flush_tile(float* __restrict__ rec, int base, int out_dim, int in_dim, int rtile,
                                           int ctile, int lane, const f32x16& acc) {
    const int jj = lane & 31, hh = lane >> 5;
    const int col = 32 * ctile + jj;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * rtile + rowmap(r, hh);
        if (row < out_dim && col < in_dim) rec[base + row * in_dim + col] = acc[r];
    }
}

template <bool PE_INTERNAL, int LAYOUT, bool BF16X3 = false>
__global__ __launch_bounds__(DEC_BLOCK, 1) void decoder_wgrad_kernel(
    const float* __restrict__ feat, const float* __restrict__ x, const float* __restrict__ embed_pos,
    const float* __restrict__ saved, const float* __restrict__ dact, const float* __restrict__ dsmall,
    float* __restrict__ partial, uint32_t M, uint32_t n_btiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* XT = lds;
    float* YT = lds + WG_ROWS * WG_LDW;
    float* X8 = lds + 2 * WG_ROWS * WG_LDW;       // [dlogits(5) | drgb(3)] rows; rows 8..31 are zero
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    for (int q = tid; q < 24 * WG_LDW; q += DEC_BLOCK) X8[8 * WG_LDW + q] = 0.0f;      // rows 8..31 stay zero
    f32x16 aS1[3], aW2[4], aW1[2];
    f32x4 aS2[2], aRGB[2];              // 16x16 tiles: rows 0..7 = the 8 small-gradient rows, columns 32w + 16ct + (lane & 15)
    zero_acc(aS1), zero_acc(aW2), zero_acc(aW1);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) aS2[ct][r] = 0.0f, aRGB[ct][r] = 0.0f;
    float db3 = 0.f, db2 = 0.f;
    const int brow = tid & 127, bhalf = tid >> 7;

    auto row_sum = [&](const float* T, int row, int c0, int n) {
        // four 16-byte reads in flight at a time (fully unrolled, the loads of a row sum are all hoisted in front of the
        // adds and their registers, on top of the 128 prefetch registers, push loop invariants into scratch -- and
        // every scratch reload waits on vmcnt, i.e. for the whole prefetch it was meant to overlap; DESIGN_NOTES.md 4b)
        const float4* T4 = reinterpret_cast<const float4*>(T + row * WG_LDW + c0);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 1
        for (int c = 0; c < n / 4; c += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = T4[c + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) s0 += v[u].x + v[u].z, s1 += v[u].y + v[u].w;
        }
        return s0 + s1;
    };

    // prefetch registers: X / Y operands of the next phase (+ grid features, coordinates, small gradients)
    float4 nx[16], ny[16], nsm;
    float ngf[16];
    // Everything that selects a tile is kept in scalar registers (w is wave-uniform): with one wave per SIMD a
    // vector instruction is never hidden behind this wave's own MFMAs (DESIGN_NOTES.md 4b), so per-load address
    // arithmetic, per-load liveness selects and zero-fills were costing as much as the matrix work they fed.
    // Loads are unconditional from a tile that exists; a dead wave tile (tail of M) zeroes its X operand instead.
    const int n_wtiles = (int)((M + 31u) / 32u);
    const uint32_t lane16 = 16u * (uint32_t)lane;
    // reversed tile order: the tiles the chain kernel wrote last are still in the 256 MB Infinity Cache (318 -> 310 us)
    auto tile_of = [&](uint32_t bt) { return (int)(n_btiles - 1u - (bt < n_btiles ? bt : n_btiles - 1u)) * 4 + w; };
    auto live_of = [&](uint32_t bt) { return bt < n_btiles && tile_of(bt) < n_wtiles; };
    auto src_tile = [&](uint32_t bt) { const int t = tile_of(bt); return t < n_wtiles ? t : n_wtiles - 1; };
    auto sample_of = [&](uint32_t bt) {
        const uint32_t s_raw = (uint32_t)(src_tile(bt) * 32 + j);
        return s_raw < M ? s_raw : M - 1;
    };
    auto zero16 = [&](float4* r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    };

    {   // first block tile's first phase: X = dG3, Y = [sdf_emb | grid]
        const int t0 = src_tile(blockIdx.x);
        const char* bx = act_base(dact, t0, 2, 0);
        const char* by = act_base(saved, t0, 1, 0);
        const uint32_t s0 = sample_of(blockIdx.x);
#pragma unroll
        for (int k = 0; k < 16; ++k) nx[k] = act_piece(bx, k, lane16);
#pragma unroll
        for (int k = 0; k < 8; ++k) ny[k] = act_piece(by, k, lane16);
#pragma unroll
        for (int u = 0; u < 16; ++u) ngf[u] = load_feat<LAYOUT>(feat, s0, u, h, M);
    }
    for (uint32_t bt = blockIdx.x; bt < n_btiles; bt += gridDim.x) {
        const bool tile_live = live_of(bt);
        const int tile = src_tile(bt);
        const uint32_t s = sample_of(bt);

        // ---------------- phase sdf0: X = dG3, Y = [sdf_emb | grid]
        if (!tile_live) zero16(nx);
        put_act(XT, nx, 0, 4, 0, w, lane);
        put_act(YT, ny, 0, 2, 0, w, lane);
#pragma unroll
        for (int u = 0; u < 16; ++u) YT[(64 + 2 * u + h) * WG_LDW + 32 * w + j] = ngf[u];
        __syncthreads();
        {                                                                        // next: X = dH2, Y = H1
            const char* bx = act_base(dact, tile, 1, 0);
            const char* by = act_base(saved, tile, 0, 0);
            auto fetch = [&](int g) {
                nx[g] = act_piece(bx, g, lane16);
                ny[g] = act_piece(by, g, lane16);
            };
            if constexpr (BF16X3) wgrad_mma_bf16<3>(XT, YT, w, 0, lane, aS1, fetch);
            else wgrad_mma<3>(XT, YT, w, 0, lane, aS1, fetch);
        }
        db3 += row_sum(XT, brow, 64 * bhalf, 64);
        __syncthreads();

        // ---------------- phase pts2: X = dH2, Y = H1
        if (!tile_live) zero16(nx);
        put_act(XT, nx, 0, 4, 0, w, lane);
        put_act(YT, ny, 0, 4, 0, w, lane);
        __syncthreads();
        {                                                    // next: X = dG1 (+ the 8 small rows), Y = [e | rgb_emb]
            const char* bx = act_base(dact, tile, 0, 0);
            const char* by = act_base(saved, tile, 1, 2);
            auto fetch = [&](int g) {
                nx[g] = act_piece(bx, g, lane16);
                if (g < 8) ny[g] = act_piece(by, g, lane16);
                if (g == 8) {
                    const int64_t sg = (int64_t)(tile_of(bt) - w) * 32 + brow;
                    nsm = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (sg < (int64_t)M) nsm = reinterpret_cast<const float4*>(dsmall + sg * 8)[bhalf];
                }
            };
            if constexpr (BF16X3) wgrad_mma_bf16<4>(XT, YT, w, 0, lane, aW2, fetch);
            else wgrad_mma<4>(XT, YT, w, 0, lane, aW2, fetch);
        }
        db2 += row_sum(XT, brow, 64 * bhalf, 64);
        __syncthreads();

        // ---------------- phase pts0 + rgb0: X = dG1 and the 8 small rows,
        // Y = [e (rows 0..50) | ONES (row 51) | zero (52..63) | rgb_emb (64..127)]
        // (d w_rgb0 = drgb^T [rgb_emb | e] rides on the staging of e: one phase, two barriers and one e-staging less;
        //  the ones row makes column 51 of both products the bias gradients: sums over samples of dG1 / dlogits / drgb)
        if (!tile_live) zero16(nx);
        put_act(XT, nx, 0, 4, 0, w, lane);
        put_act(YT, ny, 2, 2, 0, w, lane);
        {
            float ev[E_SLOTS];
            load_e<PE_INTERNAL>(x, embed_pos, s, h, ev);
#pragma unroll
            for (int t = 0; t < E_SLOTS; ++t) {
                const int e = eidx(t, h);
                if (e >= 0) YT[e * WG_LDW + 32 * w + j] = ev[t];
            }
        }
        {
            float* pad = YT + (51 + bhalf) * WG_LDW + brow;          // rows 51..63, two rows per pass
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (bhalf + 2 * i < 13) pad[2 * i * WG_LDW] = (bhalf + 2 * i == 0) ? 1.0f : 0.0f;
        }
        X8[(4 * bhalf + 0) * WG_LDW + brow] = nsm.x;
        X8[(4 * bhalf + 1) * WG_LDW + brow] = nsm.y;
        X8[(4 * bhalf + 2) * WG_LDW + brow] = nsm.z;
        X8[(4 * bhalf + 3) * WG_LDW + brow] = nsm.w;
        __syncthreads();
        {                                                                        // next: Y = H3
            const char* by = act_base(saved, tile, 2, 0);
            auto fetch = [&](int g) { ny[g] = act_piece(by, g, lane16); };
            if constexpr (BF16X3) wgrad_mma_bf16<2>(XT, YT, w, 0, lane, aW1, fetch);
            else wgrad_mma<2>(XT, YT, w, 0, lane, aW1, fetch);
        }
        wgrad_mma_small(X8, YT, w, lane, aRGB);
        __syncthreads();

        // ---------------- phase sdf2: X = the 8 small rows (still in X8), Y = H3
        put_act(YT, ny, 0, 4, 0, w, lane);
        __syncthreads();
        {                                                                        // next block tile's first phase
            const uint32_t nb = bt + gridDim.x;
            const int nt = src_tile(nb);
            const char* bx = act_base(dact, nt, 2, 0);
            const char* by = act_base(saved, nt, 1, 0);
            const uint32_t ns = sample_of(nb);
            wgrad_mma_small(X8, YT, w, lane, aS2, [&](int g) {
                nx[g] = act_piece(bx, g, lane16);
                if (g < 8) ny[g] = act_piece(by, g, lane16);
                ngf[g] = load_feat<LAYOUT>(feat, ns, g, h, M);
            });
        }
        __syncthreads();
    }

    // ---------------- flush this block's partial record
    float* rec = partial + (size_t)blockIdx.x * (G_STRIDE + 0);
    for (int q = tid; q < G_STRIDE; q += DEC_BLOCK) rec[q] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) flush_tile(rec, G_W_SDF0, HID, N_SDF_IN, w, ct, lane, aS1[ct]);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) flush_tile(rec, G_W_PTS2, HID, HID, w, ct, lane, aW2[ct]);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) flush_tile(rec, G_W_PTS0, HID, N_E, w, ct, lane, aW1[ct]);
    {
        // 16x16 accumulator: register r of lane l = row 4*(l >> 4) + r, column l & 15 of its tile.
        // rows 0..4 of aS2 = d w_sdf2[c][col]; rows 5..7 of aRGB = d w_rgb0[c][..]; column 51 of aRGB (ones row of the
        // pts0 phase's Y) = d b_sdf2 (rows 0..4) and d b_rgb0 (rows 5..7)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r, col = 32 * w + 16 * ct + (lane & 15);
                if (row < N_CLASS) rec[G_W_SDF2 + row * HID + col] = aS2[ct][r];
                // aRGB columns follow the phase's Y rows: [e (0..50) | ones | pad | rgb_emb (64..127)] -> w_rgb0 columns [64 + e | rgb_emb]
                const int rc = col < N_E ? 64 + col : (col >= 64 ? col - 64 : -1);
                if (row >= 5 && row < 8 && rc >= 0) rec[G_W_RGB0 + (row - 5) * N_RGB_IN + rc] = aRGB[ct][r];
                if (col == N_E && row < N_CLASS) rec[G_B_SDF2 + row] = aRGB[ct][r];
                if (col == N_E && row >= 5 && row < 8) rec[G_B_RGB0 + row - 5] = aRGB[ct][r];
            }
    }
    {
        // column 51 of the pts0 product is the ones row: d b_pts0 (rows of aW1[1])
        const int jj = lane & 31, hh = lane >> 5;
        if (jj == N_E - 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rec[G_B_PTS0 + 32 * w + rowmap(r, hh)] = aW1[1][r];
        }
    }
    // remaining bias partials: two column halves per row -> atomics inside the block's own record
    atomicAdd(&rec[G_B_SDF0 + brow], db3);
    atomicAdd(&rec[G_B_PTS2 + brow], db2);
}


struct GradPtrs {
    float* p[10];   // order of the G_* record: w_pts0 b_pts0 w_pts2 b_pts2 w_rgb0 b_rgb0 w_sdf0 b_sdf0 w_sdf2 b_sdf2
};

constexpr int WG_REDUCE_SLICES = 8;
__global__ __launch_bounds__(256) void decoder_wgrad_reduce_kernel(const float* __restrict__ partial, uint32_t nrec,
                                                                   GradPtrs g) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= G_TOTAL) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;          // independent chains keep 4 loads in flight
    uint32_t b = blockIdx.y;
    for (; b + 3 * WG_REDUCE_SLICES < nrec; b += 4 * WG_REDUCE_SLICES) {
        a0 += partial[(size_t)b * G_STRIDE + q];
        a1 += partial[(size_t)(b + WG_REDUCE_SLICES) * G_STRIDE + q];
        a2 += partial[(size_t)(b + 2 * WG_REDUCE_SLICES) * G_STRIDE + q];
        a3 += partial[(size_t)(b + 3 * WG_REDUCE_SLICES) * G_STRIDE + q];
    }
    for (; b < nrec; b += WG_REDUCE_SLICES) a0 += partial[(size_t)b * G_STRIDE + q];
    const float a = (a0 + a1) + (a2 + a3);
    const int bounds[11] = {G_W_PTS0, G_B_PTS0, G_W_PTS2, G_B_PTS2, G_W_RGB0, G_B_RGB0,
                            G_W_SDF0, G_B_SDF0, G_W_SDF2, G_B_SDF2, G_TOTAL};
#pragma unroll
    for (int k = 0; k < 10; ++k)
        if (q >= bounds[k] && q < bounds[k + 1]) unsafeAtomicAdd(&g.p[k][q - bounds[k]], a);
}

__global__ __launch_bounds__(256) void decoder_pack_kernel(W w, float* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < PACKED_FLOATS) packed[idx] = packed_value(w, idx);
}

static W to_w(const mipsf_decoder_weights& s) {
    W w;
    w.w_pts0 = s.w_pts0, w.b_pts0 = s.b_pts0, w.w_pts2 = s.w_pts2, w.b_pts2 = s.b_pts2, w.w_rgb0 = s.w_rgb0;
    w.b_rgb0 = s.b_rgb0, w.w_sdf0 = s.w_sdf0, w.b_sdf0 = s.b_sdf0, w.w_sdf2 = s.w_sdf2, w.b_sdf2 = s.b_sdf2;
    return w;
}

constexpr uint32_t WG_MAX_BLOCKS = 256;

// sum of `nrec` per-block partial records (G_* layout) into the ten gradient tensors; shared with wgrad16.hip
int wgrad_reduce_launch(const float* partial, uint32_t nrec, const mipsf_decoder_grads* grads, hipStream_t s) {
    GradPtrs g;
    g.p[0] = grads->w_pts0, g.p[1] = grads->b_pts0, g.p[2] = grads->w_pts2, g.p[3] = grads->b_pts2;
    g.p[4] = grads->w_rgb0, g.p[5] = grads->b_rgb0, g.p[6] = grads->w_sdf0, g.p[7] = grads->b_sdf0;
    g.p[8] = grads->w_sdf2, g.p[9] = grads->b_sdf2;
    for (int k = 0; k < 10; ++k) MIPSF_REQUIRE(g.p[k] != nullptr, "null gradient pointer %d", k);
    hipLaunchKernelGGL(decoder_wgrad_reduce_kernel, dim3((G_TOTAL + 255) / 256, WG_REDUCE_SLICES), dim3(256), 0, s, partial,
                       nrec, g);
    return check_launch("decoder_wgrad_reduce");
}

static inline uint64_t n_wave_tiles(uint32_t M) { return ((uint64_t)M + 31) / 32; }
static inline uint64_t n_block_tiles(uint32_t M) { return ((uint64_t)M + 127) / 128; }

// buffer sizes (mipsf_buffer_size, capi.hip)
uint64_t decoder_packed_floats() { return (uint64_t)PACKED_FLOATS; }
// saved / dact are addressed per 128-sample block tile by the weight-gradient kernel -> round up to 4 wave tiles
uint64_t decoder_saved_floats(uint32_t M) { return n_block_tiles(M) * 4 * (ACT_TILE_FLOATS + MASK_TILE_WORDS); }
uint64_t decoder_dact_floats(uint32_t M) { return n_block_tiles(M) * 4 * ACT_TILE_FLOATS + n_block_tiles(M) * 128 * 8; }
uint64_t decoder_wgrad_partial_floats() { return (uint64_t)WG_MAX_BLOCKS * G_STRIDE; }

}  // namespace mipsf

using namespace mipsf;

extern "C" {

int mipsf_decoder_pack(const mipsf_decoder_weights* w, float* packed, void* stream) {
    MIPSF_REQUIRE(w && packed, "null pointer");
    hipLaunchKernelGGL(decoder_pack_kernel, dim3((PACKED_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       to_w(*w), packed);
    return check_launch("decoder_pack");
}

int mipsf_decoder_pack_host(const mipsf_decoder_weights* w, float* packed_host) {
    MIPSF_REQUIRE(w && packed_host, "null pointer");
    const W ww = to_w(*w);
    for (int idx = 0; idx < PACKED_FLOATS; ++idx) packed_host[idx] = packed_value(ww, idx);
    return 0;
}

static int decoder_fwd_launch(const float* packed, const float* feat, int feat_layout, const float* x,
                              const float* embed_pos, int pe_mode, float* out, float* saved, bool sdf_only, uint32_t M,
                              void* stream) {
    if (M == 0) return 0;
    MIPSF_REQUIRE(packed && feat && x && out, "null pointer");
    MIPSF_REQUIRE(pe_mode == 0 || embed_pos, "pe_mode 1 needs embed_pos");
    MIPSF_REQUIRE(feat_layout == MIPSF_FEAT_AOS || feat_layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout");
    const uint32_t n_tiles = (uint32_t)n_wave_tiles(M);
    const uint32_t blocks = (n_tiles + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
    const int cus = device_cus();
    if (cus <= 0) return 3;
    // persistent LDS-resident weights pay off once every CU has several rounds of tiles to amortise the 140 KB fill
    const bool persistent = n_tiles >= (uint32_t)cus * 8u * MIPSF_FWD_LDS_MIN_ROUNDS;
#define FWD(PE, LAY, SV, SDF)                                                                                    \
    do {                                                                                                         \
        if (persistent) {                                                                                        \
            static bool attr_set_dev[MAX_DEVICES] = {false};                                                     \
            bool& attr_set = attr_set_dev[device_slot()];                                                        \
            if (!attr_set) {                                                                                     \
                if (hipFuncSetAttribute((const void*)decoder_fwd_lds_kernel<PE, LAY, SV, SDF>,                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS_BYTES) != hipSuccess) { \
                    set_error("cannot raise dynamic LDS to %d bytes", FWD_LDS_BYTES);                            \
                    return 4;                                                                                    \
                }                                                                                                \
                attr_set = true;                                                                                 \
            }                                                                                                    \
            hipLaunchKernelGGL((decoder_fwd_lds_kernel<PE, LAY, SV, SDF>), dim3(cus), dim3(FWD_LDS_BLOCK), FWD_LDS_BYTES, s, \
                               packed, feat, x, embed_pos, out, saved, M, 0, n_tiles);                           \
        } else {                                                                                                 \
            hipLaunchKernelGGL((decoder_fwd_kernel<PE, LAY, SV, SDF>), dim3(blocks), dim3(DEC_BLOCK), 0, s, packed, feat, x, \
                               embed_pos, out, saved, M, 0);                                                     \
        }                                                                                                        \
    } while (0)
#define FWD_SV(PE, LAY)                                  \
    do {                                                 \
        if (sdf_only) FWD(PE, LAY, false, true);         \
        else if (saved != nullptr) FWD(PE, LAY, true, false); \
        else FWD(PE, LAY, false, false);                 \
    } while (0)
    if (pe_mode == 0) {
        if (feat_layout == MIPSF_FEAT_AOS) FWD_SV(true, MIPSF_FEAT_AOS); else FWD_SV(true, MIPSF_FEAT_LEVEL_MAJOR);
    } else {
        if (feat_layout == MIPSF_FEAT_AOS) FWD_SV(false, MIPSF_FEAT_AOS); else FWD_SV(false, MIPSF_FEAT_LEVEL_MAJOR);
    }
#undef FWD_SV
#undef FWD
    return check_launch("decoder_fwd");
}

int mipsf_decoder_fwd(const float* packed, const float* feat, int feat_layout, const float* x,
                      const float* embed_pos, int pe_mode, float* out, float* saved, uint32_t M, void* stream) {
    return decoder_fwd_launch(packed, feat, feat_layout, x, embed_pos, pe_mode, out, saved, false, M, stream);
}

int mipsf_decoder_fwd_sdf(const float* packed, const float* feat, int feat_layout, const float* x,
                          const float* embed_pos, int pe_mode, float* sdf, uint32_t M, void* stream) {
    return decoder_fwd_launch(packed, feat, feat_layout, x, embed_pos, pe_mode, sdf, nullptr, true, M, stream);
}

int mipsf_decoder_bwd_chain(const float* packed, int feat_layout, const float* x, int pe_mode, const float* out,
                            const float* dout, const float* saved, float* dfeat, float* dx, float* dembed_pos,
                            float* dact, uint32_t M, void* stream) {
    if (M == 0) return 0;
    MIPSF_REQUIRE(packed && x && out && dout && saved && dfeat && dx && dact, "null pointer");
    MIPSF_REQUIRE(pe_mode == 0 || dembed_pos, "pe_mode 1 needs dembed_pos");
    MIPSF_REQUIRE(feat_layout == MIPSF_FEAT_AOS || feat_layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t blocks = (uint32_t)((n_wave_tiles(M) + 3) / 4);
    float* dsmall = dact + n_block_tiles(M) * 4 * ACT_TILE_FLOATS;
#define BWD(PE, LAY) \
    hipLaunchKernelGGL((decoder_bwd_lds_kernel<PE, LAY>), dim3(blocks), dim3(DEC_BLOCK), 0, s, packed, x, out, dout, saved, dfeat, dx, dembed_pos, dact, dsmall, M, MIPSF_PIN_ARG)
    if (pe_mode == 0) { if (feat_layout == MIPSF_FEAT_AOS) BWD(true, MIPSF_FEAT_AOS); else BWD(true, MIPSF_FEAT_LEVEL_MAJOR); }
    else { if (feat_layout == MIPSF_FEAT_AOS) BWD(false, MIPSF_FEAT_AOS); else BWD(false, MIPSF_FEAT_LEVEL_MAJOR); }
#undef BWD
    return check_launch("decoder_bwd_chain");
}

static int decoder_wgrad_launch(const float* feat, int feat_layout, const float* x, const float* embed_pos, int pe_mode,
                               const float* saved, const float* dact, const mipsf_decoder_grads* grads, float* partial,
                               bool bf16x3, uint32_t M, void* stream);
int mipsf_decoder_wgrad(const float* feat, int feat_layout, const float* x, const float* embed_pos, int pe_mode,
                        const float* saved, const float* dact, const mipsf_decoder_grads* grads, float* partial,
                        int precision, uint32_t M, void* stream) {
    MIPSF_REQUIRE(precision == MIPSF_PREC_F32 || precision == MIPSF_PREC_BF16X3, "precision must be f32 or bf16x3");
    return decoder_wgrad_launch(feat, feat_layout, x, embed_pos, pe_mode, saved, dact, grads, partial,
                                precision == MIPSF_PREC_BF16X3, M, stream);
}
static int decoder_wgrad_launch(const float* feat, int feat_layout, const float* x, const float* embed_pos, int pe_mode,
                               const float* saved, const float* dact, const mipsf_decoder_grads* grads, float* partial,
                               bool bf16x3, uint32_t M, void* stream) {
    if (M == 0) return 0;
    MIPSF_REQUIRE(feat && x && saved && dact && partial && grads, "null pointer");
    MIPSF_REQUIRE(pe_mode == 0 || embed_pos, "pe_mode 1 needs embed_pos");
    MIPSF_REQUIRE(feat_layout == MIPSF_FEAT_AOS || feat_layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n_bt = (uint32_t)n_block_tiles(M);
    const float* dsmall = dact + n_block_tiles(M) * 4 * ACT_TILE_FLOATS;
    GradPtrs g;
    g.p[0] = grads->w_pts0, g.p[1] = grads->b_pts0, g.p[2] = grads->w_pts2, g.p[3] = grads->b_pts2;
    g.p[4] = grads->w_rgb0, g.p[5] = grads->b_rgb0, g.p[6] = grads->w_sdf0, g.p[7] = grads->b_sdf0;
    g.p[8] = grads->w_sdf2, g.p[9] = grads->b_sdf2;
    for (int k = 0; k < 10; ++k) MIPSF_REQUIRE(g.p[k] != nullptr, "null gradient pointer %d", k);
    const int cus = device_cus();
    if (cus <= 0) return 3;
    uint32_t wg_blocks = n_bt < (uint32_t)cus ? n_bt : (uint32_t)cus;
    if (wg_blocks > WG_MAX_BLOCKS) wg_blocks = WG_MAX_BLOCKS;
#define WG(PE, LAY, BF)                                                                                          \
    do {                                                                                                         \
        static bool attr_set_dev[MAX_DEVICES] = {false};                                                         \
        bool& attr_set = attr_set_dev[device_slot()];                                                            \
        if (!attr_set) {                                                                                         \
            if (hipFuncSetAttribute((const void*)decoder_wgrad_kernel<PE, LAY, BF>,                              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES) != hipSuccess) {   \
                set_error("cannot raise dynamic LDS to %d bytes", WG_LDS_BYTES);                                 \
                return 4;                                                                                        \
            }                                                                                                    \
            attr_set = true;                                                                                     \
        }                                                                                                        \
        hipLaunchKernelGGL((decoder_wgrad_kernel<PE, LAY, BF>), dim3(wg_blocks), dim3(DEC_BLOCK), WG_LDS_BYTES, s, \
                           feat, x, embed_pos, saved, dact, dsmall, partial, M, n_bt);                           \
    } while (0)
#define WG_P(PE, LAY) do { if (bf16x3) WG(PE, LAY, true); else WG(PE, LAY, false); } while (0)
    if (pe_mode == 0) { if (feat_layout == MIPSF_FEAT_AOS) WG_P(true, MIPSF_FEAT_AOS); else WG_P(true, MIPSF_FEAT_LEVEL_MAJOR); }
    else { if (feat_layout == MIPSF_FEAT_AOS) WG_P(false, MIPSF_FEAT_AOS); else WG_P(false, MIPSF_FEAT_LEVEL_MAJOR); }
#undef WG_P
#undef WG
    if (int e = check_launch("decoder_wgrad")) return e;
    hipLaunchKernelGGL(decoder_wgrad_reduce_kernel, dim3((G_TOTAL + 255) / 256, WG_REDUCE_SLICES), dim3(256), 0, s, partial, wg_blocks, g);
    return check_launch("decoder_wgrad_reduce");
}

int mipsf_decoder_bwd(const float* packed, const float* feat, int feat_layout, const float* x,
                      const float* embed_pos, int pe_mode, const float* out, const float* dout,
                      const float* saved, float* dfeat, float* dx, float* dembed_pos,
                      const mipsf_decoder_grads* grads, float* dact, float* partial, uint32_t M, void* stream) {
    if (int e = mipsf_decoder_bwd_chain(packed, feat_layout, x, pe_mode, out, dout, saved, dfeat, dx, dembed_pos, dact,
                                        M, stream))
        return e;
    return mipsf_decoder_wgrad(feat, feat_layout, x, embed_pos, pe_mode, saved, dact, grads, partial, MIPSF_PREC_F32, M, stream);
}

}  // extern "C"
