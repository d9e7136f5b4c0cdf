/*
 * mipsf.h -- C ABI of libmipsf_hip.so: the MI355X (gfx950) implementation of the MIPS-Fusion
 * render-and-optimise hot path.
 *
 * Every entry point is a plain `extern "C"` function over raw device pointers and sizes.  All
 * pointers are DEVICE pointers (HBM) unless the name ends in `_host`; the caller owns every buffer;
 * kernels are enqueued on the `hipStream_t` passed as `void* stream` and never synchronise the
 * device.  Return value: 0 = ok, non-zero = error (message via mipsf_last_error(), thread local).
 * No exceptions cross this boundary.
 *
 * Reference interfaces replaced (paths under the upstream repository root):
 *   mipsf_hashgrid_*      tinycudann.Encoding(HashGrid)   model/encodings.py:11-26, used model/scene_rep.py:40,122
 *   mipsf_freq_*          tinycudann.Encoding(Frequency)  model/encodings.py:29-39, used model/scene_rep.py:37,123
 *   mipsf_decoder_*       MLP_reg.forward (+autograd)     model/decoder.py:53-75 (layers :32-50)
 *   mipsf_sample_rays     render_rays steps 1-2 + run_network normalisation
 *                                                         model/scene_rep.py:156-179, 134-142
 *   mipsf_render_*        sdf2weights / raw2outputs / forward losses
 *                                                         model/scene_rep.py:58-103, 211-236;
 *                                                         helper_functions/utils.py:21-111
 *   mipsf_rays_bwd        autograd of pts = o + d*z and of the fp64 normalisation
 *   mipsf_pose_rays_*     per-ray pose gather + R*d     mipsfusion.py:320-322, 531-532; geometry_helper.py:11-17
 *   mipsf_adam_step       torch.optim.Adam.step           mipsfusion.py:580-584, 190, 330-335
 *   mipsf_ro_fitness      RandomOptimizer.get_fitness     RandomOptimizer.py:113-131
 *   mipsf_ro_particles/_update  one RandomOptimizer round  RandomOptimizer.py:184-224
 */
#ifndef MIPSF_H
#define MIPSF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPSF_MAX_LEVELS 32
#define MIPSF_ABI_VERSION 1

/* ------------------------------------------------------------------ errors / info */
const char* mipsf_last_error(void);
int mipsf_abi_version(void);
/* number of compute units of the current device (used to size persistent grids); <0 on error */
int mipsf_device_cu_count(void);

/* --------------------------------------------------------------- hash grid (a5) */
typedef struct mipsf_grid_meta {
    uint32_t n_levels;              /* L                                   */
    uint32_t n_features;            /* F, only 2 is built                   */
    uint32_t log2_hashmap_size;     /* T = 2^this                           */
    uint32_t base_resolution;       /* N_min                                */
    float per_level_scale;          /* b as fp32                            */
    float log2_per_level_scale;     /* log2f(b)                             */
    uint32_t n_params;              /* total floats = offsets[L] * F        */
    uint32_t offsets[MIPSF_MAX_LEVELS + 1];   /* entry offset of each level */
    uint32_t resolutions[MIPSF_MAX_LEVELS];
    float scales[MIPSF_MAX_LEVELS];
} mipsf_grid_meta;

/* Host-only: fill the level table exactly as tiny-cuda-nn's GridEncodingTemplated constructor does. */
int mipsf_hashgrid_meta_init(mipsf_grid_meta* meta_host, uint32_t n_levels, uint32_t n_features,
                             uint32_t log2_hashmap_size, uint32_t base_resolution, double per_level_scale);

/* Feature layouts.  AOS: out[i*L*F + level*F + f] (what tcnn.Encoding returns).
 * LEVEL_MAJOR: out[(level*M + i)*F + f] (internal layout between the grid and the decoder). */
#define MIPSF_FEAT_AOS 0
#define MIPSF_FEAT_LEVEL_MAJOR 1

/* x: [M,3] fp32 already normalised (scene_rep.py:140-142 + :119); params: [n_params]; out: [M,L*F]. */
int mipsf_hashgrid_fwd(const float* x, const float* params, float* out, uint32_t M,
                       const mipsf_grid_meta* meta_host, int layout, void* stream);
/* Same, and also stores the Jacobian d out / d x: jac[((level*3 + d)*M + i)*2 + f] = d out_f(level) / d x_d (the
 * quantity tcnn's kernel_grid_backward_input recomputes from the table).  With it the backward obtains dL/dx from a
 * streaming pass (mipsf_hashgrid_dx_from_jac) instead of gathering 8 table entries per level again. */
int mipsf_hashgrid_fwd_jac(const float* x, const float* params, float* out, float* jac, uint32_t M,
                           const mipsf_grid_meta* meta_host, int layout, void* stream);
/* dx [M,3] += sum over levels of jac . dL/dout   (bit-identical to the dx part of mipsf_hashgrid_bwd) */
/* ... over the 32-sample tiles listed in tile_live only (the buffer mipsf_decoder_bwd_chain16_ex filled for the same
 * batch; NULL = every sample): the other samples have a zero feature gradient and add nothing. */
int mipsf_hashgrid_dx_from_jac_tiles(const float* jac, const float* dout, float* dx, const uint32_t* tile_live, uint32_t M,
                                     const mipsf_grid_meta* meta, int layout, void* stream);
int mipsf_hashgrid_dx_from_jac(const float* jac, const float* dout, float* dx, uint32_t M,
                               const mipsf_grid_meta* meta_host, int layout, void* stream);
/* dparams (nullable: frozen grid) += scatter of dL/dout (accumulated on chip in LDS slices, see hashgrid.hip);
 * dx (nullable) += dL/dx [M,3].
 * scratch: mipsf_hashgrid_bwd_scratch_floats(meta, M, dx != NULL) floats owned by the caller. */
uint64_t mipsf_hashgrid_bwd_scratch_floats(const mipsf_grid_meta* meta_host, uint32_t M, int need_dx);
int mipsf_hashgrid_bwd(const float* x, const float* params, const float* dout, float* dparams, float* dx,
                       float* scratch, uint32_t M, const mipsf_grid_meta* meta_host, int layout, void* stream);
/* The same with a small counter block the CALLER keeps between calls (mipsf_hashgrid_counter_words(meta) uint32 words,
 * all zero before the first call; every call leaves it ready for the next one): one launch fewer -- the routed scatter is
 * then three launches (route, accumulate, fold of split bins).  One block per stream: calls that share a block must be
 * ordered. */
uint64_t mipsf_hashgrid_counter_words(const mipsf_grid_meta* meta_host);
int mipsf_hashgrid_bwd_keep(const float* x, const float* params, const float* dout, float* dparams, float* dx,
                            float* scratch, uint32_t* counters, uint32_t M, const mipsf_grid_meta* meta_host, int feat_layout,
                            void* stream);
/* The same with flags.  MIPSF_HG_DPARAMS_ZERO: the caller vouches that dparams is all zero on entry (the gradient buffer of
 * an optimiser that clears it, a fresh allocation): table slices are then stored instead of read-modify-written. */
#define MIPSF_HG_DPARAMS_ZERO 1u
int mipsf_hashgrid_bwd_keep_ex(const float* x, const float* params, const float* dout, float* dparams, float* dx,
                               float* scratch, uint32_t* counters, uint32_t M, const mipsf_grid_meta* meta_host,
                               int feat_layout, uint32_t flags, void* stream);
/* The same in two halves.  The routing of the scatter (which table slices every sample touches: a third of the
 * backward's time) depends on x only: mipsf_hashgrid_route may run as soon as x exists -- e.g. on a second stream next
 * to the forward pass -- into the same scratch buffer, and mipsf_hashgrid_bwd_routed then does the rest. */
int mipsf_hashgrid_route(const float* x, float* scratch, uint32_t M, const mipsf_grid_meta* meta_host, void* stream);
int mipsf_hashgrid_bwd_routed(const float* x, const float* params, const float* dout, float* dparams, float* dx,
                              float* scratch, uint32_t M, const mipsf_grid_meta* meta_host, int layout, void* stream);
/* parity probe: idx[(i*L + level)*8 + corner] = entry index inside the level (uint32). */
int mipsf_hashgrid_indices(const float* x, uint32_t* idx, uint32_t M, const mipsf_grid_meta* meta_host,
                           void* stream);

/* --------------------------------------------------------------- frequency (a6) */
int mipsf_freq_fwd(const float* x, float* out, uint32_t M, uint32_t n_dims, uint32_t n_freq, void* stream);
int mipsf_freq_bwd(const float* x, const float* dout, float* dx, uint32_t M, uint32_t n_dims,
                   uint32_t n_freq, void* stream);

/* ------------------------------------------------------------------ decoder (a7) */
/* Architecture is the reference's fixed one: e = [x(3), pe(48)] -> 128 -> 128 = [sdf_emb 64 | rgb_emb 64];
 * rgb = L(115->3)([rgb_emb, e]); prob = softmax(L(128->5)(relu(L(96->128)([sdf_emb, grid32])))). */
typedef struct mipsf_decoder_weights {   /* nn.Linear layout: weight [out,in] row-major, bias [out] */
    const float* w_pts0; const float* b_pts0;   /* [128,51]  */
    const float* w_pts2; const float* b_pts2;   /* [128,128] */
    const float* w_rgb0; const float* b_rgb0;   /* [3,115]   */
    const float* w_sdf0; const float* b_sdf0;   /* [128,96]  */
    const float* w_sdf2; const float* b_sdf2;   /* [5,128]   */
} mipsf_decoder_weights;

typedef struct mipsf_decoder_grads {
    float* w_pts0; float* b_pts0; float* w_pts2; float* b_pts2; float* w_rgb0; float* b_rgb0;
    float* w_sdf0; float* b_sdf0; float* w_sdf2; float* b_sdf2;
} mipsf_decoder_grads;

/* sizes (in floats) of the scratch buffers the caller must provide */
uint32_t mipsf_decoder_packed_floats(void);            /* MFMA operand images of the weights        */
uint64_t mipsf_decoder_saved_floats(uint32_t M);       /* activations kept for backward              */
uint64_t mipsf_decoder_dact_floats(uint32_t M);        /* pre-activation gradients (chain -> wgrad)  */
uint64_t mipsf_decoder_wgrad_partial_floats(void);     /* per-block partial weight gradients         */

/* repack nn.Linear weights into MFMA A-operand images (run after every optimiser step) */
int mipsf_decoder_pack(const mipsf_decoder_weights* w_host_struct, float* packed, void* stream);
/* host mirror of the same packing (plain CPU; used by the layout unit tests) */
int mipsf_decoder_pack_host(const mipsf_decoder_weights* w_host_ptrs, float* packed_host);

/* Forward.  pe_mode 0: positional encoding computed in-kernel from x (n_freq = 8) -- `embed_pos` ignored.
 *           pe_mode 1: `embed_pos` [M,48] is an input (module API of MLP_reg.forward).
 * feat: grid features in `feat_layout`; x: [M,3]; out: [M,10] = rgb(3) sdf entropy prob(5).
 * saved: nullable; when given, activations for backward are stored (mipsf_decoder_saved_floats(M)). */
int mipsf_decoder_fwd(const float* packed, const float* feat, int feat_layout, const float* x,
                      const float* embed_pos, int pe_mode, float* out, float* saved, uint32_t M, void* stream);
/* SDF column only: what JointEncoding.query_sdf keeps of MLP_reg.forward (model/scene_rep.py:106-107,
 * model/decoder.py:53-75; callers: RandomOptimizer.get_fitness RandomOptimizer.py:113-131, Mesher SDF grids).
 * sdf: [M].  Bit-identical to column 3 of mipsf_decoder_fwd; layer 2 computes its sdf_emb half only, no rgb head. */
int mipsf_decoder_fwd_sdf(const float* packed, const float* feat, int feat_layout, const float* x,
                          const float* embed_pos, int pe_mode, float* sdf, uint32_t M, void* stream);
/* ---- the same forward on the f16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate; csrc/decoder16.hip).
 * precision MIPSF_PREC_F16X3: every fp32 operand is carried as hi + lo halves and a product is three MFMAs
 *   (hi*hi + hi*lo + lo*hi): ~3e-7 relative, the training path -- `saved` has the layout mipsf_decoder_bwd_chain /
 *   _wgrad read, results agree with mipsf_decoder_fwd to fp32 round-off class.
 * precision MIPSF_PREC_F16: plain f16 operands (11 bits), fp32 accumulate: forward-only consumers with a stated
 *   tolerance (RandomOptimizer fitness; BASELINE config 5 "fp16 decoder on CDNA4"), `saved` must be NULL.
 * precision MIPSF_PREC_BF16X6: every fp32 operand -- weight and activation -- is carried EXACTLY as three bf16 pieces
 *   (8 + 8 + 8 = fp32's 24 significant bits, fp32's exponent range: v_mfma_f32_32x32x16_bf16) and a product is SIX MFMAs
 *   (p0*p0 + p0*p1 + p1*p0 + p1*p1 + p0*p2 + p2*p0); the dropped pairs are below 2^-23 of |a||w|, under the rounding of the
 *   fp32 accumulation: the arithmetic of the reference's fp32 nn.Linear layers (model/decoder.py:32-50) on the 16-bit matrix
 *   pipe.  Needs a buffer packed by mipsf_decoder_pack16_ex(..., MIPSF_PREC_BF16X6, ...) (_packed16_floats_ex floats: the f16
 *   layout with bf16 planes 0, 1 in place of hi, lo + plane 2 behind it); training (`saved`, lean_record) as for f16x3.
 * Positional encoding is always computed in-kernel (pe_mode 0).  sdf_only != 0: out is [M] (column 3 only).
 * packed16: mipsf_decoder_packed16_floats() floats written by mipsf_decoder_pack16 (compact hi / lo operand images of the two
 * narrow heads + their biases in fp32, hi and lo operand images of the three hidden layers, forward and backward sets); the
 * layout is private to the library (csrc/decoder_layout.h): a buffer packed by one build is for that build's kernels. */
#define MIPSF_PREC_F32 0
#define MIPSF_PREC_F16X3 1
#define MIPSF_PREC_F16 2
#define MIPSF_PREC_BF16X3 3
#define MIPSF_PREC_BF16X6 4
uint32_t mipsf_decoder_packed16_floats(void);
int mipsf_decoder_pack16(const mipsf_decoder_weights* w_host_struct, float* packed16, void* stream);
/* ... for a given arithmetic: MIPSF_PREC_F16X3 / MIPSF_PREC_F16 = the two calls above; MIPSF_PREC_BF16X6 = the three bf16
 * planes.  A buffer packed for one family must not be handed to the kernels of the other. */
uint32_t mipsf_decoder_packed16_floats_ex(int precision);
int mipsf_decoder_pack16_ex(const mipsf_decoder_weights* w_host_struct, float* packed16, int precision, void* stream);
int mipsf_decoder_fwd16(const float* packed16, const float* feat, int feat_layout, const float* x, float* out,
                        float* saved, int sdf_only, int precision, uint32_t M, void* stream);
/* lean_record != 0 (f16x3 with `saved` only): the record keeps H2, H3 and the ReLU masks but NOT H1 -- a third of the
 * record's bytes; valid when the weight gradients come from mipsf_decoder_wgrad16 with `packed16` given (it recomputes H1
 * from x); the backward chain never reads H1.  The buffer keeps its size and layout (the H1 pieces stay unwritten). */
/* lean_record == 2: only the ReLU masks are kept (32 B per sample): all the backward chain reads of the record; for callers
 * that will not ask for weight gradients (a frozen decoder: tracking). */
int mipsf_decoder_fwd16_ex(const float* packed16, const float* feat, int feat_layout, const float* x, float* out,
                           float* saved, int sdf_only, int precision, int lean_record, uint32_t M, void* stream);
/* mipsf_decoder_bwd_chain on the f16 matrix cores (hi/lo split operands, fp32 accumulate; pe_mode 0 only): the same
 * outputs and the same `dact` record, so mipsf_decoder_wgrad follows it unchanged.  saved: as written by
 * mipsf_decoder_fwd / _fwd16 (only the ReLU masks are read).  dact may be NULL (all three entry points): no weight gradients
 * will be asked for (a frozen decoder), the record is not written. */
int mipsf_decoder_bwd_chain16(const float* packed16, int feat_layout, const float* x, const float* out, const float* dout,
                              const float* saved, float* dfeat, float* dx, float* dact, uint32_t M, void* stream);
/* The same with ZERO-TILE flags.  Samples behind the truncation band receive an exactly zero gradient from the losses
 * (scene_rep.py:58-78, helper_functions/utils.py:21-49: no mask covers them); along a ray they are the tail, so whole
 * 32-sample tiles are zero.  tile_live (mipsf_decoder_tile_words(M) words, NULL = the call above) receives lists of
 * the tiles with a non-zero incoming gradient (opaque: work counters + eight lists, csrc/decoder16.hip); a tile that is
 * not listed gets dfeat = dx = 0 and NO entry in `dact` (nothing else is read or written for it).  The buffer then
 * goes to mipsf_decoder_wgrad16_tiles, which visits the listed tiles only. */
uint64_t mipsf_decoder_tile_words(uint32_t M);
int mipsf_decoder_bwd_chain16_ex(const float* packed16, int feat_layout, const float* x, const float* out,
                                 const float* dout, const float* saved, float* dfeat, float* dx, float* dact,
                                 uint32_t* tile_live, uint32_t M, void* stream);
/* One launch fewer per training step: the forward clears the counters of the live-tile buffer its backward chain will fill
 * (tile_live_clear, mipsf_decoder_tile_words(M) words), the chain is told so (header_is_clear = 1; the buffer must not have
 * been used in between -- a second backward through the same record passes 0 and pays the memset). */
int mipsf_decoder_fwd16_ex2(const float* packed16, const float* feat, int feat_layout, const float* x, float* out,
                            float* saved, int sdf_only, int precision, int lean_record, uint32_t* tile_live_clear,
                            uint32_t M, void* stream);
/* flags of mipsf_decoder_bwd_chain16_ex2 (its last int; 1 is the former header_is_clear):
 *   MIPSF_CHAIN_HEADER_CLEAR  the forward cleared tile_live's counters (above)
 *   MIPSF_CHAIN_LEAN_DACT     `dact` keeps dG1 and the sdf_emb half of dH2 only -- dG3 and the rgb_emb half of dH2, half of the
 *                             record, are not written: each is ONE narrow product of the 5 logit / 3 colour gradients (kept in
 *                             `dact`'s small-row part) and, for dG3, the ReLU masks of `saved`; the weight-gradient call that
 *                             follows must be mipsf_decoder_wgrad16_tiles_ex with MIPSF_WGRAD_LEAN_DACT (f16x3, packed16 given),
 *                             which recomputes them bit for bit. */
#define MIPSF_CHAIN_HEADER_CLEAR 1
#define MIPSF_CHAIN_LEAN_DACT 2
/*   MIPSF_CHAIN_BF16X6        packed16 holds bf16 planes (mipsf_decoder_pack16_ex with MIPSF_PREC_BF16X6): the chain's products
 *                             run in the six-product bf16 arithmetic of mipsf_decoder_fwd16 (fp32 operands carried exactly) */
#define MIPSF_CHAIN_BF16X6 4
int mipsf_decoder_bwd_chain16_ex2(const float* packed16, int feat_layout, const float* x, const float* out,
                                  const float* dout, const float* saved, float* dfeat, float* dx, float* dact,
                                  uint32_t* tile_live, int flags, uint32_t M, void* stream);
/* Backward.  dout [M,10].  Outputs: dfeat (layout as feat), dx [M,3] (pe_mode 0: includes the PE chain),
 * dembed_pos [M,48] (pe_mode 1 only).  Weight gradients are ACCUMULATED into `grads`.
 * dact / partial: scratch of the sizes above. */
int mipsf_decoder_bwd(const float* packed, const float* feat, int feat_layout, const float* x,
                      const float* embed_pos, int pe_mode, const float* out, const float* dout,
                      const float* saved, float* dfeat, float* dx, float* dembed_pos,
                      const mipsf_decoder_grads* grads_host_struct, float* dact, float* partial,
                      uint32_t M, void* stream);
/* The two halves of mipsf_decoder_bwd as separate entry points (same arguments; used when the activation
 * chain and the weight-gradient GEMMs are to be scheduled or timed separately):
 *   _bwd_chain : d(out) -> dfeat, dx, dembed_pos, dact          (register-chained MFMA, no LDS)
 *   _wgrad     : (saved, dact) -> weight/bias gradients          (LDS-transposed MFMA GEMMs + reduce) */
int mipsf_decoder_bwd_chain(const float* packed, int feat_layout, const float* x, int pe_mode,
                            const float* out, const float* dout, const float* saved, float* dfeat, float* dx,
                            float* dembed_pos, float* dact, uint32_t M, void* stream);
int mipsf_decoder_wgrad(const float* feat, int feat_layout, const float* x, const float* embed_pos, int pe_mode,
                        const float* saved, const float* dact, const mipsf_decoder_grads* grads_host_struct,
                        float* partial, uint32_t M, void* stream);

/* mipsf_decoder_wgrad with a choice of arithmetic for its three large products (d w_sdf0, d w_pts2, d w_pts0):
 * MIPSF_PREC_F32 = the fp32-input MFMA above; MIPSF_PREC_BF16X3 = bf16 matrix cores on hi/lo split operands (16-17
 * significant bits per operand, fp32 exponent range, fp32 accumulate): a weight gradient is a sum over every sample
 * of the batch, its error stays ~1e-7 of its magnitude. */
int mipsf_decoder_wgrad_ex(const float* feat, int feat_layout, const float* x, const float* embed_pos, int pe_mode,
                           const float* saved, const float* dact, const mipsf_decoder_grads* grads_host_struct,
                           float* partial, int precision, uint32_t M, void* stream);

/* The same weight gradients from the same records by the STREAMING kernel of csrc/wgrad16.hip: no LDS, the 16-bit matrix
 * cores transpose the records (an exact 0/1-matrix product per 16-bit plane) and multiply them.  arithmetic:
 * MIPSF_PREC_F16X3 (hi + lo f16 planes, every 32 x 32 gradient block under its own power-of-two scale: fp32-class),
 * MIPSF_PREC_BF16X6 (three bf16 planes, six products, no scale: fp32-class, slower), MIPSF_PREC_BF16X3 (two planes,
 * 2^-16).  pe_mode 0 only (the positional encoding is recomputed in-kernel). */
int mipsf_decoder_wgrad16(const float* feat, int feat_layout, const float* x, const float* saved, const float* dact,
                          const mipsf_decoder_grads* grads_host_struct, float* partial, int arithmetic, uint32_t M,
                          void* stream);
/* packed16 != NULL (MIPSF_PREC_F16X3 only): H1 is not read from `saved` but RECOMPUTED from x with the forward's own
 * layer-1 operand images (bit-identical to what the forward computed) -- the companion of the lean record of
 * mipsf_decoder_fwd16_ex.  packed16 == NULL: mipsf_decoder_wgrad16. */
int mipsf_decoder_wgrad16_ex(const float* packed16, const float* feat, int feat_layout, const float* x, const float* saved,
                             const float* dact, const mipsf_decoder_grads* grads_host_struct, float* partial,
                             int arithmetic, uint32_t M, void* stream);
/* ... over the tiles listed by mipsf_decoder_bwd_chain16_ex only (tile_live as written there; NULL = every tile). */
int mipsf_decoder_wgrad16_tiles(const float* packed16, const float* feat, int feat_layout, const float* x,
                                const float* saved, const float* dact, const uint32_t* tile_live,
                                const mipsf_decoder_grads* grads_host_struct, float* partial, int arithmetic, uint32_t M,
                                void* stream);
/* ... with flags.  MIPSF_WGRAD_LEAN_DACT: `dact` is the lean gradient record of mipsf_decoder_bwd_chain16_ex2 with
 * MIPSF_CHAIN_LEAN_DACT (f16x3 with packed16 only): dG3 and the rgb_emb half of dH2 are recomputed from the small rows and the
 * ReLU masks of `saved`. */
#define MIPSF_WGRAD_LEAN_DACT 1u
int mipsf_decoder_wgrad16_tiles_ex(const float* packed16, const float* feat, int feat_layout, const float* x,
                                   const float* saved, const float* dact, const uint32_t* tile_live,
                                   const mipsf_decoder_grads* grads_host_struct, float* partial, int arithmetic,
                                   uint32_t flags, uint32_t M, void* stream);

/* -------------------------------------------------- sample placement (a3 + a4) */
typedef struct mipsf_render_cfg {
    uint32_t n_uniform;      /* training.n_samples_d (or n_samples when no depth guidance) */
    uint32_t n_near;         /* training.n_range_d   (0 when no depth guidance)            */
    int perturb;             /* training.perturb > 0                                       */
    int use_bound;           /* grid.use_bound_normalize                                   */
    double bound_min[3];     /* mapping.bound[:,0]   (fp64, mipsfusion.py:94-96)           */
    double bound_max[3];
    double half_len[3];      /* mapping.localMLP_max_len                                   */
    double norm_factor;      /* training.norm_factor                                       */
    float trunc;             /* training.trunc                                             */
    float sc_factor;         /* data.sc_factor                                             */
    float depth_trunc;       /* cam.depth_trunc                                            */
    int rgb_missing_nonzero; /* training.rgb_missing != 0                                  */
    float emd_w;             /* EMD_w of JointEncoding.forward                             */
} mipsf_render_cfg;

/* z_uniform [n_uniform], z_near_offsets [n_near], z_near_nodepth [n_near]: the three torch.linspace tables
 * (computed once on the host by torch so that placement is bit-identical).  target_d nullable (then
 * n_near must be 0).  noise [N,S] U[0,1) (nullable when !perturb).  Outputs: z_vals [N,S]; xn [N*S,3]
 * normalised fp32 coordinates; counts[N,2] (uint32, written) = per-ray {#front, #band} of
 * helper_functions/utils.py:33-44 (only when target_d != NULL); mipsf_render_fwd sums them. */
int mipsf_sample_rays(const float* rays_o, const float* rays_d, const float* target_d, const float* noise,
                      const float* z_uniform, const float* z_near_offsets, const float* z_near_nodepth,
                      const mipsf_render_cfg* cfg_host, float* z_vals, float* xn, uint32_t* counts,
                      uint32_t N, void* stream);
/* normalise arbitrary points (run_network, scene_rep.py:134-146): pts [M,3] fp32 -> xn [M,3] fp32 */
int mipsf_normalise_points(const float* pts, const mipsf_render_cfg* cfg_host, float* xn, uint32_t M,
                           void* stream);

/* ------------------------------------------------ compositing + losses (a8, a9) */
/* raw [N,S,10], z_vals [N,S] -> per-ray rgb[N,3], depth, depth_var, disp, acc [N], weights [N,S] (nullable).
 * When `losses` != NULL (training): target_rgb [N,3], target_d [N,1], counts[N,2] from mipsf_sample_rays;
 * losses[8] = {rgb_loss, depth_loss, sdf_loss, fs_loss, psnr, fs_weight, sdf_weight, n_valid_depth};
 * partial: scratch [N*8]. */
int mipsf_render_fwd(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                     const uint32_t* counts, const mipsf_render_cfg* cfg_host, float* rgb, float* depth,
                     float* depth_var, float* disp, float* acc, float* weights, float* losses,
                     float* partial, uint32_t N, uint32_t S, void* stream);
/* The same with the training objective formed in the same launch: loss_total[0] = sum_k loss_weights[k] * losses[k], k < 4
 * (both device pointers; MIPSFusion.get_loss_from_ret, mipsfusion.py:142-152). */
int mipsf_render_fwd_ex(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                        const uint32_t* counts, const mipsf_render_cfg* cfg_host, float* rgb, float* depth,
                        float* depth_var, float* disp, float* acc, float* weights, float* losses,
                        float* partial, const float* loss_weights, float* loss_total, uint32_t N, uint32_t S,
                        void* stream);
/* The same in ONE launch: ticket = one uint32 the CALLER keeps (zero before the first call, left at zero by every call; one
 * per stream): the last workgroup of the render kernel finishes the losses (fp64 sums of the fp32 per-ray rows in a fixed
 * order: the result does not depend on which workgroup is last).  ticket NULL = mipsf_render_fwd_ex. */
int mipsf_render_fwd_ex2(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                         const uint32_t* counts, const mipsf_render_cfg* cfg_host, float* rgb, float* depth,
                         float* depth_var, float* disp, float* acc, float* weights, float* losses,
                         float* partial, const float* loss_weights, float* loss_total, uint32_t* ticket, uint32_t N,
                         uint32_t S, void* stream);
/* floats of `partial` that mipsf_render_fwd* need for N rays: max(8 N, 18 ceil(N / 16)) -- the one-launch form keeps one row
 * of nine doubles per 16-ray workgroup there, which exceeds 8 N floats for N < 3 */
uint64_t mipsf_render_partial_floats(uint32_t N);
/* A SHARE of a ray-data-parallel batch (helper_functions/utils.py:43-47 forms fs_weight / sdf_weight from
 * counts over the WHOLE batch, scene_rep.py:218 averages depth_loss over the batch's valid rays): the per-ray maps of this
 * share and the nine fp64 sums its losses are made of -- {rgb_sq, depth_sq(valid), fs_sq, sdf_sq, fs_emd, sdf_emd, n_valid,
 * n_front, n_band} -- in sums[9] (device).  The caller adds the shares' sums (72 bytes) and finishes the losses of the whole
 * batch with mipsf_loss_finalize_sums(N_total); mipsf_render_bwd_ex2(N_norm = N_total) differentiates the share. */
int mipsf_render_fwd_sums(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                          const uint32_t* counts, const mipsf_render_cfg* cfg_host, float* rgb, float* depth,
                          float* depth_var, float* disp, float* acc, float* weights, float* partial, double* sums,
                          uint32_t* ticket, uint32_t N, uint32_t S, void* stream);
int mipsf_loss_finalize_sums(const double* sums, const mipsf_render_cfg* cfg_host, uint32_t N_total, uint32_t S,
                             float* losses, const float* loss_weights, float* loss_total, void* stream);
/* Gradients wrt raw.  g_losses[4] (device): d total / d {rgb_loss, depth_loss, sdf_loss, fs_loss};
 * g_rgb [N,3], g_depth [N] nullable extra gradients on the rendered maps.  draw [N,S,10] is written. */
int mipsf_render_bwd(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                     const uint32_t* counts, const float* losses, const mipsf_render_cfg* cfg_host,
                     const float* g_losses, const float* g_rgb, const float* g_depth, float* draw,
                     uint32_t N, uint32_t S, void* stream);
/* g_losses nullable; g_total[1] (device, nullable): gradient of mipsf_render_fwd_ex's loss_total -- the kernel uses
 * g_losses[k] + g_total[0] * loss_weights[k]. */
int mipsf_render_bwd_ex(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                        const uint32_t* counts, const float* losses, const mipsf_render_cfg* cfg_host,
                        const float* g_losses, const float* g_total, const float* loss_weights, const float* g_rgb,
                        const float* g_depth, float* draw, uint32_t N, uint32_t S, void* stream);
/* ... of a SHARE of a batch: N rays here, losses normalised over N_norm >= N rays (mipsf_loss_finalize_sums) */
int mipsf_render_bwd_ex2(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                         const uint32_t* counts, const float* losses, const mipsf_render_cfg* cfg_host,
                         const float* g_losses, const float* g_total, const float* loss_weights, const float* g_rgb,
                         const float* g_depth, float* draw, uint32_t N, uint32_t N_norm, uint32_t S, void* stream);
/* Row gather of the ray table + ray construction from the pose parameters + sample placement in one launch
 * (mipsf_gather_pose_rays_fwd + mipsf_sample_rays: keyframeSet.py:264-290, mipsfusion.py:320-322, scene_rep.py:156-179);
 * rays_o / rays_d are not written.  Backward: d(xn) -> pose gradients in one launch (mipsf_rays_bwd + mipsf_pose_rays_bwd_ex);
 * scratch: mipsf_place_pose_scratch_floats floats whose first word is a ticket (zero on entry, zero on return). */
int mipsf_gather_pose_place_fwd(const float* db, uint64_t n_rows, const int64_t* idx, const float* fixed_poses,
                                const float* rot, const float* trans, uint32_t F, uint32_t K, const int64_t* owner,
                                const float* noise, const float* z_uniform, const float* z_near_offsets,
                                const float* z_near_nodepth, const mipsf_render_cfg* cfg_host, float* d_cam, float* rgb,
                                float* depth, float* z_vals, float* xn, uint32_t* counts, uint32_t N, void* stream);
uint64_t mipsf_place_pose_scratch_floats(uint32_t F, uint32_t K, uint32_t N);
int mipsf_place_pose_bwd(const float* dxn, const float* z_vals, const mipsf_render_cfg* cfg_host, const float* rot, uint32_t F,
                         uint32_t K, const int64_t* owner, const float* d_cam, float* d_rot, float* d_trans, float* scratch,
                         uint32_t N, uint32_t S, int accumulate, void* stream);
/* dxn [N*S,3] -> d_rays_o [N,3], d_rays_d [N,3] (written) */
int mipsf_rays_bwd(const float* dxn, const float* z_vals, const mipsf_render_cfg* cfg_host, float* d_rays_o,
                   float* d_rays_d, uint32_t N, uint32_t S, void* stream);
/* dxn [M,3] -> dpts [M,3]: autograd of mipsf_normalise_points */
int mipsf_normalise_bwd(const float* dxn, const mipsf_render_cfg* cfg_host, float* dpts, uint32_t M,
                        void* stream);

/* ---------------------------------------------------- rays from poses (a2) */
/* poses_all = [fixed_poses (F x 4x4 row-major) | K optimisable poses given as quaternion (w,x,y,z) + translation];
 * owner[n] indexes poses_all (negative = from the end, as mipsfusion.py:315 does with -1);
 * rays_d[n] = R[owner[n]] * d_cam[n], rays_o[n] = t[owner[n]]   (mipsfusion.py:320-322, geometry_helper.py:11-17). */
int mipsf_pose_rays_fwd(const float* fixed_poses, const float* rot, const float* trans, uint32_t F, uint32_t K,
                        const int64_t* owner, const float* d_cam, float* rays_o, float* rays_d, uint32_t N,
                        void* stream);
/* mipsf_gather_rays (split outputs) + mipsf_pose_rays_fwd in one launch: the rows idx[N] of the ray table db [n_rows,7]
 * -> d_cam [N,3], rgb [N,3], depth [N] and the world rays of those directions under poses_all[owner]
 * (keyframeSet.py:264-290 + mipsfusion.py:296-322). */
int mipsf_gather_pose_rays_fwd(const float* db, uint64_t n_rows, const int64_t* idx, const float* fixed_poses,
                               const float* rot, const float* trans, uint32_t F, uint32_t K, const int64_t* owner,
                               float* d_cam, float* rgb, float* depth, float* rays_o, float* rays_d, uint32_t N,
                               void* stream);
/* d_rot [K,4], d_trans [K,3] are WRITTEN.  scratch: mipsf_pose_rays_scratch_floats(F, K, N) floats whose FIRST word
 * must be zero on entry (clear it once after allocating) and is zero again on return -- it is the ticket that lets
 * the last workgroup finish the reduction and the quaternion chain in the same launch; the rest needs no init. */
uint64_t mipsf_pose_rays_scratch_floats(uint32_t F, uint32_t K, uint32_t N);
int mipsf_pose_rays_bwd(const float* g_rays_o, const float* g_rays_d, const float* rot, uint32_t F, uint32_t K,
                        const int64_t* owner, const float* d_cam, float* d_rot, float* d_trans, float* scratch,
                        uint32_t N, void* stream);
/* accumulate != 0: d_rot / d_trans are ADDED to (the parameters' own .grad buffers: no separate accumulation pass) */
int mipsf_pose_rays_bwd_ex(const float* g_rays_o, const float* g_rays_d, const float* rot, uint32_t F, uint32_t K,
                           const int64_t* owner, const float* d_cam, float* d_rot, float* d_trans, float* scratch,
                           uint32_t N, int accumulate, void* stream);

/* ------------------------------------------------------------------- Adam (a11) */
/* One dense torch.optim.Adam step over n floats.  step = 1-based count after increment.
 * zero_grad != 0 clears grad in the same pass. */
int mipsf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, uint32_t step, int zero_grad,
                    void* stream);

/* hipGraph-capturable form: the two step-dependent scalars {lr/bc1, 1/sqrt(bc2)} are read from `hyper_dev` (device,
 * 2 floats) instead of being baked into the launch; mipsf_adam_advance increments the device step counter and
 * refreshes them (enqueue it once per optimiser step, before the _ex calls). */
int mipsf_adam_advance(int32_t* step_dev, float* hyper_dev, float lr, float beta1, float beta2, void* stream);
/* the same for up to MIPSF_ADAM_MAX_GROUPS parameter groups in one launch (host arrays of device pointers / scalars) */
#define MIPSF_ADAM_MAX_GROUPS 8
int mipsf_adam_advance_n(int32_t* const* step_dev, float* const* hyper_dev, const float* lr, const float* beta1,
                         const float* beta2, uint32_t n_groups, void* stream);
int mipsf_adam_step_ex(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, uint32_t step,
                       const float* hyper_dev, int zero_grad, void* stream);

/* the same step over up to 16 small tensors that share one param group (one launch; used for the decoder) */
#define MIPSF_ADAM_MAX_TENSORS 16
typedef struct mipsf_adam_tensors {
    uint32_t count;
    float* param[MIPSF_ADAM_MAX_TENSORS];
    float* grad[MIPSF_ADAM_MAX_TENSORS];
    float* exp_avg[MIPSF_ADAM_MAX_TENSORS];
    float* exp_avg_sq[MIPSF_ADAM_MAX_TENSORS];
    uint64_t numel[MIPSF_ADAM_MAX_TENSORS];
} mipsf_adam_tensors;
int mipsf_adam_step_multi(const mipsf_adam_tensors* tensors_host_struct, float lr, float beta1, float beta2,
                          float eps, float weight_decay, uint32_t step, int zero_grad, void* stream);
int mipsf_adam_step_multi_ex(const mipsf_adam_tensors* tensors_host_struct, float lr, float beta1, float beta2,
                             float eps, float weight_decay, uint32_t step, const float* hyper_dev, int zero_grad,
                             void* stream);

/* A whole optimiser step in ONE launch when every tensor is tiny (the pose optimisers of tracking / local BA: a [K,4]
 * and a [K,3] tensor in two param groups with their own lr): one workgroup advances each group's device step counter,
 * refreshes its {lr/bc1, 1/sqrt(bc2)} pair and applies the step to all tensors -- three launches (advance + one
 * multi-step per group) become one.  Arithmetic identical to mipsf_adam_advance_n + mipsf_adam_step_multi_ex. */
#define MIPSF_ADAM_SMALL_MAX_NUMEL 16384
typedef struct mipsf_adam_small {
    uint32_t n_groups, n_tensors;
    int32_t* step_dev[MIPSF_ADAM_MAX_GROUPS];
    float* hyper_dev[MIPSF_ADAM_MAX_GROUPS];
    float lr[MIPSF_ADAM_MAX_GROUPS], beta1[MIPSF_ADAM_MAX_GROUPS], beta2[MIPSF_ADAM_MAX_GROUPS];
    float eps[MIPSF_ADAM_MAX_GROUPS], weight_decay[MIPSF_ADAM_MAX_GROUPS];
    float* param[MIPSF_ADAM_MAX_TENSORS];
    float* grad[MIPSF_ADAM_MAX_TENSORS];
    float* exp_avg[MIPSF_ADAM_MAX_TENSORS];
    float* exp_avg_sq[MIPSF_ADAM_MAX_TENSORS];
    uint32_t numel[MIPSF_ADAM_MAX_TENSORS];
    uint32_t group_of[MIPSF_ADAM_MAX_TENSORS];
} mipsf_adam_small;
int mipsf_adam_step_small(const mipsf_adam_small* desc_host_struct, int zero_grad, void* stream);
/* The same descriptor without the size limit: the whole step of an optimiser with tensors of ANY size in one launch (the map
 * optimiser: hash table + decoder).  ticket: MIPSF_ADAM_TICKET_WORDS uint32 the caller keeps (zero before the first call,
 * left ready by every call; the block belongs to ONE optimiser: it also carries the scalars the last workgroup leaves for
 * that optimiser's next step). */
#define MIPSF_ADAM_TICKET_WORDS 576
int mipsf_adam_step_all(const mipsf_adam_small* desc_host_struct, int zero_grad, uint32_t* ticket, void* stream);

/* -------------------------------------------------- RandomOptimizer fitness (a12) */
/* sdf [P,n] (column 3 of run_network output, stride `sdf_stride` floats) , valid [n] ->
 * mean_masked[P] = mean_j(valid_j * |sdf*trunc|)  (RandomOptimizer.py:125-129) */
int mipsf_ro_fitness(const float* raw, uint32_t raw_stride, const float* target_d, float trunc,
                     float* mean_masked, uint32_t P, uint32_t n, void* stream);
/* the same on the output of mipsf_decoder_fwd_sdf: sdf[p*n + j], or sdf[j*P + p] when point_major != 0
 * (the order mipsf_ro_particles_pm writes) */
int mipsf_ro_fitness_sdf(const float* sdf, const float* target_d, float trunc, float* mean_masked, uint32_t P,
                         uint32_t n, int point_major, void* stream);

/* -------------------------------------- RandomOptimizer particle step */
/* The search state lives on the device so that a tracking round needs no host round trip (the reference
 * synchronises on `if success_flag:` every round, RandomOptimizer.py:204,211).  state[MIPSF_RO_STATE_FLOATS]:
 *   [0..8] rot_cur (row-major 3x3)   [9..11] trans_cur   [12..17] search size (initial_scaling_factor x 6 at start)
 *   written by _ro_update: [18] success flag, [19] mean SDF of the advanced particles, [20] fitness of particle 0,
 *   [21..27] weighted mean transform (qw qx qy qz tx ty tz), [28] number of advanced particles. */
#define MIPSF_RO_STATE_FLOATS 32
/* pst [P,6] pre-sampled particle template; rays_d_cam [n,3], target_d [n] of the lattice pixels ->
 * pst7 [P,7] rescaled 7-D particle poses (pose_6D_to_7D, RandomOptimizer.py:57-63) and
 * xn [P*n,3]: lattice points moved by every particle's absolute pose (get_abs_pose :72-76, batch_points_trans
 * :84-88) and normalised like run_network (scene_rep.py:134-142), ready for mipsf_hashgrid_fwd. */
int mipsf_ro_particles(const float* pst, const float* state, const float* rays_d_cam, const float* target_d,
                       const mipsf_render_cfg* cfg_host, float* xn, float* pst7, uint32_t P, uint32_t n,
                       void* stream);
/* same, samples written point-major (xn[(j*P + p)*3 ..]): the 64 samples of a hash-grid wavefront are then 64
 * particles' copies of one lattice point, i.e. the same few table cells -- 2-3x faster grid lookups for the round */
int mipsf_ro_particles_pm(const float* pst, const float* state, const float* rays_d_cam, const float* target_d,
                          const mipsf_render_cfg* cfg, float* xn, float* pst7, uint32_t P, uint32_t n, void* stream);
/* mean_masked [P] from mipsf_ro_fitness -> advanced-particle weights, weighted mean transform, new rot/trans and
 * search size in `state` (RandomOptimizer.py:196-224; sdf_weight = 1000, rescale = tracking.RO.rescaling_factor). */
int mipsf_ro_update(const float* mean_masked, const float* pst7, float* state, float sdf_weight, float rescale,
                    uint32_t P, void* stream);

/* ---------------------------------------- keyframe ray database gather */
/* db [n_rows,7] = rows [direction(3) | rgb(3) | depth] of the device-resident keyframe database
 * (model/keyframeSet.py:25); idx [N] int64 flat row indices generated on the host (python `random.sample`, so the
 * index stream is the reference's; negative = from the end).  Any of the outputs may be NULL:
 * rays7 [N,7] (what sample_rays_in_submap returns, keyframeSet.py:386-436), d_cam [N,3], rgb [N,3], depth [N]
 * (the three slices mipsfusion.py:315-317 uploads every iteration). */
int mipsf_gather_rays(const float* db, uint64_t n_rows, const int64_t* idx, uint32_t N, float* rays7, float* d_cam,
                      float* rgb, float* depth, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIPSF_H */
