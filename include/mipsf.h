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
 * Shape of the interface (ABI 2): a kernel family with options takes ONE argument block
 * (`mipsf_<family>(const mipsf_<family>_args* a, void* stream)`): zero-initialise the block, set
 * `struct_size = sizeof(block)` (the library refuses a block of another size: a caller built against another header)
 * and the fields you use; every optional field means "off" when zero / NULL.  Scratch sizes come from one query,
 * mipsf_buffer_size().  45 entry points.
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
#define MIPSF_ABI_VERSION 2      /* round 5: one argument block per kernel family instead of a suffix per option */

/* ------------------------------------------------------------------ errors / info */
const char* mipsf_last_error(void);
int mipsf_abi_version(void);
/* number of compute units of the current device (used to size persistent grids); <0 on error */
int mipsf_device_cu_count(void);

/* level table of the multiresolution hash grid (a5); filled by mipsf_hashgrid_meta_init */
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

/* ------------------------------------------------------------- scratch / record sizes */
/* Number of elements (floats, or uint32 words where the name says WORDS) of a buffer the caller must provide.
 * `n` = M samples or N rays, `a`, `b` = the small integers named below, meta (host) for the grid entries.
 * (uint64_t)-1 = bad query (mipsf_last_error says why); 0 is a size (an empty batch). */
#define MIPSF_SIZE_HASHGRID_BWD_SCRATCH 1     /* n = M, a = (dx != NULL), meta: scratch of mipsf_hashgrid_bwd / _route     */
#define MIPSF_SIZE_HASHGRID_COUNTER_WORDS 2   /* meta: the caller-kept counter block of mipsf_hashgrid_bwd                  */
#define MIPSF_SIZE_DECODER_PACKED 3           /* fp32 MFMA operand images of the decoder's weights                          */
#define MIPSF_SIZE_DECODER_SAVED 4            /* n = M: activations kept for the backward                                   */
#define MIPSF_SIZE_DECODER_DACT 5             /* n = M: pre-activation gradients (chain -> weight gradients)                */
#define MIPSF_SIZE_DECODER_WGRAD_PARTIAL 6    /* per-workgroup partial weight gradients                                     */
#define MIPSF_SIZE_DECODER_PACKED16 7         /* a = MIPSF_PREC_*: 16-bit operand images (mipsf_decoder_pack16)             */
#define MIPSF_SIZE_DECODER_TILE_WORDS 8       /* n = M: the live-tile lists of mipsf_decoder_bwd_chain16                    */
#define MIPSF_SIZE_RENDER_PARTIAL 9           /* n = N: `partial` of mipsf_render_fwd = max(8 N, 18 ceil(N / 16))           */
#define MIPSF_SIZE_PLACE_POSE_SCRATCH 10      /* n = N, a = F, b = K: mipsf_place_pose_bwd                                  */
#define MIPSF_SIZE_POSE_RAYS_SCRATCH 11       /* n = N, a = F, b = K: mipsf_pose_rays_bwd                                   */
uint64_t mipsf_buffer_size(int which, uint32_t n, uint32_t a, uint32_t b, const mipsf_grid_meta* meta_host);

/* --------------------------------------------------------------- hash grid (a5) */
/* Host-only: fill the level table exactly as tiny-cuda-nn's GridEncodingTemplated constructor does. */
int mipsf_hashgrid_meta_init(mipsf_grid_meta* meta_host, uint32_t n_levels, uint32_t n_features,
                             uint32_t log2_hashmap_size, uint32_t base_resolution, double per_level_scale);

/* Feature layouts.  AOS: out[i*L*F + level*F + f] (what tcnn.Encoding returns).
 * LEVEL_MAJOR: out[(level*M + i)*F + f] (internal layout between the grid and the decoder). */
#define MIPSF_FEAT_AOS 0
#define MIPSF_FEAT_LEVEL_MAJOR 1

/* x: [M,3] fp32 already normalised (scene_rep.py:140-142 + :119); params: [n_params]; out: [M,L*F].
 * jac (nullable): also store the Jacobian d out / d x, jac[((level*3 + d)*M + i)*2 + f] = d out_f(level) / d x_d (the
 * quantity tcnn's kernel_grid_backward_input recomputes from the table).  With it the backward obtains dL/dx from a
 * streaming pass (mipsf_hashgrid_dx_from_jac) instead of gathering 8 table entries per level again. */
int mipsf_hashgrid_fwd(const float* x, const float* params, float* out, float* jac, uint32_t M,
                       const mipsf_grid_meta* meta_host, int layout, void* stream);
/* dx [M,3] += sum over levels of jac . dL/dout   (bit-identical to the dx part of mipsf_hashgrid_bwd)
 * tile_live (nullable): only the 32-sample tiles listed in the buffer mipsf_decoder_bwd_chain16 filled for the same batch --
 * the other samples have a zero feature gradient and add nothing. */
int mipsf_hashgrid_dx_from_jac(const float* jac, const float* dout, float* dx, const uint32_t* tile_live, uint32_t M,
                               const mipsf_grid_meta* meta_host, int layout, void* stream);
/* dparams (nullable: frozen grid) += scatter of dL/dout (accumulated on chip in LDS slices, see hashgrid.hip);
 * dx (nullable) += dL/dx [M,3].  Replaces tcnn's kernel_grid_backward (+ _input) reached through model/encodings.py:14-25.
 *   scratch    MIPSF_SIZE_HASHGRID_BWD_SCRATCH floats owned by the caller
 *   counters   nullable: MIPSF_SIZE_HASHGRID_COUNTER_WORDS uint32 the CALLER keeps between calls (all zero before the first
 *              call; every call leaves the block ready for the next one): one launch fewer -- the routed scatter is then three
 *              launches (route, accumulate, fold of split bins).  One block per stream: calls that share a block must be
 *              ordered.  NULL: the block sits in `scratch` and one more launch clears it
 *   flags      MIPSF_HG_DPARAMS_ZERO: the caller vouches that dparams is all zero on entry (the gradient buffer of an optimiser
 *              that clears it, a fresh allocation): table slices are stored instead of read-modify-written.
 *              MIPSF_HG_ROUTED: `scratch` was filled by mipsf_hashgrid_route for this x (the routing third of the call
 *              depends on x only and may run earlier, e.g. on a second stream next to the forward pass) */
#define MIPSF_HG_DPARAMS_ZERO 1u
#define MIPSF_HG_ROUTED 2u
typedef struct mipsf_hashgrid_bwd_args {
    uint32_t struct_size;
    uint32_t M;
    const float* x;
    const float* params;
    const float* dout;
    float* dparams;                 /* nullable: frozen grid */
    float* dx;                      /* nullable */
    float* scratch;
    uint32_t* counters;             /* nullable */
    const mipsf_grid_meta* meta;    /* host */
    int feat_layout;
    uint32_t flags;
} mipsf_hashgrid_bwd_args;
int mipsf_hashgrid_bwd(const mipsf_hashgrid_bwd_args* args_host, void* stream);
int mipsf_hashgrid_route(const float* x, float* scratch, uint32_t M, const mipsf_grid_meta* meta_host, void* stream);
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

#define MIPSF_PREC_F32 0
#define MIPSF_PREC_F16X3 1
#define MIPSF_PREC_F16 2
#define MIPSF_PREC_BF16X3 3
#define MIPSF_PREC_BF16X6 4

/* ---- fp32-input matrix cores (exact fp32 products; csrc/decoder.hip) */
/* repack nn.Linear weights into MFMA A-operand images (run after every optimiser step); MIPSF_SIZE_DECODER_PACKED floats */
int mipsf_decoder_pack(const mipsf_decoder_weights* w_host_struct, float* packed, void* stream);
/* host mirror of the same packing (plain CPU; used by the layout unit tests) */
int mipsf_decoder_pack_host(const mipsf_decoder_weights* w_host_ptrs, float* packed_host);
/* Forward.  pe_mode 0: positional encoding computed in-kernel from x (n_freq = 8) -- `embed_pos` ignored.
 *           pe_mode 1: `embed_pos` [M,48] is an input (module API of MLP_reg.forward).
 * feat: grid features in `feat_layout`; x: [M,3]; out: [M,10] = rgb(3) sdf entropy prob(5).
 * saved: nullable; when given, activations for backward are stored (MIPSF_SIZE_DECODER_SAVED). */
int mipsf_decoder_fwd(const float* packed, const float* feat, int feat_layout, const float* x,
                      const float* embed_pos, int pe_mode, float* out, float* saved, uint32_t M, void* stream);
/* SDF column only: what JointEncoding.query_sdf keeps of MLP_reg.forward (model/scene_rep.py:106-107,
 * model/decoder.py:53-75; callers: RandomOptimizer.get_fitness RandomOptimizer.py:113-131, Mesher SDF grids).
 * sdf: [M].  Bit-identical to column 3 of mipsf_decoder_fwd; layer 2 computes its sdf_emb half only, no rgb head. */
int mipsf_decoder_fwd_sdf(const float* packed, const float* feat, int feat_layout, const float* x,
                          const float* embed_pos, int pe_mode, float* sdf, uint32_t M, void* stream);
/* Backward.  dout [M,10].  Outputs: dfeat (layout as feat), dx [M,3] (pe_mode 0: includes the PE chain),
 * dembed_pos [M,48] (pe_mode 1 only).  Weight gradients are ACCUMULATED into `grads`.
 * dact / partial: scratch (MIPSF_SIZE_DECODER_DACT / _WGRAD_PARTIAL). */
int mipsf_decoder_bwd(const float* packed, const float* feat, int feat_layout, const float* x,
                      const float* embed_pos, int pe_mode, const float* out, const float* dout,
                      const float* saved, float* dfeat, float* dx, float* dembed_pos,
                      const mipsf_decoder_grads* grads_host_struct, float* dact, float* partial,
                      uint32_t M, void* stream);
/* The two halves of mipsf_decoder_bwd as separate entry points (used when the activation chain and the weight-gradient
 * GEMMs are to be scheduled or timed separately):
 *   _bwd_chain : d(out) -> dfeat, dx, dembed_pos, dact          (register-chained MFMA, no LDS)
 *   _wgrad     : (saved, dact) -> weight/bias gradients          (LDS-transposed MFMA GEMMs + reduce); precision =
 *                MIPSF_PREC_F32, or MIPSF_PREC_BF16X3 = bf16 matrix cores on hi/lo split operands for its three large
 *                products (16-17 significant bits per operand, fp32 accumulate: ~1e-7 of a gradient's magnitude) */
int mipsf_decoder_bwd_chain(const float* packed, int feat_layout, const float* x, int pe_mode,
                            const float* out, const float* dout, const float* saved, float* dfeat, float* dx,
                            float* dembed_pos, float* dact, uint32_t M, void* stream);
int mipsf_decoder_wgrad(const float* feat, int feat_layout, const float* x, const float* embed_pos, int pe_mode,
                        const float* saved, const float* dact, const mipsf_decoder_grads* grads_host_struct,
                        float* partial, int precision, uint32_t M, void* stream);

/* ---- the same decoder on the 16-bit matrix cores (v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate; csrc/decoder16.hip,
 *      csrc/wgrad16.hip).  Positional encoding always in-kernel (pe_mode 0).
 * precision MIPSF_PREC_BF16X6 (the default of the Python modules): every fp32 operand -- weight, activation, gradient -- is
 *   carried EXACTLY as three bf16 pieces (8 + 8 + 8 = fp32's 24 significant bits, fp32's exponent range) and a product is SIX
 *   MFMAs (p0*p0 + p0*p1 + p1*p0 + p1*p1 + p0*p2 + p2*p0); the dropped pairs are each <= 2^-24 of |a||w|, comparable to ONE
 *   rounding of the fp32 accumulation: the arithmetic of the reference's fp32 nn.Linear layers (model/decoder.py:32-50).
 * precision MIPSF_PREC_F16X3: hi + lo f16 halves (22-23 operand bits), three MFMAs per product: ~3e-7 relative.
 * precision MIPSF_PREC_F16: plain f16 operands (11 bits): forward-only consumers with a stated tolerance (`saved` NULL).
 * packed16: MIPSF_SIZE_DECODER_PACKED16(precision) floats written by mipsf_decoder_pack16 for the SAME family (F16X3 and F16
 * share one buffer, BF16X6 has its own; the layout is private to the library build, csrc/decoder_layout.h). */
int mipsf_decoder_pack16(const mipsf_decoder_weights* w_host_struct, float* packed16, int precision, void* stream);
/* Forward (MLP_reg.forward, model/decoder.py:53-75).
 *   sdf_only         != 0: out is [M] (column 3 only), no record
 *   saved            nullable: the activation record for the backward
 *   lean_record      1: the record keeps H2, H3 and the ReLU masks but NOT H1 (a third of its bytes): valid when the weight
 *                    gradients come from mipsf_decoder_wgrad16 with packed16 given (it recomputes H1 from x); 2: only the ReLU
 *                    masks (32 B per sample): all the backward chain reads; for a frozen decoder (tracking)
 *   tile_live_clear  nullable: the live-tile buffer the backward chain of THIS forward will fill: its counters are cleared by
 *                    this launch (the chain is then told MIPSF_CHAIN_HEADER_CLEAR: one memset launch fewer per step) */
typedef struct mipsf_decoder_fwd16_args {
    uint32_t struct_size;
    uint32_t M;
    const float* packed16;
    const float* feat;
    const float* x;
    float* out;
    float* saved;                   /* nullable */
    uint32_t* tile_live_clear;      /* nullable */
    int feat_layout;
    int precision;
    int sdf_only;
    int lean_record;
    uint32_t packed16_floats;       /* 0 = unchecked; else the size of the packed16 buffer: refused unless it is
                                       MIPSF_SIZE_DECODER_PACKED16 of `precision`'s family (an f16 buffer handed to the
                                       bf16x6 kernels would be read past its end) */
} mipsf_decoder_fwd16_args;
int mipsf_decoder_fwd16(const mipsf_decoder_fwd16_args* args_host, void* stream);
/* Activation-gradient chain: d(out) -> dfeat, dx, dact (autograd of MLP_reg.forward with respect to its inputs); leaves the
 * `dact` record of mipsf_decoder_bwd_chain.  saved: as written by the forward (only the ReLU masks are read).
 *   dact       nullable: no weight gradients will be asked for (a frozen decoder), the record is not written
 *   tile_live  nullable, MIPSF_SIZE_DECODER_TILE_WORDS words: ZERO-TILE short cut.  Samples behind the truncation band get an
 *              exactly zero gradient from the losses (scene_rep.py:58-78, helper_functions/utils.py:21-49); along a ray they
 *              are the tail, so whole 32-sample tiles are zero.  The buffer receives lists of the tiles with a non-zero
 *              incoming gradient (opaque: work counters + eight lists); a tile that is not listed gets dfeat = dx = 0 and NO
 *              entry in `dact`.  Hand the buffer to mipsf_decoder_wgrad16 / mipsf_hashgrid_dx_from_jac
 *   flags      MIPSF_CHAIN_HEADER_CLEAR  the forward cleared tile_live's counters (tile_live_clear above) and nobody used the
 *                                        buffer since (a second backward through the same record passes 0 and pays the memset)
 *              MIPSF_CHAIN_LEAN_DACT     `dact` keeps dG1 and the sdf_emb half of dH2 only -- dG3 and the rgb_emb half of dH2
 *                                        are ONE narrow product each of the 5 logit / 3 colour gradients (kept in `dact`'s
 *                                        small-row part) and the ReLU masks; mipsf_decoder_wgrad16 with
 *                                        MIPSF_WGRAD_LEAN_DACT recomputes them bit for bit
 *              MIPSF_CHAIN_BF16X6        packed16 holds bf16 planes: the six-product arithmetic (else f16x3) */
#define MIPSF_CHAIN_HEADER_CLEAR 1
#define MIPSF_CHAIN_LEAN_DACT 2
#define MIPSF_CHAIN_BF16X6 4
typedef struct mipsf_decoder_chain16_args {
    uint32_t struct_size;
    uint32_t M;
    const float* packed16;
    const float* x;
    const float* out;
    const float* dout;
    const float* saved;
    float* dfeat;
    float* dx;
    float* dact;                    /* nullable */
    uint32_t* tile_live;            /* nullable */
    int feat_layout;
    int flags;
    uint32_t packed16_floats;       /* 0 = unchecked (see mipsf_decoder_fwd16_args) */
} mipsf_decoder_chain16_args;
int mipsf_decoder_bwd_chain16(const mipsf_decoder_chain16_args* args_host, void* stream);
/* Weight gradients from the records by the STREAMING kernel of csrc/wgrad16.hip: the 16-bit matrix cores transpose the
 * records (an exact 0/1-matrix product per 16-bit plane) and multiply them; ACCUMULATED into `grads`.
 *   arithmetic  MIPSF_PREC_F16X3 (hi + lo f16 planes, every 32 x 32 gradient block under its own power-of-two scale),
 *               MIPSF_PREC_BF16X6 (three bf16 planes, six products, no scale), MIPSF_PREC_BF16X3 (two planes, 2^-16)
 *   packed16    nullable (F16X3 / BF16X6): H1 is not read from `saved` but RECOMPUTED from x with the forward's own layer-1
 *               operand images (bit-identical) -- the companion of the lean record
 *   tile_live   nullable: only the tiles the chain listed
 *   flags       MIPSF_WGRAD_LEAN_DACT: `dact` is the lean gradient record (packed16 given) */
#define MIPSF_WGRAD_LEAN_DACT 1u
typedef struct mipsf_decoder_wgrad16_args {
    uint32_t struct_size;
    uint32_t M;
    const float* packed16;          /* nullable */
    const float* feat;
    const float* x;
    const float* saved;
    const float* dact;
    const uint32_t* tile_live;      /* nullable */
    const mipsf_decoder_grads* grads;   /* host struct of device pointers */
    float* partial;                 /* MIPSF_SIZE_DECODER_WGRAD_PARTIAL */
    int feat_layout;
    int arithmetic;
    uint32_t flags;
    uint32_t packed16_floats;       /* 0 = unchecked (see mipsf_decoder_fwd16_args) */
} mipsf_decoder_wgrad16_args;
int mipsf_decoder_wgrad16(const mipsf_decoder_wgrad16_args* args_host, void* stream);

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

/* z_uniform [n_uniform] (nullable when n_uniform = 0: training.n_samples_d = 0, the reference's `z_vals = z_samples` branch,
 * scene_rep.py:166-167), z_near_offsets [n_near], z_near_nodepth [n_near]: the three torch.linspace tables
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
/* sdf2weights / raw2outputs (+ the four training losses): model/scene_rep.py:58-103, 211-236; helper_functions/utils.py:21-111.
 * raw [N,S,10], z_vals [N,S] -> per-ray rgb[N,3], depth, depth_var, disp, acc [N], weights [N,S] (nullable).
 * Training (`losses` or `sums` given): target_rgb [N,3], target_d [N,1], counts[N,2] from mipsf_sample_rays;
 *   losses[8] = {rgb_loss, depth_loss, sdf_loss, fs_loss, psnr, fs_weight, sdf_weight, n_valid_depth}
 *   partial       scratch, MIPSF_SIZE_RENDER_PARTIAL floats
 *   loss_weights  nullable [4] (device): loss_total[0] = sum_k loss_weights[k] * losses[k] is formed in the same launch
 *                 (MIPSFusion.get_loss_from_ret, mipsfusion.py:142-152)
 *   ticket        nullable: one uint32 the CALLER keeps (zero before the first call, left at zero by every call; one per
 *                 stream): ONE launch -- the last workgroup of the render kernel finishes the losses (fp64 sums of the fp32
 *                 per-ray rows in a fixed order: the result does not depend on which workgroup is last).  NULL: two launches
 *   sums          nullable [9] fp64 (device): a SHARE of a ray-data-parallel batch (helper_functions/utils.py:43-47 forms
 *                 fs_weight / sdf_weight from counts over the WHOLE batch, scene_rep.py:218 averages depth_loss over the
 *                 batch's valid rays): instead of finishing the losses the call leaves the nine sums they are made of --
 *                 {rgb_sq, depth_sq(valid), fs_sq, sdf_sq, fs_emd, sdf_emd, n_valid, n_front, n_band}; the caller adds the
 *                 shares' sums (72 bytes) and finishes with mipsf_loss_finalize_sums(N_total); `losses` must be NULL */
typedef struct mipsf_render_fwd_args {
    uint32_t struct_size;
    uint32_t N, S;
    const float* raw;
    const float* z_vals;
    const float* target_rgb;        /* nullable (evaluation) */
    const float* target_d;          /* nullable */
    const uint32_t* counts;         /* nullable */
    const mipsf_render_cfg* cfg;    /* host */
    float* rgb; float* depth; float* depth_var; float* disp; float* acc;
    float* weights;                 /* nullable */
    float* losses;                  /* nullable */
    float* partial;                 /* nullable when neither losses nor sums */
    const float* loss_weights;      /* nullable */
    float* loss_total;              /* nullable */
    uint32_t* ticket;               /* nullable */
    double* sums;                   /* nullable */
    float* draw;                    /* nullable; [N,S,10]: ALSO the backward of the objective, in the same launch -- d loss_total /
                                     * d raw for an objective gradient of exactly 1 (what mipsf_render_bwd writes for g_total = {1},
                                     * no other gradient; bit for bit).  Needs ticket, loss_weights, loss_total and S <= 128.  A block
                                     * whose struct_size ends before this field is accepted (draw = NULL) */
} mipsf_render_fwd_args;
int mipsf_render_fwd(const mipsf_render_fwd_args* args_host, void* stream);
int mipsf_loss_finalize_sums(const double* sums, const mipsf_render_cfg* cfg_host, uint32_t N_total, uint32_t S,
                             float* losses, const float* loss_weights, float* loss_total, void* stream);
/* Gradients wrt raw: draw [N,S,10] is written.
 *   g_losses      nullable [4] (device): d total / d {rgb_loss, depth_loss, sdf_loss, fs_loss}
 *   g_total       nullable [1] (device): gradient of loss_total; the kernel uses g_losses[k] + g_total[0] * loss_weights[k]
 *   g_rgb [N,3], g_depth [N]  nullable extra gradients on the rendered maps
 *   N_norm        0 = N; > N: the rays are a SHARE of a batch whose losses were normalised over N_norm rays
 *                 (mipsf_loss_finalize_sums) */
typedef struct mipsf_render_bwd_args {
    uint32_t struct_size;
    uint32_t N, S, N_norm;
    const float* raw;
    const float* z_vals;
    const float* target_rgb;
    const float* target_d;
    const uint32_t* counts;
    const float* losses;
    const mipsf_render_cfg* cfg;    /* host */
    const float* g_losses;          /* nullable */
    const float* g_total;           /* nullable */
    const float* loss_weights;      /* nullable */
    const float* g_rgb;             /* nullable */
    const float* g_depth;           /* nullable */
    float* draw;
    uint32_t flags;                 /* MIPSF_RENDER_BWD_*; a block whose struct_size ends before this field is accepted (0) */
} mipsf_render_bwd_args;
/* draw already holds mipsf_render_fwd's `draw` (the gradient for g_total = {1}): the kernel returns at once when g_total[0] is
 * exactly 1 and rewrites draw otherwise.  g_total must be the only gradient given */
#define MIPSF_RENDER_BWD_KEEP_IF_UNIT 1u
int mipsf_render_bwd(const mipsf_render_bwd_args* args_host, void* stream);
/* Row gather of the ray table + ray construction from the pose parameters + sample placement in one launch
 * (mipsf_pose_rays_fwd with a ray table + mipsf_sample_rays: keyframeSet.py:264-290, mipsfusion.py:320-322,
 * scene_rep.py:156-179); rays_o / rays_d are not written.  Backward: d(xn) -> pose gradients in one launch (mipsf_rays_bwd +
 * mipsf_pose_rays_bwd); scratch: MIPSF_SIZE_PLACE_POSE_SCRATCH floats whose first word is a ticket (zero on entry, zero on
 * return). */
int mipsf_gather_pose_place_fwd(const float* db, uint64_t n_rows, const int64_t* idx, const float* fixed_poses,
                                const float* rot, const float* trans, uint32_t F, uint32_t K, const int64_t* owner,
                                const float* noise, const float* z_uniform, const float* z_near_offsets,
                                const float* z_near_nodepth, const mipsf_render_cfg* cfg_host, float* d_cam, float* rgb,
                                float* depth, float* z_vals, float* xn, uint32_t* counts, uint32_t N, void* stream);
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
/* db (nullable): rows idx[N] of the ray table db [n_rows,7] are gathered first -> d_cam [N,3], rgb [N,3], depth [N] are
 * WRITTEN (keyframeSet.py:264-290 + mipsfusion.py:296-322); db NULL: d_cam is an input, idx / rgb / depth are ignored. */
int mipsf_pose_rays_fwd(const float* db, uint64_t n_rows, const int64_t* idx, const float* fixed_poses,
                        const float* rot, const float* trans, uint32_t F, uint32_t K, const int64_t* owner,
                        float* d_cam, float* rgb, float* depth, float* rays_o, float* rays_d, uint32_t N,
                        void* stream);
/* One pose handed from one stage of a frame to the next on the device: the reference passes a 4x4 between its stages
 * (mipsfusion.py:479-501, 556-575), i.e. every stage starts from matrix_to_quaternion of the previous stage's matrix, a
 * unit quaternion.  src: MIPSF_POSE_MATRIX = 12 floats [3x3 rotation row-major | translation] (words 0..11 of the
 * RandomOptimizer's state), MIPSF_POSE_QUATERNION = 7 floats [w x y z | t] (a stage's pose Parameters; must not alias the
 * destination) -> rot[4] (w >= 0), trans[3].  geometry_helper.qt_to_transform_matrix / matrix_to_quaternion
 * (geometry_helper.py:11-33) operation by operation in IEEE fp32.  No host round trip. */
#define MIPSF_POSE_MATRIX 0
#define MIPSF_POSE_QUATERNION 1
int mipsf_pose_handover(const float* src, int src_kind, float* rot, float* trans, void* stream);
/* d_rot [K,4], d_trans [K,3] are written, or ADDED to when accumulate != 0 (the parameters' own .grad buffers: no separate
 * accumulation pass).  scratch: MIPSF_SIZE_POSE_RAYS_SCRATCH floats whose FIRST word must be zero on entry (clear it once
 * after allocating) and is zero again on return -- it is the ticket that lets the last workgroup finish the reduction and
 * the quaternion chain in the same launch; the rest needs no init. */
int mipsf_pose_rays_bwd(const float* g_rays_o, const float* g_rays_d, const float* rot, uint32_t F, uint32_t K,
                        const int64_t* owner, const float* d_cam, float* d_rot, float* d_trans, float* scratch,
                        uint32_t N, int accumulate, void* stream);

/* ------------------------------------------------------------------- Adam (a11) */
/* One dense torch.optim.Adam step over n floats (mipsfusion.py:580-584, 190, 330-335).  step = 1-based count after
 * increment.  zero_grad != 0 clears grad in the same pass.
 * hyper_dev (nullable): hipGraph-capturable form -- the two step-dependent scalars {lr/bc1, 1/sqrt(bc2)} are read from device
 * memory (2 floats) instead of being baked into the launch (`step` is then ignored); mipsf_adam_advance_n increments the
 * device step counters and refreshes the pairs (enqueue it once per optimiser step, before the step calls). */
int mipsf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, uint32_t step,
                    const float* hyper_dev, int zero_grad, void* stream);
/* up to MIPSF_ADAM_MAX_GROUPS parameter groups in one launch (host arrays of device pointers / scalars) */
#define MIPSF_ADAM_MAX_GROUPS 8
int mipsf_adam_advance_n(int32_t* const* step_dev, float* const* hyper_dev, const float* lr, const float* beta1,
                         const float* beta2, uint32_t n_groups, void* stream);

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
                          float eps, float weight_decay, uint32_t step, const float* hyper_dev, int zero_grad,
                          void* stream);

/* A whole optimiser step in ONE launch when every tensor is tiny (the pose optimisers of tracking / local BA: a [K,4]
 * and a [K,3] tensor in two param groups with their own lr): one workgroup advances each group's device step counter,
 * refreshes its {lr/bc1, 1/sqrt(bc2)} pair and applies the step to all tensors -- three launches (advance + one
 * multi-step per group) become one.  Arithmetic identical to mipsf_adam_advance_n + mipsf_adam_step_multi. */
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
 * (the order mipsf_ro_particles writes with point_major) */
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
/* point_major != 0: samples written point-major (xn[(j*P + p)*3 ..]): the 64 samples of a hash-grid wavefront are then 64
 * particles' copies of one lattice point, i.e. the same few table cells -- 2-3x faster grid lookups for the round */
int mipsf_ro_particles(const float* pst, const float* state, const float* rays_d_cam, const float* target_d,
                       const mipsf_render_cfg* cfg_host, float* xn, float* pst7, uint32_t P, uint32_t n,
                       int point_major, void* stream);
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
