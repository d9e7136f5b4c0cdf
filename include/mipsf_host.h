/* libmipsf_hostrng.so -- host-side (CPU, no GPU code) replicas of the three library routines the reference's samplers
 * spend their host time in.  Each entry point produces what the routine it replaces produces, bit for bit, from that
 * routine's own generator state; mipsfusion_amd/hostrng.py compares them against the routines once per process and uses
 * the routines themselves on any mismatch.  Sources: mipsfusion_amd/csrc/host/hostrng.c, hosttopk.cpp.
 *
 *   reference call site                                         replaced routine            entry point
 *   helper_functions/sampling_helper.py:30, :62                 torch.randn_like (CPU)      mipsf_mt_normal_f32
 *   model/scene_rep.py:176                                      torch.rand (CPU)            mipsf_mt_uniform_f32
 *   helper_functions/sampling_helper.py:24-33, :55-68           valid * |draw| + torch.topk mipsf_topk_valid_scores
 *   model/keyframeSet.py:386-436, mipsfusion.py:135-138         random.sample(range(n), k)  mipsf_py_sample_range
 */
#ifndef MIPSF_HOST_H
#define MIPSF_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* A Mersenne twister (MT19937) position.  For the torch entry points: at::mt19937's (state, left, next) as found in
 * torch.get_rng_state() (CPUGeneratorImplState: state u64[624] at byte 24, left at 8, next at 16).  For the python entry
 * point: random.getstate()[1] = state[624] followed by the index, which goes into `next`; `left` is unused. */
typedef struct {
    uint32_t state[624];
    int32_t left;
    uint32_t next;
} mipsf_mt;

/* out[0..n) = tensor.uniform_() of float32: (random() & 0xffffff) * 2^-24 per value. */
void mipsf_mt_uniform_f32(mipsf_mt* g, float* out, int64_t n);

/* out[0..n) = tensor.normal_() of float32, n >= 16 (torch's vectorised path: a uniform fill, then Box-Muller on blocks of
 * 8 + 8 values, the last block recomputed over the tail as torch does).  threads: reserved (the fill is single-threaded).
 * Returns 0. */
int mipsf_mt_normal_f32(mipsf_mt* g, float* out, int64_t n, int threads);

/* out_idx[0..k) = torch.topk(scores, k)[1] for scores[i] = (depth[i] > 0 and not blocked[i]) * |draw[i]|, in torch's
 * order (ties included).  blocked: n bytes or NULL; scratch: 8 k bytes.  Returns 0, or -1 when torch would not take its
 * partial_sort path (k * 64 > n) or the draw holds a NaN / an infinity: the caller then runs torch.topk itself. */
int mipsf_topk_valid_scores(const float* depth, const float* draw, const uint8_t* blocked, int64_t n, int64_t k,
                            int64_t* out_idx, void* scratch);

/* out[0..k) = random.sample(range(n), k) of CPython 3.10 (both branches).  scratch: 8 n bytes.  Returns 0, or -1 for
 * arguments python refuses or n >= 2^32 (the caller then calls python's function, which raises or answers). */
int mipsf_py_sample_range(mipsf_mt* g, int64_t n, int64_t k, int64_t* out, void* scratch);

/* version of this interface */
int mipsf_hostrng_abi(void);

#ifdef __cplusplus
}
#endif
#endif
