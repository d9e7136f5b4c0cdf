// Host-side replica of the generator-free half of the reference's valid-pixel samplers (sampling_helper.py:24-33, :55-68):
//     scores = valid * |draw|      valid = (depth > 0), lattice pixels zeroed      (1.1 MB temporaries x 5 in torch)
//     idx    = torch.topk(scores, k)[1]
// in ONE pass over (depth, draw) with the index ORDER torch returns, ties included.  torch's CPU top-k for k * 64 <= n
// (ATen/native/cpu/TopKImpl.h) is std::partial_sort over (value, index) pairs with the comparator `a.value > b.value`
// (NaN first): a max-"heap" of the k best so far whose root is the WORST of them; an element enters only if it beats
// the root.  Elements that do not beat the root change nothing, so they can be rejected eight at a time with one vector
// compare against the root's value -- all but ~k ln(n/k) of them -- and the rest go through exactly the heap
// operations std::partial_sort performs (libstdc++'s __adjust_heap / __push_heap, restated below; make_heap and
// sort_heap are the library's own).  There is no shortcut around replaying the heap: its history decides the order of
// EQUAL scores, and equal scores among the k best are the rule, not the exception (Box-Muller values of 24-bit uniforms:
// a third of the calls at k = 1024, nearly all at k = 4096, measured).
// mipsfusion_amd/hostrng.py checks the result against torch.topk at start-up, with tie-heavy inputs (fewer valid pixels
// than k: the zeros' order is pure heap mechanics), and falls back to torch.
#include <immintrin.h>
#include <stdint.h>

#include "mipsf_host.h"

#include <algorithm>
#include <cmath>
#include <utility>

namespace {
// (value, index) as torch's queue holds them, in 8 bytes: k = 4096 entries stay inside the L1 cache.  The scores of this
// pass are never NaN (finite draws; the entry point refuses inputs where they could be), so TopKImpl.h's comparator for
// largest = true, `(isnan(x) && !isnan(y)) || x > y`, is `x > y`.
struct elem_t {
    float first;
    int32_t second;
};
struct Before {
    bool operator()(const elem_t& x, const elem_t& y) const { return x.first > y.first; }
};

// std::__adjust_heap(first, 0, len, value, comp): the hole at the root sinks to a leaf along the children that come
// LAST in comp's order, then `value` is pushed up from there (std::__push_heap).
inline void replace_root(elem_t* first, int64_t len, elem_t value) {
    int64_t hole = 0, child = 0;
    const int64_t inner = (len - 1) / 2;
    while (child < inner) {
        child = 2 * (child + 1);
        child -= (int64_t)(first[child].first > first[child - 1].first);     // (no branch: it is a coin flip)
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    int64_t parent = (hole - 1) / 2;
    while (hole > 0 && first[parent].first > value.first) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

inline float score_of(const float* depth, const float* draw, const uint8_t* blocked, int64_t i) {
    const bool valid = depth[i] > 0.0f && !(blocked && blocked[i]);
    return (valid ? 1.0f : 0.0f) * std::fabs(draw[i]);
}
}  // namespace

extern "C" {

// scratch: k (float, int32) pairs = 8 k bytes.  Returns 0; -1 when torch would not take its partial_sort path or the
// draw holds a NaN / an infinity (scores could be NaN, which torch sorts first): the caller then runs torch.topk itself.
int mipsf_topk_valid_scores(const float* depth, const float* draw, const uint8_t* blocked, int64_t n, int64_t k,
                            int64_t* out_idx, void* scratch) {
    if (k <= 0 || k > n || k * 64 > n || n > INT32_MAX) return -1;
    elem_t* heap = static_cast<elem_t*>(scratch);
    const Before comp;
    bool finite = true;
    for (int64_t i = 0; i < k; ++i) {
        heap[i] = elem_t{score_of(depth, draw, blocked, i), (int32_t)i};
        finite &= std::isfinite(draw[i]);
    }
    std::make_heap(heap, heap + k, comp);
    auto offer = [&](int64_t i) {
        const elem_t e{score_of(depth, draw, blocked, i), (int32_t)i};
        if (comp(e, heap[0])) replace_root(heap, k, e);
    };
    int64_t i = k;
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff)), zero = _mm256_setzero_ps();
    const __m256 inf = _mm256_set1_ps(INFINITY);
    __m256 bad = zero;
    for (; i + 8 <= n; i += 8) {
        // lanes that could beat the root: |draw| > root value (an upper bound of the score; the root only ever rises
        // while the block is processed, and every lane that passes is offered with the exact comparison)
        const __m256 a = _mm256_and_ps(_mm256_loadu_ps(draw + i), absmask);
        const __m256 thr = _mm256_set1_ps(heap[0].first);
        const __m256 pos = _mm256_cmp_ps(_mm256_loadu_ps(depth + i), zero, _CMP_GT_OQ);
        bad = _mm256_or_ps(bad, _mm256_cmp_ps(a, inf, _CMP_NLT_UQ));                          // NaN or infinity
        int m = _mm256_movemask_ps(_mm256_and_ps(_mm256_cmp_ps(a, thr, _CMP_GT_OQ), pos));
        while (m) {
            const int l = __builtin_ctz(m);
            m &= m - 1;
            offer(i + l);
        }
    }
    for (; i < n; ++i) {
        finite &= std::isfinite(draw[i]);
        offer(i);
    }
    if (!finite || _mm256_movemask_ps(bad)) return -1;
    std::sort_heap(heap, heap + k, comp);
    for (int64_t j = 0; j < k; ++j) out_idx[j] = heap[j].second;
    return 0;
}

}  // extern "C"
