// Multiresolution hash-grid encoding for gfx950 (replaces tinycudann.Encoding(HashGrid),
// reference call site model/encodings.py:11-26; arithmetic restated from tiny-cuda-nn 1.7
// grid.h / common_device.h -- see oracle/tcnn_cpu.py for the assumption list A1-A6).
//
// Work decomposition: one thread per (sample, level); every wave works on ONE level and on 64
// consecutive samples (consecutive samples are neighbours on a ray, so their cells coincide on the
// coarse levels and their table lines are shared).
//
// XCD-aware launch: the chip has 8 XCDs with private 4 MiB L2s and workgroup b lands on XCD b % 8.
// A hashed level's table is exactly 2^19 * 8 B = 4 MiB, so levels are pinned to XCDs:
// XCD k serves levels {k+8, k} one after the other (fine level first), which keeps each L2 filled with
// one level's table instead of thrashing all 34 MiB through every L2.  The mapping only affects speed.
#include "common.h"
#include "decoder_layout.h"      // layout of the backward chain's live-tile lists (mipsf_hashgrid_dx_from_jac)

namespace mipsf {

constexpr uint32_t P1 = 2654435761u;
constexpr uint32_t P2 = 805459861u;
constexpr int HG_BLOCK = 256;

struct Cell {
    uint32_t c[3];
    float f[3];
};

__device__ __forceinline__ Cell locate(const float* __restrict__ x, uint32_t i, float scale) {
    Cell r;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float pos = fmaf(scale, x[3 * (size_t)i + d], 0.5f);
        const float fl = floorf(pos);
        r.c[d] = (uint32_t)(int)fl;
        r.f[d] = pos - fl;
    }
    return r;
}

// entry index of corner (cx,cy,cz) inside a level of `size` entries.
// MODE 0: dense stride walk; in-box points wrap at most once (tcnn's half-cell offset), so the modulo is a
//         conditional subtract with the real `%` only for far-outside points.
// MODE 1: coherent prime hash, power-of-two level size (every hashed level in practice) -> mask.
// MODE 2: coherent prime hash, general size -> `%`.
template <int MODE>
__device__ __forceinline__ uint32_t corner_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res,
                                                 uint32_t size) {
    if (MODE == 0) {
        uint32_t idx = cx + cy * res + cz * res * res;
        if (idx >= size) {
            idx -= size;
            if (idx >= size) idx %= size;
        }
        return idx;
    }
    const uint32_t h = cx ^ (cy * P1) ^ (cz * P2);
    return MODE == 1 ? (h & (size - 1u)) : (h % size);
}

// dense walk applies while stride <= size; the hash takes over when size < res^3 (uint32 arithmetic,
// tcnn grid_index).  For 3-D grids with res <= 1625 no intermediate stride overflows.
__host__ __device__ inline bool level_is_hashed(uint32_t res, uint32_t size) {
    uint64_t stride = 1;
    for (int d = 0; d < 3; ++d) {
        if (stride > size) break;
        stride *= res;
        stride &= 0xFFFFFFFFull;
    }
    return size < stride;
}

__device__ __forceinline__ int decode_level(uint32_t n_levels, uint32_t nchunk, uint32_t& chunk) {
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u;
    const uint32_t j = b >> 3;
    const uint32_t slot = j / nchunk;
    chunk = j - slot * nchunk;
    const uint32_t nslots = (n_levels + 7u) / 8u;
    return (int)(xcd + 8u * (nslots - 1u - slot));
}

template <int MODE>
__device__ __forceinline__ void corner_indices_m(const Cell& cell, uint32_t res, uint32_t size, uint32_t idx[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        idx[c] = corner_index<MODE>(cell.c[0] + (c & 1), cell.c[1] + ((c >> 1) & 1), cell.c[2] + ((c >> 2) & 1),
                                    res, size);
    }
}

// wave-uniform dispatch on the level's addressing mode
__device__ __forceinline__ int level_mode(uint32_t res, uint32_t size) {
    if (!level_is_hashed(res, size)) return 0;
    return (size & (size - 1u)) == 0u ? 1 : 2;
}

__device__ __forceinline__ void corner_indices(int mode, const Cell& cell, uint32_t res, uint32_t size,
                                               uint32_t idx[8]) {
    if (mode == 0) corner_indices_m<0>(cell, res, size, idx);
    else if (mode == 1) corner_indices_m<1>(cell, res, size, idx);
    else corner_indices_m<2>(cell, res, size, idx);
}

__device__ __forceinline__ void corner_weights(const Cell& cell, float w[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float t = (c & 1) ? cell.f[0] : 1.0f - cell.f[0];
        t = t * ((c & 2) ? cell.f[1] : 1.0f - cell.f[1]);
        t = t * ((c & 4) ? cell.f[2] : 1.0f - cell.f[2]);
        w[c] = t;
    }
}

template <int LAYOUT>
__device__ __forceinline__ size_t feat_index(uint32_t i, uint32_t level, uint32_t M, uint32_t L) {
    return LAYOUT == MIPSF_FEAT_AOS ? ((size_t)i * L + level) * 2 : ((size_t)level * M + i) * 2;
}

// d f0 / d x_d (s0) and d f1 / d x_d (s1) of one level from its 8 corner values (tcnn kernel_grid_backward_input):
// scale * sum over the 4 corner pairs along d of w_other * (right - left)
__device__ __forceinline__ void level_jacobian(const Cell& cell, const float2 (&v)[8], float scale, int d, float& s0,
                                               float& s1) {
    s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
        const int d0 = d == 0 ? 1 : 0;
        const int d1 = d == 2 ? 1 : 2;
        const int b0 = sub & 1, b1 = (sub >> 1) & 1;
        float wgt = scale;
        wgt = wgt * (b0 ? cell.f[d0] : 1.0f - cell.f[d0]);
        wgt = wgt * (b1 ? cell.f[d1] : 1.0f - cell.f[d1]);
        const int left = (b0 << d0) | (b1 << d1);
        const int right = left | (1 << d);
        s0 = s0 + wgt * (v[right].x - v[left].x);
        s1 = s1 + wgt * (v[right].y - v[left].y);
    }
}

// ------------------------------------------------------------------------------ forward
// JAC: also store d out / d x of this (sample, level) -- 3 x (d f0/d x_d, d f1/d x_d), planes [L][3][M][2] -- so that
// the backward gets dL/dx from a streaming pass (hashgrid_dx_jac_kernel) instead of gathering the table again.
template <int LAYOUT, bool JAC>
__global__ __launch_bounds__(HG_BLOCK) void hashgrid_fwd_kernel(const float* __restrict__ x,
                                                                const float2* __restrict__ table,
                                                                float* __restrict__ out, float* __restrict__ jac,
                                                                uint32_t M, GridLevels g, uint32_t nchunk) {
    uint32_t chunk;
    const int level = decode_level(g.n_levels, nchunk, chunk);
    if (level >= (int)g.n_levels) return;
    const uint32_t i = chunk * HG_BLOCK + threadIdx.x;
    if (i >= M) return;

    const uint32_t off = g.offsets[level];
    const uint32_t size = g.offsets[level + 1] - off;
    const uint32_t res = g.res[level];
    const Cell cell = locate(x, i, g.scale[level]);
    uint32_t idx[8];
    corner_indices(level_mode(res, size), cell, res, size, idx);
    float w[8];
    corner_weights(cell, w);
    const float2* lvl = table + off;
    float2 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = lvl[idx[c]];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a0 = fmaf(w[c], v[c].x, a0);
        a1 = fmaf(w[c], v[c].y, a1);
    }
    // one 8-byte store per lane: a wavefront writes 512 contiguous bytes (level-major layout).  The first version
    // issued the two floats -- and the six of the Jacobian, 24 bytes apart from lane to lane -- as scalar stores:
    // 203 MB reached HBM for 134 MB of results (rocprofv3 WRITE_SIZE, profiles/r01_k)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t* dst = reinterpret_cast<f32x2_t*>(out + feat_index<LAYOUT>(i, level, M, g.n_levels));
    __builtin_nontemporal_store(f32x2_t{a0, a1}, dst);
    if (JAC) {
        // [L][3][M][2]: the three derivative pairs of a (sample, level) go to three planes, each written 8 bytes per
        // lane, contiguous across the wavefront
        f32x2_t* j2 = reinterpret_cast<f32x2_t*>(jac) + (size_t)level * 3 * M + i;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s0, s1;
            level_jacobian(cell, v, g.scale[level], d, s0, s1);
            __builtin_nontemporal_store(f32x2_t{s0, s1}, j2 + (size_t)d * M);
        }
    }
}

// ----------------------------------------------------------------------------- backward
// dL/dparams is a scatter of 16 floats per (sample, level).  Global fp32 atomics top out at ~18 G atomics/s on
// MI355X whatever the access pattern (measured: 4.0 ms for the 67 M atomics of a 4096x64 batch), so the scatter
// target is moved ON CHIP: the 256 CUs own 40 MiB of LDS.
//   * a level that fits 8192 entries (128 KiB of fp64 pairs) is one slice, a larger one is cut into 8192-entry
//     slices (power of two: slice = index >> 13) -- one "bin" per slice;
//   * ROUTE: every (sample, level) is sent to the bins its 8 corners fall into (a record is the sample index,
//     duplicates inside a bin are merged, so a bin holds at most M records and gets a fixed M-record region of
//     scratch: one pass, no counting pre-pass), so that no workgroup ever looks at a sample that does not touch
//     its slice.  (The first version let every slice owner scan ALL samples of its level:
//     52 x redundant index arithmetic on the hashed levels, 640 us.)
//   * ACCUMULATE: one workgroup per (bin, part of <= 24576 records) re-derives the corners of its samples and
//     adds the contributions that fall into its slice with LDS atomics in fp64.  ds_add_f64 runs at ~0.33 cycles
//     per lane-op per CU on gfx950, ds_add_f32 at 3.0 (serialised lane by lane; tools/micro/lds_atomic.hip), and
//     fp64 sums make the result independent of the arrival order to fp32 precision;
//   * the slice is added to dparams with coalesced read-modify-writes; bins with several parts go through
//     partial slices and a reduce kernel.  No global float atomics, no inter-workgroup communication.
constexpr int SC_BLOCK = 1024;                      // threads of an accumulate workgroup
// Layout of the fp64 slice in LDS.  Interleaved ([entry][2 features], round 1) puts every lane of one ds_add_f64 instruction on
// an address = 0 (or 8) mod 16: half of the LDS banks never take part, and random entries collide twice as often as they must
// (PMC, rounds 2-5: SQ_LDS_BANK_CONFLICT a third of the kernel's LDS cycles).  Planar ([feature][SC_MAX_SLICE entries]) spreads
// an instruction's lanes over all banks.
#ifndef MIPSF_SC_PLANAR
#define MIPSF_SC_PLANAR 1
#endif
#if MIPSF_SC_PLANAR
#define SC_ACC0(e) (e)
#define SC_ACC1(e) (sc_plane1 + (e))      /* sc_plane1 = plan.max_slice: the bin counts sit behind 2 x max_slice doubles */
#else
#define SC_ACC0(e) (2 * (e))
#define SC_ACC1(e) (2 * (e) + 1)
#endif
#ifndef MIPSF_RT_BLOCK
#define MIPSF_RT_BLOCK 512
#endif
// threads of a routing workgroup (x SC_ROUTE_UNR samples of one level).  The kernel is a chain of short phases between
// barriers (liveness loads, compaction, ranking with LDS atomics, one device atomic per bin, staging, write-out): with 512
// threads four workgroups share a CU and their phases overlap (1024: 128.7 us for the whole scatter, 512: 122.3, 256: 128.0)
constexpr int RT_BLOCK = MIPSF_RT_BLOCK;
#ifndef MIPSF_SC_MAX_SLICE
#define MIPSF_SC_MAX_SLICE 8192
#endif
// entries of the largest one-slice level: 8192 * 2 doubles = 128 KB of the CU's 160 KB of LDS (the persistent accumulate
// kernel keeps the bin counts, 32 KB for the 8192 bins of a 2^22 table, next to the slice)
constexpr uint32_t SC_MAX_SLICE = MIPSF_SC_MAX_SLICE;
#ifndef MIPSF_SC_SLICE_LOG2
#define MIPSF_SC_SLICE_LOG2 13
#endif
constexpr uint32_t SC_SLICE_LOG2 = MIPSF_SC_SLICE_LOG2;   // slices of multi-slice levels: 8192 entries
#ifndef MIPSF_SC_PART
#define MIPSF_SC_PART 24576
#endif
#ifndef MIPSF_SC_PART_DENSE
#define MIPSF_SC_PART_DENSE 8192
#endif
#ifndef MIPSF_SC_RUN
#define MIPSF_SC_RUN 4
#endif
#ifndef MIPSF_SC_SKIP_ZERO
#define MIPSF_SC_SKIP_ZERO 1    // experiments: 0 = records for samples with a zero feature gradient, too
#endif
// Records per accumulate workgroup.  A hashed level's bins are split only when they must be (a part that is not the bin's
// only one goes through a 64 KB partial slice and the reduce kernel); the dense levels' slices are small and their bins
// hold every live sample of the batch: they are cut finer, because the longest work item bounds the kernel's span once
// half of the samples are skipped (zero feature gradient) -- 24576-record parts: 111 us, 12288: 73 us.
constexpr uint32_t SC_PART = MIPSF_SC_PART;
constexpr uint32_t SC_PART_DENSE = MIPSF_SC_PART_DENSE;
constexpr uint32_t SC_PART_MIN = SC_PART < SC_PART_DENSE ? SC_PART : SC_PART_DENSE;
constexpr uint32_t SC_RUN = MIPSF_SC_RUN;                   // consecutive records merged per thread in the accumulate kernel
constexpr uint32_t SC_ROUTE_UNR = 4;                // samples per thread in the routing kernel
constexpr uint32_t SC_MAX_NS = 512;                 // slices per level (2^22-entry levels)
constexpr uint32_t SC_MAX_BINS = 8192;
#ifndef MIPSF_SC_MASKED_MAX_M
#define MIPSF_SC_MASKED_MAX_M (1u << 24)      // (test builds lower it to run the mask-less path on small batches)
#endif
constexpr uint32_t SC_MASKED_MAX_M = MIPSF_SC_MASKED_MAX_M;    // up to here a routing record has room for the 8-bit corner mask

struct ScatterPlan {
    uint32_t n_levels;
    // NB: every array is uint32_t on purpose.  With sub-dword arrays in a kernel-argument struct hipcc (ROCm 7.2)
    // folds `base + 2*level` into the SBASE of a scalar dword load; the hardware ignores SBASE's low two bits, so
    // odd levels silently read element [level-1].
    uint32_t n_slices[MIPSF_MAX_LEVELS];
    uint32_t slice_entries[MIPSF_MAX_LEVELS];
    uint32_t slice_shift[MIPSF_MAX_LEVELS];       // slice of table index i = i >> slice_shift (31: one slice)
    uint32_t bin0[MIPSF_MAX_LEVELS + 1];          // first bin of each level
    uint32_t n_bins;
    uint32_t max_items;                           // upper bound on accumulate work items for this M
    uint32_t max_slice;                           // entries of the largest slice (= the accumulate kernel's LDS / 16)
    uint32_t dense_bins;                          // bins [0, dense_bins) belong to the dense levels (they come first)
    uint32_t fresh;                               // 1: the caller vouches that dparams is all zero on entry: slices are stored, not added
    // scratch layout, in 4-byte words from the start of the scratch buffer
    uint32_t w_count, w_first, w_parts, w_nitems, w_items;
    uint64_t w_records, w_partial, w_end;         // records: n_bins regions of bin_cap words
};

static ScatterPlan make_plan(const GridLevels& g, uint32_t M) {
    ScatterPlan p = {};
    p.n_levels = g.n_levels;
    uint32_t bins = 0;
    for (uint32_t l = 0; l < g.n_levels; ++l) {
        const uint32_t size = g.offsets[l + 1] - g.offsets[l];
        // a level that does not fit one slice is cut into power-of-two slices: the routing kernel finds the slice of
        // a corner with one shift instead of a float division + two fix-ups per corner (8 corners x 4.2 M records)
        const bool one = size <= SC_MAX_SLICE;
        const uint32_t ns = one ? 1u : (size + (1u << SC_SLICE_LOG2) - 1) >> SC_SLICE_LOG2;
        p.n_slices[l] = ns;
        p.slice_entries[l] = one ? size : (1u << SC_SLICE_LOG2);
        p.slice_shift[l] = one ? 31u : SC_SLICE_LOG2;
        p.bin0[l] = bins;
        bins += ns;
    }
    p.bin0[g.n_levels] = bins;
    p.n_bins = bins;
    for (uint32_t l = 0; l < g.n_levels; ++l) {          // (hashed levels follow the dense ones: resolutions only grow)
        if (level_is_hashed(g.res[l], g.offsets[l + 1] - g.offsets[l])) break;
        p.dense_bins = p.bin0[l + 1];
    }
    for (uint32_t l = 0; l < g.n_levels; ++l) p.max_slice = p.slice_entries[l] > p.max_slice ? p.slice_entries[l] : p.max_slice;
    const uint64_t max_records = 8ull * g.n_levels * M;
    p.max_items = (uint32_t)((max_records + SC_PART_MIN - 1) / SC_PART_MIN) + bins;
    uint64_t w = 0;
    p.w_count = (uint32_t)w, w += bins;
    p.w_nitems = (uint32_t)w, w += 4;
    p.w_first = (uint32_t)w, w += bins;
    p.w_parts = (uint32_t)w, w += bins;
    w = (w + 3) / 4 * 4;
    p.w_items = (uint32_t)w;                                // end of the counter block {counts | head, tickets | first | parts}
    w = (w + 15) / 16 * 16;
    p.w_records = w, w += (uint64_t)bins * M;
    w = (w + 15) / 16 * 16;
    p.w_partial = w, w += (uint64_t)p.max_items * SC_MAX_SLICE * 2;
    p.w_end = w;
    return p;
}

__global__ void scatter_zero_kernel(uint32_t* __restrict__ ws, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ws[i] = 0u;
}

// rank of this lane's record inside its bin, LDS counter `cnt[s]`.  Levels with few bins (coarse dense levels:
// every lane of a wave wants the same one or two counters) aggregate per wave -- one atomic per distinct bin instead
// of 64 serialised same-address atomics.
__device__ __forceinline__ uint32_t ranked_add(uint32_t* cnt, uint32_t s, bool active, bool aggregate) {
    if (!aggregate) return active ? atomicAdd(&cnt[s], 1u) : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rank = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t s0 = (uint32_t)__shfl((int)s, leader, 64);
        const unsigned long long m = __ballot(active && s == s0);
        uint32_t b = 0;
        if ((int)lane == leader) b = atomicAdd(&cnt[s0], (uint32_t)__popcll(m));
        b = (uint32_t)__shfl((int)b, leader, 64);
        if (active && s == s0) rank = b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return rank;
}

// The slices the 8 corners of one (sample, level) fall into, each with the mask of its corners: slot k is in use when
// m[k] != 0.  GENERAL form: slot c = corner c when it is the first corner of its slice (28 + 64 comparisons).  The two
// FAST forms build at most four slots from what the addressing guarantees and say whether the guarantee held:
//   dense level   the index grows with the corner number, so all corners lie between corner 0's slice and corner 7's;
//                 two slots, valid when every corner is in one of the two (always, while a slice is thicker than the
//                 cell's index span res^2 + res + 1 and the point is inside the box);
//   hashed level, power-of-two size: x enters the hash as `cx ^ ...`, and cx, cx + 1 < 2^shift leave the bits >= shift
//                 -- the slice -- alone: the corners come in four x-pairs that share a slice; four hashes instead of 8.
// The routing kernel was VALU-bound (70 % of its wave cycles, ~320 vector instructions per (sample, level)).
__device__ __forceinline__ void route_groups_general(int mode, const Cell& cell, uint32_t res, uint32_t size,
                                                     uint32_t shift, uint32_t (&s)[8], uint32_t (&m)[8]) {
    uint32_t idx[8];
    corner_indices(mode, cell, res, size, idx);
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = idx[c] >> shift;            // (shift 31 = the level is one slice)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        bool first = true;
        uint32_t mm = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < c) first = first && (s[k] != s[c]);
            mm |= (s[k] == s[c]) ? (1u << k) : 0u;
        }
        m[c] = first ? mm : 0u;
    }
}

__device__ __forceinline__ bool route_groups_dense(const Cell& cell, uint32_t res, uint32_t size, uint32_t shift,
                                                   uint32_t (&s)[8], uint32_t (&m)[8]) {
    uint32_t idx[8];
    corner_indices_m<0>(cell, res, size, idx);
    const uint32_t lo = idx[0] >> shift, hi = idx[7] >> shift;
    uint32_t m_lo = 0, m_hi = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t q = idx[c] >> shift;
        m_lo |= (q == lo) ? (1u << c) : 0u;
        m_hi |= (q == hi) ? (1u << c) : 0u;
    }
    s[0] = lo, m[0] = m_lo;
    s[1] = hi, m[1] = hi != lo ? m_hi : 0u;
    return (m_lo | m_hi) == 0xffu;
}

__device__ __forceinline__ bool route_groups_hashed_pow2(const Cell& cell, uint32_t size, uint32_t shift,
                                                         uint32_t (&s)[8], uint32_t (&m)[8]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t cy = cell.c[1] + (p & 1), cz = cell.c[2] + (p >> 1);
        s[p] = ((cell.c[0] ^ (cy * P1) ^ (cz * P2)) & (size - 1u)) >> shift;
        m[p] = 3u << (2 * p);                                      // corners 2p (x) and 2p + 1 (x + 1)
    }
#pragma unroll
    for (int p = 1; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < p; ++q) {
            const bool same = m[p] != 0u && m[q] != 0u && s[q] == s[p];
            m[q] |= same ? m[p] : 0u;
            m[p] = same ? 0u : m[p];
        }
    // (cx AND cx + 1 below 2^shift.  Written as `cx + 1 < 2^shift` until round 6: a point just outside the box on the low
    // side has cx = (uint32_t)-1, cx + 1 wraps to 0 and the "guarantee" held for a pair whose corners hash into different
    // slices -- the second corner's contribution was added outside the LDS slice.  Found when the planar slice layout turned
    // those stray additions into visible 1e-11-sized gradients of other entries, tests/test_gpu_configs.py.)
    return cell.c[0] < (1u << shift) - 1u;
}

// One workgroup = one level x 4096 consecutive samples: rank the records inside the workgroup with LDS counters,
// reserve room in every bin with ONE global atomic per bin, write the sample indices.  (Staging the records bin by
// bin in LDS to make the stores coalesced was measured slower: 67 vs 61 us.)
//
// dout (optional: the feature gradient the records will be used with, in `layout`): a (sample, level) whose two feature
// gradients are exactly zero adds nothing to the table gradient and gets no record.  Half of a mapping batch is like that
// -- samples behind the truncation band carry no loss term and no rendering weight (45-56 % of the samples, on every
// level, tools/micro/dout_zero_probe.py) -- so routing and accumulation handle half the records.  Exact: the skipped
// contributions are +-0.
#ifndef MIPSF_RT_LEVELS
#define MIPSF_RT_LEVELS 1
#endif
// levels routed by ONE workgroup.  Which samples are live, their compaction and their coordinates do not depend on the
// level, so a workgroup could route its 2048 samples for level g, g + G, g + 2G, ... (G = ceil(n_levels / RT_LEVELS) level
// groups) and pay the liveness round trip, the scan with its three barriers and the x gather once.  MEASURED (round 5, whole
// scatter on the headline step, tools/replay.py): 1 level 108.6 us, 2: 116.3, 4: 116.4, 8: 136.8 -- the liveness loads are
// bound by their bytes, not by one latency (4 levels: 13 700 cycles instead of 4 500), per level 18 800 cycles instead of
// 19 900, and a quarter of the workgroups balance worse.  Kept as a switch.  Also measured and dropped: liveness from one bit
// per sample written by the backward chain (32 KB, warm in L2) instead of from the d feat planes -- 115 us either way: the
// phase's latency is covered by the other workgroups of the CU.  With several levels a sample counts as live when its
// feature gradient is non-zero on ANY of them (a record for a pair with a zero gradient adds +-0: exact).
constexpr uint32_t RT_LEVELS = MIPSF_RT_LEVELS;
__host__ __device__ inline uint32_t route_groups(uint32_t n_levels) { return (n_levels + RT_LEVELS - 1) / RT_LEVELS; }

template <int LAYOUT>
__global__ __launch_bounds__(RT_BLOCK) void scatter_route_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ dout, uint32_t M, GridLevels g,
                                                                ScatterPlan plan, uint32_t* __restrict__ ws,
                                                                uint32_t* __restrict__ cw) {
    __shared__ uint32_t cnt[SC_MAX_NS];
    __shared__ uint32_t base[SC_MAX_NS];
#ifndef MIPSF_SC_STAGE
#define MIPSF_SC_STAGE 1        // experiments: 0 = every lane stores its records straight to the bins
#endif
    // Records are collected bin by bin in LDS and leave as runs of consecutive words: a lane-per-record store sends 64
    // four-byte writes to 64 different lines (the bins of the 64 lanes), 10 M of them per launch -- half of this kernel's
    // time (ablation, tools/micro/route_probe.py: 52 us, 28 without the stores, 21 with a coalesced stand-in).
#ifndef MIPSF_SC_STAGE_CAP
#define MIPSF_SC_STAGE_CAP (10 * RT_BLOCK)
#endif
    // Records beyond the staging capacity go straight to their bins (lane-per-record stores).  A workgroup's 2048 samples
    // make <= 4 records each on most levels; with the dead half of a mapping batch skipped that is ~4000 records, and
    // 5120 words keep the kernel's LDS small enough for several workgroups per CU, whose barrier- and round-trip-separated
    // phases then overlap
    constexpr uint32_t STAGE_CAP = MIPSF_SC_STAGE_CAP;
    __shared__ uint32_t lstart[SC_MAX_NS];
    __shared__ uint32_t stage[MIPSF_SC_STAGE ? STAGE_CAP : 1];
    const uint32_t n_groups = route_groups(plan.n_levels);
    const uint32_t group = blockIdx.x % n_groups;
    const uint32_t chunk = blockIdx.x / n_groups;
#ifndef MIPSF_SC_AGG_MAX
#define MIPSF_SC_AGG_MAX 1   // measured on the mapping workload: 65 us (<= 1 bin), 67 (<= 2), 88 (<= 8)
#endif
#ifndef MIPSF_SC_ROUTE_FAST
#define MIPSF_SC_ROUTE_FAST 1   // experiments: 0 = the general grouping for every sample
#endif
#ifdef MIPSF_RT_TRACE
    unsigned long long tr_t[10];
    unsigned long long tr_acc[9] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#define RT_MARK(k) do { __builtin_amdgcn_sched_barrier(0); tr_t[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RT_MARK(k) do { } while (0)
#endif
    RT_MARK(0);
    const uint32_t s0 = chunk * (RT_BLOCK * SC_ROUTE_UNR);
    // COMPACTION.  The dead samples (zero gradient) are the tails of the rays, i.e. every wave of 64 consecutive samples
    // has some: skipping them lane by lane leaves the grouping and ranking below as expensive as before.  The workgroup's
    // live samples are first packed (order kept) into `live_list`; round u then works on entries u * RT_BLOCK + thread, and
    // a batch that is half dead takes two rounds of full waves instead of four of half-empty ones.
    __shared__ uint32_t live_list[RT_BLOCK * SC_ROUTE_UNR];
    __shared__ uint32_t wave_base[(RT_BLOCK / 64) * SC_ROUTE_UNR + 1];
    {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        unsigned long long alive[SC_ROUTE_UNR];
        // every liveness load of the workgroup's levels is in flight before the first is looked at: one round trip
        float2 gy[SC_ROUTE_UNR][RT_LEVELS];
        if (dout != nullptr) {
#pragma unroll
            for (uint32_t u = 0; u < SC_ROUTE_UNR; ++u) {
                const uint32_t i = s0 + u * RT_BLOCK + threadIdx.x;
#pragma unroll
                for (uint32_t k = 0; k < RT_LEVELS; ++k) {
                    const uint32_t level = group + k * n_groups;
                    gy[u][k] = (i < M && level < plan.n_levels)
                                   ? *reinterpret_cast<const float2*>(dout + feat_index<LAYOUT>(i, level, M, plan.n_levels))
                                   : make_float2(0.f, 0.f);
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < SC_ROUTE_UNR; ++u) {
            const uint32_t i = s0 + u * RT_BLOCK + threadIdx.x;
            bool live = i < M;
            if (live && dout != nullptr) {
                live = false;
#pragma unroll
                for (uint32_t k = 0; k < RT_LEVELS; ++k) live = live || !(gy[u][k].x == 0.0f && gy[u][k].y == 0.0f);   // (NaN gradients stay live)
            }
            alive[u] = __ballot(live);
            if (lane == 0) wave_base[1 + u * (RT_BLOCK / 64) + wave] = (uint32_t)__popcll(alive[u]);
        }
        RT_MARK(1);
        __syncthreads();
        static_assert((RT_BLOCK / 64) * SC_ROUTE_UNR <= 64, "one wave scans the (round, wave) counts");
        if (threadIdx.x < 64) {                      // inclusive prefix of the <= 64 (round, wave) counts
            uint32_t v = threadIdx.x < (RT_BLOCK / 64) * SC_ROUTE_UNR ? wave_base[1 + threadIdx.x] : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
                v += (int)threadIdx.x >= d ? o : 0u;
            }
            if (threadIdx.x < (RT_BLOCK / 64) * SC_ROUTE_UNR) wave_base[1 + threadIdx.x] = v;
            if (threadIdx.x == 0) wave_base[0] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < SC_ROUTE_UNR; ++u)
            if (alive[u] >> lane & 1ull)
                live_list[wave_base[u * (RT_BLOCK / 64) + wave] + (uint32_t)__popcll(alive[u] & ((1ull << lane) - 1ull))] =
                    s0 + u * RT_BLOCK + threadIdx.x;
        __syncthreads();
    }
    RT_MARK(2);
    const uint32_t n_live = wave_base[(RT_BLOCK / 64) * SC_ROUTE_UNR];
    // the samples this thread routes (round u: entry u * RT_BLOCK + thread of the list) and their coordinates: loaded once,
    // located on every level of the group
    uint32_t mine[SC_ROUTE_UNR];
    float xr[SC_ROUTE_UNR][3];
#pragma unroll
    for (uint32_t u = 0; u < SC_ROUTE_UNR; ++u) {
        const uint32_t c = u * RT_BLOCK + threadIdx.x;
        const uint32_t i = c < n_live ? live_list[c] : 0u;
        mine[u] = i;
#pragma unroll
        for (int d = 0; d < 3; ++d) xr[u][d] = x[3 * (size_t)i + d];
    }
    const bool masked = M <= SC_MASKED_MAX_M;
    uint32_t* rec = ws + plan.w_records;
#ifdef MIPSF_RT_TRACE
    tr_acc[0] = tr_t[1] - tr_t[0], tr_acc[1] = tr_t[2] - tr_t[1];
    uint32_t tr_levels = 0;
#endif
    for (uint32_t k = 0; k < RT_LEVELS; ++k) {
    const uint32_t level = group + k * n_groups;
    if (level >= plan.n_levels) break;
    RT_MARK(2);
    const uint32_t size = g.offsets[level + 1] - g.offsets[level];
    const uint32_t res = g.res[level];
    const float scale = g.scale[level];
    const uint32_t ns = plan.n_slices[level], bin0 = plan.bin0[level];
    const uint32_t shift = plan.slice_shift[level];
    const int mode = level_mode(res, size);
    const bool aggregate = ns <= MIPSF_SC_AGG_MAX;
    if (k != 0u) __syncthreads();                // (the previous level's write-out has read cnt / lstart / base / stage)
    for (uint32_t q = threadIdx.x; q < ns; q += RT_BLOCK) cnt[q] = 0u;
    __syncthreads();
    // across the barrier, per sample: 8 slots of (slice < 512, rank < 4096) as 16-bit halves, masks as bytes
    uint32_t sp[SC_ROUTE_UNR][4], rp[SC_ROUTE_UNR][4], mp[SC_ROUTE_UNR][2];
#pragma unroll
    for (uint32_t u = 0; u < SC_ROUTE_UNR; ++u) {
        const uint32_t c = u * RT_BLOCK + threadIdx.x;
        const bool live = c < n_live;
        uint32_t s[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        bool general = false;
        if (!__any(live)) {                          // (whole wave past the end of the list)
#pragma unroll
            for (int q = 0; q < 4; ++q) sp[u][q] = 0u, rp[u][q] = 0u;
            mp[u][0] = 0u, mp[u][1] = 0u;
            continue;
        }
        Cell cell;
#pragma unroll
        for (int d = 0; d < 3; ++d) {               // (locate(), from the registers)
            const float pos = fmaf(scale, xr[u][d], 0.5f);
            const float fl = floorf(pos);
            cell.c[d] = (uint32_t)(int)fl;
            cell.f[d] = pos - fl;
        }
        if (live) {
            if (MIPSF_SC_ROUTE_FAST && mode == 0) general = !route_groups_dense(cell, res, size, shift, s, m);
            else if (MIPSF_SC_ROUTE_FAST && mode == 1) general = !route_groups_hashed_pow2(cell, size, shift, s, m);
            else general = true;
        }
        if (__any(general)) {                       // (wave-uniform: the ranking below needs whole waves)
            if (general) route_groups_general(mode, cell, res, size, shift, s, m);
        }
        const bool wide = __any((m[4] | m[5] | m[6] | m[7]) != 0u) != 0;
        uint32_t r[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = ranked_add(cnt, s[q], m[q] != 0u, aggregate);
        if (wide) {
#pragma unroll
            for (int q = 4; q < 8; ++q) r[q] = ranked_add(cnt, s[q], m[q] != 0u, aggregate);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sp[u][q] = s[2 * q] | (s[2 * q + 1] << 16);
            rp[u][q] = r[2 * q] | (r[2 * q + 1] << 16);
        }
        mp[u][0] = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
        mp[u][1] = m[4] | (m[5] << 8) | (m[6] << 16) | (m[7] << 24);
    }
    RT_MARK(3);
    __syncthreads();
    RT_MARK(4);
    for (uint32_t q = threadIdx.x; q < ns; q += RT_BLOCK)
        base[q] = cnt[q] ? atomicAdd(&cw[plan.w_count + bin0 + q], cnt[q]) : 0u;
    if (MIPSF_SC_STAGE && threadIdx.x < 64) {       // exclusive prefix of the bin counts: where a bin starts in `stage`
        constexpr uint32_t PER = SC_MAX_NS / 64;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (uint32_t q2 = 0; q2 < PER; ++q2) {
            const uint32_t q = threadIdx.x * PER + q2;
            c[q2] = q < ns ? cnt[q] : 0u;
            sum += c[q2];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
            incl += (int)threadIdx.x >= d ? v : 0u;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (uint32_t q2 = 0; q2 < PER; ++q2) {
            const uint32_t q = threadIdx.x * PER + q2;
            if (q < ns) lstart[q] = run;
            run += c[q2];
        }
    }
    RT_MARK(5);
    __syncthreads();
    RT_MARK(6);
    // record = sample index | (the corners of the sample's cell that fall into this slice) << 24: the accumulate kernel
    // then hashes those corners only (on a hashed level 2 of 8: the pair along x) instead of all 8 plus 8 membership tests
#pragma unroll
    for (uint32_t u = 0; u < SC_ROUTE_UNR; ++u) {
        const uint32_t i = mine[u];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q == 4 && !__any(mp[u][1] != 0u)) break;
            const uint32_t m = (mp[u][q >> 2] >> (8 * (q & 3))) & 0xffu;
            if (m) {
                const uint32_t sl = (sp[u][q >> 1] >> (16 * (q & 1))) & 0xffffu;
                const uint32_t rk = (rp[u][q >> 1] >> (16 * (q & 1))) & 0xffffu;
                if (MIPSF_SC_STAGE) {
                    const uint32_t p = lstart[sl] + rk;
                    if (p < STAGE_CAP) {
                        stage[p] = masked ? (i | (m << 24)) : i;
                        continue;
                    }
                }
                rec[(size_t)(bin0 + sl) * M + base[sl] + rk] = masked ? (i | (m << 24)) : i;
            }
        }
    }
    RT_MARK(7);
    if (MIPSF_SC_STAGE) {
        __syncthreads();
        RT_MARK(8);
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        for (uint32_t q = wave; q < ns; q += RT_BLOCK / 64) {
            const uint32_t n = cnt[q], src = lstart[q];
            uint32_t* dstp = rec + (size_t)(bin0 + q) * M + base[q];
            for (uint32_t p = lane; p < n && src + p < STAGE_CAP; p += 64) dstp[p] = stage[src + p];
        }
    }
#ifdef MIPSF_RT_TRACE
    RT_MARK(9);
    for (int q = 2; q < 9; ++q) tr_acc[q] += tr_t[q + 1] - tr_t[q];
    ++tr_levels;
#endif
    }
#ifdef MIPSF_RT_TRACE
    if ((threadIdx.x & 63u) == 0u) {
        unsigned long long* tr = reinterpret_cast<unsigned long long*>(ws + plan.w_end + 64) + (size_t)(blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6)) * 10;
        for (int q = 0; q < 9; ++q) tr[q] = tr_acc[q];
        tr[9] = (unsigned long long)(group | (tr_levels << 8));
    }
#endif
}
__device__ __forceinline__ uint32_t part_of(const ScatterPlan& plan, uint32_t bin) {
    return bin < plan.dense_bins ? SC_PART_DENSE : SC_PART;
}

// One work item = (bin, part of its records): the slice's gradient is accumulated in LDS (fp64) and added to dparams (the
// bin's only part) or left as a partial slice for the reduce kernel.  desc = {bin | part << 16, records of the bin, parts of
// the bin}, item_index = position in the item order (names the partial slice).
template <int LAYOUT>
__device__ __forceinline__ void scatter_item(const float* __restrict__ x, const float* __restrict__ dout,
                                             float* __restrict__ dparams, uint32_t M, const GridLevels& g,
                                             const ScatterPlan& plan, uint32_t* __restrict__ ws, double* __restrict__ acc,
                                             const uint4 desc, const uint32_t item_index) {
    const uint32_t item = desc.x;
    const uint32_t sc_plane1 = plan.max_slice;
    (void)sc_plane1;
    const uint32_t bin = item & 0xffffu, part = item >> 16;
    const bool single = desc.z == 1u;             // the bin's only work item: its slice goes straight into dparams
    uint32_t level = 0;
    while (level + 1 < plan.n_levels && bin >= plan.bin0[level + 1]) ++level;
    const uint32_t slice = bin - plan.bin0[level];
    const uint32_t off = g.offsets[level];
    const uint32_t size = g.offsets[level + 1] - off;
    const uint32_t se = plan.slice_entries[level];
    const uint32_t begin = slice * se;
    const uint32_t count = begin + se <= size ? se : size - begin;
    const uint32_t res = g.res[level];
    const float scale = g.scale[level];
    const uint32_t n_rec_bin = desc.y;
    const uint32_t r0 = part * part_of(plan, bin);
    const uint32_t n_rec = n_rec_bin - r0 < part_of(plan, bin) ? n_rec_bin - r0 : part_of(plan, bin);
    const uint32_t* __restrict__ rec = ws + plan.w_records + (size_t)bin * M + r0;

    // ... and the current gradient values of the slice are requested now: by the time the records are through they have
    // arrived (they used to cost a memory round trip behind the last barrier, 3 us per item)
    {   // the slice is cleared (only the entries of this level's slice: the fixed cost of an item used to include all 160 KB)
#if MIPSF_SC_PLANAR
        for (uint32_t e = threadIdx.x; e < count; e += SC_BLOCK) acc[SC_ACC0(e)] = 0.0, acc[SC_ACC1(e)] = 0.0;
#else
        double2* z = reinterpret_cast<double2*>(acc);
        for (uint32_t e = threadIdx.x; e < count; e += SC_BLOCK) z[e] = make_double2(0.0, 0.0);
#endif
    }
    float2* dst = reinterpret_cast<float2*>(dparams) + off + begin;
    constexpr uint32_t FL = SC_MAX_SLICE / SC_BLOCK;
    float2 cur[FL];
    if (single && !plan.fresh) {
#pragma unroll
        for (uint32_t k = 0; k < FL; ++k) {
            const uint32_t e = threadIdx.x + k * SC_BLOCK;
            cur[k] = e < count ? dst[e] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();

    const int mode = level_mode(res, size);
    const bool masked = M <= SC_MASKED_MAX_M;      // records carry the corner mask of scatter_route_kernel
    // Every thread takes SC_RUN consecutive records at a time (records are in sample order, so they are a
    // stretch of one ray) and merges the ones that sit in the same cell in registers; the atomics go out when the
    // cell changes.  On the coarse levels a ray spends ~10 samples per cell: up to SC_RUN x fewer atomics, and the
    // lanes of a wave mostly hold different cells -- same-address ds_add_f64 degrades from 0.33 to 3 cycles per
    // lane-op (tools/micro/lds_atomic.hip).  A wave still reads 64 x SC_RUN consecutive records (coalesced).
    uint32_t cc[3] = {0u, 0u, 0u};
    float sum[8][2];
    bool open = false;
    auto flush = [&]() {
#ifdef MIPSF_SC_ABL_DENSE_NOATOM      // ablation: a dense item without its LDS atomics (wrong results)
        if (mode == 0) { asm volatile("" :: "v"(sum[0][0]), "v"(sum[7][1])); return; }
#endif
        Cell cell;
        cell.c[0] = cc[0], cell.c[1] = cc[1], cell.c[2] = cc[2];
        uint32_t idx[8];
        corner_indices(mode, cell, res, size, idx);
        uint32_t hit = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            idx[c] -= begin;
            hit |= (idx[c] < count) ? (1u << c) : 0u;
        }
        // each lane walks ITS OWN hit list, so the wave issues max-over-lanes(hits) dense atomic pairs instead of 8
        // sparse ones (an LDS atomic instruction has a fixed cost of ~20 cycles however few lanes are active)
        if (mode == 0) {
            // dense level: a cell is inside the slice with all 8 corners or (almost always) with none, so the
            // straight-line form has full lanes and needs no per-lane selects
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (hit >> c & 1u) {
                    atomicAdd(&acc[SC_ACC0(idx[c])], (double)sum[c][0]);
                    atomicAdd(&acc[SC_ACC1(idx[c])], (double)sum[c][1]);
                }
            }
            return;
        }
        while (hit) {
            const int c = __ffs((int)hit) - 1;
            hit &= hit - 1u;
            uint32_t e = idx[0];
            float v0 = sum[0][0], v1 = sum[0][1];
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                e = (c == k) ? idx[k] : e;
                v0 = (c == k) ? sum[k][0] : v0;
                v1 = (c == k) ? sum[k][1] : v1;
            }
            atomicAdd(&acc[SC_ACC0(e)], (double)v0);
            atomicAdd(&acc[SC_ACC1(e)], (double)v1);
        }
    };
    if (mode != 0) {
        // Hashed (fine) levels: a record has ~2 of its 8 corners in this slice and consecutive samples seldom share a
        // cell, so run merging would only add work.  Direct form: membership first, weights and atomics for the hits
        // only (each lane walks its own hit list).
        constexpr int UD = 4;
        for (uint32_t r = threadIdx.x; r < n_rec; r += UD * SC_BLOCK) {
            uint32_t si[UD];
            float px[UD][3];
            float2 pg[UD];
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                const uint32_t q = r + u * SC_BLOCK;
                si[u] = rec[q < n_rec ? q : n_rec - 1];
            }
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                const uint32_t ii = masked ? (si[u] & 0xffffffu) : si[u];
#ifdef MIPSF_SC_ABL_NOGATHER     // ablation: what a hashed item costs WITHOUT its two per-record gathers (wrong results)
                px[u][0] = (float)(ii & 1023u) * (1.0f / 1024.0f), px[u][1] = (float)((ii >> 10) & 1023u) * (1.0f / 1024.0f);
                px[u][2] = (float)((ii >> 5) & 1023u) * (1.0f / 1024.0f);
                pg[u] = make_float2(1e-3f, 2e-3f);
#else
                // (non-temporal loads for these two gathers: 109 -> 150 us)
                px[u][0] = x[3 * (size_t)ii], px[u][1] = x[3 * (size_t)ii + 1], px[u][2] = x[3 * (size_t)ii + 2];
                pg[u] = *reinterpret_cast<const float2*>(dout + feat_index<LAYOUT>(ii, level, M, g.n_levels));
#endif
            }
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                if (r + u * SC_BLOCK >= n_rec) break;
                Cell cell;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float pos = fmaf(scale, px[u][d], 0.5f);
                    const float fl = floorf(pos);
                    cell.c[d] = (uint32_t)(int)fl;
                    cell.f[d] = pos - fl;
                }
                uint32_t hit = si[u] >> 24;           // the routing kernel's corner mask
                if (!masked) {                        // (batches beyond 2^24 samples: membership of all 8 corners here)
                    uint32_t idx[8];
                    corner_indices(mode, cell, res, size, idx);
                    hit = 0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) hit |= (idx[c] - begin < count) ? (1u << c) : 0u;
                }
                const float2 gy = pg[u];
                if (mode == 1) {
                    // power-of-two table: the corners come as x-pairs that share (cy P1) ^ (cz P2) and the y, z weights,
                    // and the routing kernel's mask almost always holds whole pairs: one trip per PAIR (usually one,
                    // at most four per record) instead of one per corner
                    uint32_t pm = (hit | (hit >> 1)) & 0x55u;
                    while (pm) {
                        const int c = __ffs((int)pm) - 1;               // the pair's x corner: 0, 2, 4, 6
                        pm &= pm - 1u;
                        const uint32_t hz = ((cell.c[1] + ((c >> 1) & 1)) * P1) ^ ((cell.c[2] + ((c >> 2) & 1)) * P2);
                        const float wy = (c & 2) ? cell.f[1] : 1.0f - cell.f[1], wz = (c & 4) ? cell.f[2] : 1.0f - cell.f[2];
                        if (hit >> c & 1u) {
                            const uint32_t e = ((cell.c[0] ^ hz) & (size - 1u)) - begin;
                            const float wgt = ((1.0f - cell.f[0]) * wy) * wz;       // (the order of corner_weights)
                            atomicAdd(&acc[SC_ACC0(e)], (double)(wgt * gy.x));
                            atomicAdd(&acc[SC_ACC1(e)], (double)(wgt * gy.y));
                        }
                        if (hit >> (c + 1) & 1u) {
                            const uint32_t e = (((cell.c[0] + 1u) ^ hz) & (size - 1u)) - begin;
                            const float wgt = (cell.f[0] * wy) * wz;
                            atomicAdd(&acc[SC_ACC0(e)], (double)(wgt * gy.x));
                            atomicAdd(&acc[SC_ACC1(e)], (double)(wgt * gy.y));
                        }
                    }
                    continue;
                }
                while (hit) {
                    const int c = __ffs((int)hit) - 1;
                    hit &= hit - 1u;
                    const uint32_t cx = cell.c[0] + (c & 1), cy = cell.c[1] + ((c >> 1) & 1), cz = cell.c[2] + ((c >> 2) & 1);
                    const uint32_t e = (mode == 1 ? corner_index<1>(cx, cy, cz, res, size) : corner_index<2>(cx, cy, cz, res, size)) - begin;
                    float wgt = (c & 1) ? cell.f[0] : 1.0f - cell.f[0];
                    wgt = wgt * ((c & 2) ? cell.f[1] : 1.0f - cell.f[1]);
                    wgt = wgt * ((c & 4) ? cell.f[2] : 1.0f - cell.f[2]);
                    atomicAdd(&acc[SC_ACC0(e)], (double)(wgt * gy.x));
                    atomicAdd(&acc[SC_ACC1(e)], (double)(wgt * gy.y));
                }
            }
        }
    } else {
    constexpr int UNR = (int)SC_RUN;   // independent record -> x chains in flight per thread
    // (a transposed assignment -- neighbouring lanes one ray apart, so that they never meet in a coarse cell -- was
    // measured 25 % slower: the locality of the record / x / dL/dy reads matters more than the residual conflicts)
    const uint32_t chunk_id = threadIdx.x;
    for (uint32_t r = chunk_id * SC_RUN; r < n_rec; r += SC_BLOCK * SC_RUN) {
        const uint32_t re = r + SC_RUN < n_rec ? r + SC_RUN : n_rec;
        uint32_t si[UNR];
        float px[UNR][3];
        float2 pg[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            si[u] = rec[r + u < re ? r + u : re - 1];
            si[u] = masked ? (si[u] & 0xffffffu) : si[u];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t ii = si[u];
#ifdef MIPSF_SC_ABL_DENSE_NOGATHER    // ablation: a dense item without its x / d feat loads (wrong results)
            px[u][0] = (float)(ii & 4095u) * (1.0f / 4096.0f), px[u][1] = (float)((ii >> 6) & 4095u) * (1.0f / 4096.0f);
            px[u][2] = (float)((ii >> 12) & 63u) * (1.0f / 64.0f);
            pg[u] = make_float2(1e-3f, 2e-3f);
#else
            px[u][0] = x[3 * (size_t)ii], px[u][1] = x[3 * (size_t)ii + 1], px[u][2] = x[3 * (size_t)ii + 2];
            pg[u] = *reinterpret_cast<const float2*>(dout + feat_index<LAYOUT>(ii, level, M, g.n_levels));
#endif
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (r + u >= re) break;
            Cell cell;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float pos = fmaf(scale, px[u][d], 0.5f);
                const float fl = floorf(pos);
                cell.c[d] = (uint32_t)(int)fl;
                cell.f[d] = pos - fl;
            }
            const bool same = open && cell.c[0] == cc[0] && cell.c[1] == cc[1] && cell.c[2] == cc[2];
            if (!same) {
#ifndef MIPSF_SC_ABL_DENSE_NOMID      // ablation: the flushes at a cell boundary inside a thread's records dropped (wrong results)
                if (open) flush();
#endif
                cc[0] = cell.c[0], cc[1] = cell.c[1], cc[2] = cell.c[2];
#pragma unroll
                for (int c = 0; c < 8; ++c) sum[c][0] = 0.0f, sum[c][1] = 0.0f;
                open = true;
            }
            const float2 gy = pg[u];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float wgt = (c & 1) ? cell.f[0] : 1.0f - cell.f[0];
                wgt = wgt * ((c & 2) ? cell.f[1] : 1.0f - cell.f[1]);
                wgt = wgt * ((c & 4) ? cell.f[2] : 1.0f - cell.f[2]);
                sum[c][0] = sum[c][0] + wgt * gy.x;
                sum[c][1] = sum[c][1] + wgt * gy.y;
            }
        }
        // The run still open at the end of the thread's records.  (Adding up the open runs of neighbouring lanes that end in
        // the same cell first -- a segmented sum through the wave, so that only one lane of a stretch goes to LDS -- was measured
        // twice in round 5, with ds_bpermute and with DPP row shifts: no change)
        if (open) flush();
        open = false;
    }
    }
    __syncthreads();

    auto a2 = [&](uint32_t e) { return make_double2(acc[SC_ACC0(e)], acc[SC_ACC1(e)]); };
    if (single) {
#pragma unroll
        for (uint32_t k = 0; k < FL; ++k) {
            const uint32_t e = threadIdx.x + k * SC_BLOCK;
            if (e < count) {
                const double2 a = a2(e);
                dst[e] = plan.fresh ? make_float2((float)a.x, (float)a.y) : make_float2(cur[k].x + (float)a.x, cur[k].y + (float)a.y);
            }
        }
    } else {
        float2* pdst = reinterpret_cast<float2*>(reinterpret_cast<float*>(ws) + plan.w_partial) +
                       (size_t)item_index * SC_MAX_SLICE;
        for (uint32_t e = threadIdx.x; e < count; e += SC_BLOCK) {
            const double2 a = a2(e);
            pdst[e] = make_float2((float)a.x, (float)a.y);
        }
    }
}


// PERSISTENT accumulate kernel: one workgroup per CU (the fp64 slice fills most of a CU's LDS) that
//   (1) builds the work-item order from the bin counts itself -- every workgroup runs the same 1024-thread prefix scan and
//       keeps its share of it in registers (thread t: the counts of bins 8 t .. 8 t + 7 and the index of their first item);
//       this used to be a one-workgroup kernel of its own (7 us + a launch gap) in front of a grid of max_items workgroups,
//       most of which found an "unused" mark, cleared 160 KB of LDS for nothing and left;
//   (2) pulls items from ONE device-scope counter (dense levels first: the long items lead, as in LPT scheduling) until the
//       order is exhausted -- no static assignment of items to workgroups, the tail is as short as the last item.
// cw = the small counter block: {bin counts | head, ticket A, ticket B, - | first item of every bin | parts of every bin}.
// On entry the counts are the routing kernel's and head / tickets are zero; on exit everything but first / parts is zero
// again (those two are rewritten for EVERY bin by every call), so a caller that keeps the block between calls never
// launches a clearing kernel.  Nobody waits for a ticket: the LAST workgroup through a point does the clearing --
//   ticket A (after the prologue: every workgroup has read the counts)   -> the counts are cleared,
//   ticket B (after the loop: every workgroup has made its last pull)    -> head and both tickets are cleared.
template <int LAYOUT>
__global__ __launch_bounds__(SC_BLOCK) void hashgrid_scatter_persist_kernel(const float* __restrict__ x,
                                                                           const float* __restrict__ dout,
                                                                           float* __restrict__ dparams, uint32_t M,
                                                                           GridLevels g, ScatterPlan plan,
                                                                           uint32_t* __restrict__ ws,
                                                                           uint32_t* __restrict__ cw) {
    extern __shared__ __attribute__((aligned(16))) double acc[];   // [slice entries][2], then the bin counts
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t sh_total, sh_next;
    __shared__ uint4 sh_desc;
    constexpr uint32_t PER = SC_MAX_BINS / SC_BLOCK;
    const uint32_t t = threadIdx.x;
    // the counts of this thread's bins stay in LDS behind the slice (in registers they cost 8 VGPRs of the 128 a
    // 1024-thread workgroup has: the item loop spilled)
    uint32_t* scnt = reinterpret_cast<uint32_t*>(acc + 2 * (size_t)plan.max_slice);
    uint32_t litm = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t b = t * PER + k;
        const uint32_t c = b < plan.n_bins ? cw[plan.w_count + b] : 0u;
        if (b < plan.n_bins) scnt[b] = c;
        litm += (c + part_of(plan, b) - 1) / part_of(plan, b);
    }
    uint32_t oitm;                               // index of the first item of this thread's bins
    {
        uint32_t incl = litm;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
            incl += (int)(t & 63u) >= d ? v : 0u;
        }
        if ((t & 63u) == 63u) wtot[t >> 6] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t q = 0; q < (t >> 6); ++q) before += wtot[q];
        oitm = before + incl - litm;
        if (t == SC_BLOCK - 1) sh_total = before + incl;
    }
    {   // where every bin's partial slices start, for the reduce kernel (all workgroups write the same values)
        uint32_t first = oitm;
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t b = t * PER + k;
            if (b >= plan.n_bins) break;
            const uint32_t parts = (scnt[b] + part_of(plan, b) - 1) / part_of(plan, b);
            cw[plan.w_first + b] = first, cw[plan.w_parts + b] = parts;
            first += parts;
        }
    }
    __syncthreads();                             // (sh_total; and every count of this workgroup has been read and used)
    const uint32_t total = sh_total;
    if (t == 0) sh_next = __hip_atomic_fetch_add(&cw[plan.w_nitems + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh_next == gridDim.x - 1) {              // ticket A: the last workgroup past the prologue clears the counts
        for (uint32_t b = t; b < plan.n_bins; b += SC_BLOCK) cw[plan.w_count + b] = 0u;
    }
    for (;;) {
        __syncthreads();                         // (sh_next / sh_desc / the slice are free again)
        if (t == 0) sh_next = __hip_atomic_fetch_add(&cw[plan.w_nitems], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t i = sh_next;
        if (i >= total) break;
        if (i >= oitm && i < oitm + litm) {      // exactly one thread owns item i: it names the bin and the part
            uint32_t first = oitm;
            for (uint32_t k = 0; k < PER; ++k) {
                const uint32_t b = t * PER + k;
                if (b >= plan.n_bins) break;
                const uint32_t c = scnt[b];
                const uint32_t parts = (c + part_of(plan, b) - 1) / part_of(plan, b);
                if (i >= first && i < first + parts) sh_desc = make_uint4(b | ((i - first) << 16), c, parts, 0u);
                first += parts;
            }
        }
        __syncthreads();
        const uint4 desc = sh_desc;
#ifdef MIPSF_SC_TRACE    // tools/probe_scatter_trace.py: one row per item {item, bin | part << 16, records, begin, end (10 ns ticks), XCC, workgroup}
        const uint64_t tr_t0 = wall_clock64();
#endif
        scatter_item<LAYOUT>(x, dout, dparams, M, g, plan, ws, acc, desc, i);
#ifdef MIPSF_SC_TRACE
        __syncthreads();
        if (t == 0) {
            const uint64_t tr_t1 = wall_clock64();
            uint32_t* tr = ws + plan.w_end + 64 + 8 * (size_t)i;
            tr[0] = i, tr[1] = desc.x, tr[2] = desc.y, tr[3] = (uint32_t)tr_t0, tr[4] = (uint32_t)tr_t1;
            tr[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20), tr[6] = blockIdx.x, tr[7] = desc.z;
            if (i == 0) ws[plan.w_end + 63] = total;
        }
#endif
    }
    __syncthreads();
    if (t == 0) {                                // ticket B: the last workgroup out resets the queue for the next call
        if (__hip_atomic_fetch_add(&cw[plan.w_nitems + 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(&cw[plan.w_nitems], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cw[plan.w_nitems + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cw[plan.w_nitems + 2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// folds the partial slices of bins that were split over several workgroups into dparams
__global__ __launch_bounds__(256) void hashgrid_scatter_reduce_kernel(float* __restrict__ dparams, GridLevels g,
                                                                      ScatterPlan plan, const uint32_t* __restrict__ ws,
                                                                      const uint32_t* __restrict__ cw) {
    const uint32_t bin = blockIdx.x;
    const uint32_t parts = cw[plan.w_parts + bin];
    if (parts <= 1) return;
    uint32_t level = 0;
    while (level + 1 < plan.n_levels && bin >= plan.bin0[level + 1]) ++level;
    const uint32_t size = g.offsets[level + 1] - g.offsets[level];
    const uint32_t se = plan.slice_entries[level];
    const uint32_t begin = (bin - plan.bin0[level]) * se;
    const uint32_t count = begin + se <= size ? se : size - begin;
    const uint32_t e = blockIdx.y * 256 + threadIdx.x;
    if (e >= count) return;
    const float2* p2 = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(ws) + plan.w_partial) +
                       (size_t)cw[plan.w_first + bin] * SC_MAX_SLICE + e;
    float2* d = reinterpret_cast<float2*>(dparams) + g.offsets[level] + begin + e;
    float2 cur = plan.fresh ? make_float2(0.f, 0.f) : *d;
    // eight partial slices in flight (a load per iteration was one memory round trip per part: 11 parts on the coarsest
    // levels, 10 us for 10 MB); the sum keeps the order part 0, 1, 2, ...
    float2 a = make_float2(0.f, 0.f);
    for (uint32_t q = 0; q < parts; q += 8) {
        float2 v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) v[k] = q + k < parts ? p2[(size_t)(q + k) * SC_MAX_SLICE] : make_float2(0.f, 0.f);
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) a.x += v[k].x, a.y += v[k].y;
    }
    cur.x += a.x, cur.y += a.y;
    *d = cur;
}

// dL/dx of one level per thread (tcnn kernel_grid_backward_input); written to per-level partials, no atomics
template <int LAYOUT>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(HG_BLOCK) void hashgrid_dx_kernel(const float* __restrict__ x,
                                                               const float2* __restrict__ table,
                                                               const float* __restrict__ dout,
                                                               float* __restrict__ dxl, uint32_t M, GridLevels g,
                                                               uint32_t nchunk) {
    uint32_t chunk;
    const int level = decode_level(g.n_levels, nchunk, chunk);
    if (level >= (int)g.n_levels) return;
    const uint32_t i = chunk * HG_BLOCK + threadIdx.x;
    if (i >= M) return;
    const uint32_t off = g.offsets[level];
    const uint32_t size = g.offsets[level + 1] - off;
    const uint32_t res = g.res[level];
    const float scale = g.scale[level];
    const Cell cell = locate(x, i, scale);
    uint32_t idx[8];
    corner_indices(level_mode(res, size), cell, res, size, idx);
    const float2 gy = *reinterpret_cast<const float2*>(dout + feat_index<LAYOUT>(i, level, M, g.n_levels));
    const float2* lvl = table + off;
    float2 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = lvl[idx[c]];
    float gx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float s0, s1;
        level_jacobian(cell, v, scale, d, s0, s1);
        gx[d] = s0 * gy.x + s1 * gy.y;
    }
    float* o = dxl + ((size_t)level * M + i) * 3;
    o[0] = gx[0], o[1] = gx[1], o[2] = gx[2];
}

__global__ __launch_bounds__(256) void hashgrid_dx_reduce_kernel(const float* __restrict__ dxl, float* __restrict__ dx,
                                                                 uint64_t n3, uint32_t L) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    float a = 0.f;
    for (uint32_t l = 0; l < L; ++l) a += dxl[(uint64_t)l * n3 + t];
    dx[t] += a;
}

// dx += sum over levels of J_l . dL/dy_l with the Jacobian saved by the forward (same arithmetic and the same
// level order as hashgrid_dx_kernel + hashgrid_dx_reduce_kernel: bit-identical results)
// tiles (optional): the live-tile lists of the decoder's backward chain (decoder16.hip / decoder_layout.h).  Thread t then
// works on sample 32 * tile(t / 32) + t % 32 of the listed tiles only: the others have a zero feature gradient, add
// nothing, and their 384 bytes of Jacobian per sample stay unread.
template <int LAYOUT>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(256) void hashgrid_dx_jac_kernel(const float* __restrict__ jac,
                                                              const float* __restrict__ dout, float* __restrict__ dx,
                                                              uint32_t M, uint32_t L,
                                                              const uint32_t* __restrict__ tiles) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (tiles != nullptr) {
        const uint32_t k = i >> 5, cap = mipsf::dl::tl_cap((M + 31u) / 32u);
        uint32_t q = 0, first = 0, total = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t n = tiles[64 * j + 32];
            if (k >= total && n != 0u) q = j, first = total;
            total += n;
        }
        if (k >= total) return;
        i = tiles[mipsf::dl::TL_HEADER + q * cap + (k - first)] * 32u + (i & 31u);
    }
    if (i >= M) return;
    float a[3] = {0.f, 0.f, 0.f};
    for (uint32_t l = 0; l < L; ++l) {
        const float2* j2 = reinterpret_cast<const float2*>(jac) + (size_t)l * 3 * M + i;      // [L][3][M][2]
        const float2 gy = *reinterpret_cast<const float2*>(dout + feat_index<LAYOUT>(i, l, M, L));
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float2 j = j2[(size_t)d * M];
            a[d] += j.x * gy.x + j.y * gy.y;
        }
    }
    dx[3 * (size_t)i] += a[0], dx[3 * (size_t)i + 1] += a[1], dx[3 * (size_t)i + 2] += a[2];
}

__global__ __launch_bounds__(HG_BLOCK) void hashgrid_indices_kernel(const float* __restrict__ x,
                                                                    uint32_t* __restrict__ out, uint32_t M,
                                                                    GridLevels g, uint32_t nchunk) {
    uint32_t chunk;
    const int level = decode_level(g.n_levels, nchunk, chunk);
    if (level >= (int)g.n_levels) return;
    const uint32_t i = chunk * HG_BLOCK + threadIdx.x;
    if (i >= M) return;
    const uint32_t size = g.offsets[level + 1] - g.offsets[level];
    const uint32_t res = g.res[level];
    const Cell cell = locate(x, i, g.scale[level]);
    uint32_t idx[8];
    corner_indices(level_mode(res, size), cell, res, size, idx);
#pragma unroll
    for (int c = 0; c < 8; ++c) out[((size_t)i * g.n_levels + level) * 8 + c] = idx[c];
}

static int to_levels(const mipsf_grid_meta* m, GridLevels& g) {
    MIPSF_REQUIRE(m != nullptr, "meta is null");
    MIPSF_REQUIRE(m->n_features == 2, "only n_features_per_level == 2 is built");
    MIPSF_REQUIRE(m->n_levels >= 1 && m->n_levels <= MIPSF_MAX_LEVELS, "bad n_levels");
    g.n_levels = m->n_levels;
    for (uint32_t l = 0; l <= m->n_levels; ++l) g.offsets[l] = m->offsets[l];
    for (uint32_t l = 0; l < m->n_levels; ++l) {
        g.res[l] = m->resolutions[l];
        g.scale[l] = m->scales[l];
    }
    return 0;
}

static inline uint32_t grid_blocks(uint32_t M, uint32_t L, uint32_t& nchunk) {
    nchunk = (M + HG_BLOCK - 1) / HG_BLOCK;
    return 8u * ((L + 7u) / 8u) * nchunk;
}

}  // namespace mipsf

using namespace mipsf;

extern "C" {

static int hashgrid_fwd_impl(const float* x, const float* params, float* out, float* jac, uint32_t M,
                             const mipsf_grid_meta* meta, int layout, void* stream) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && params && out, "null pointer");
    MIPSF_REQUIRE(layout == MIPSF_FEAT_AOS || layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout %d", layout);
    uint32_t nchunk;
    const uint32_t nb = grid_blocks(M, g.n_levels, nchunk);
    hipStream_t s = (hipStream_t)stream;
    const float2* table = reinterpret_cast<const float2*>(params);
#define FWD(LAY, J) hipLaunchKernelGGL((hashgrid_fwd_kernel<LAY, J>), dim3(nb), dim3(HG_BLOCK), 0, s, x, table, out, jac, M, g, nchunk)
    if (layout == MIPSF_FEAT_AOS) { if (jac) FWD(MIPSF_FEAT_AOS, true); else FWD(MIPSF_FEAT_AOS, false); }
    else { if (jac) FWD(MIPSF_FEAT_LEVEL_MAJOR, true); else FWD(MIPSF_FEAT_LEVEL_MAJOR, false); }
#undef FWD
    return check_launch("hashgrid_fwd");
}

int mipsf_hashgrid_fwd(const float* x, const float* params, float* out, float* jac, uint32_t M,
                       const mipsf_grid_meta* meta, int layout, void* stream) {
    return hashgrid_fwd_impl(x, params, out, jac, M, meta, layout, stream);
}

int mipsf_hashgrid_dx_from_jac(const float* jac, const float* dout, float* dx, const uint32_t* tile_live, uint32_t M,
                               const mipsf_grid_meta* meta, int layout, void* stream) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(jac && dout && dx, "null pointer");
    MIPSF_REQUIRE(layout == MIPSF_FEAT_AOS || layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout %d", layout);
    hipStream_t s = (hipStream_t)stream;
    if (layout == MIPSF_FEAT_AOS)
        hipLaunchKernelGGL(hashgrid_dx_jac_kernel<MIPSF_FEAT_AOS>, dim3((M + 255) / 256), dim3(256), 0, s, jac, dout, dx, M, g.n_levels, tile_live);
    else
        hipLaunchKernelGGL(hashgrid_dx_jac_kernel<MIPSF_FEAT_LEVEL_MAJOR>, dim3((M + 255) / 256), dim3(256), 0, s, jac, dout, dx, M, g.n_levels, tile_live);
    return check_launch("hashgrid_dx_from_jac");
}

}  // extern "C"
namespace mipsf {
uint64_t hashgrid_bwd_scratch_floats(const mipsf_grid_meta* meta, uint32_t M, int need_dx) {
    GridLevels g;
    if (to_levels(meta, g)) return 0;
    const ScatterPlan p = make_plan(g, M);
#ifdef MIPSF_SC_TRACE
    return p.w_end + 64 + 8ull * p.max_items + 64;
#endif
#ifdef MIPSF_RT_TRACE
    return p.w_end + 64 + 20ull * (RT_BLOCK / 64) * (route_groups(g.n_levels) * ((M + RT_BLOCK * SC_ROUTE_UNR - 1) / (RT_BLOCK * SC_ROUTE_UNR))) + 64;
#endif
    return p.w_end + (need_dx ? (uint64_t)g.n_levels * M * 3 : 0) + 64;
}
uint64_t hashgrid_counter_words(const mipsf_grid_meta* meta) {
    GridLevels g;
    if (to_levels(meta, g)) return 0;
    return make_plan(g, 1).w_items;              // {counts | head, tickets | first | parts}: independent of the batch size
}
}  // namespace mipsf
extern "C" {

#ifdef MIPSF_SC_TRACE
// word offset of the trace rows, their capacity and the first bin of every level (diagnosis builds only)
uint64_t mipsf_hashgrid_trace_words(const mipsf_grid_meta* meta, uint32_t M, uint32_t* n_rows, uint32_t* bin0) {
    GridLevels g;
    if (to_levels(meta, g)) return 0;
    const ScatterPlan p = make_plan(g, M);
    *n_rows = p.max_items;
    for (uint32_t l = 0; l <= g.n_levels; ++l) bin0[l] = p.bin0[l];
    return p.w_end + 64;
}
#endif


#ifdef MIPSF_RT_TRACE
// word offset of the routing kernel's trace rows (10 x 8 bytes per wave: 9 phase durations in cycles, the per-level phases summed
// over the workgroup's levels, + level group | levels << 8) and their number
uint64_t mipsf_hashgrid_rt_trace_words(const mipsf_grid_meta* meta, uint32_t M, uint32_t* n_rows) {
    GridLevels g;
    if (to_levels(meta, g)) return 0;
    const ScatterPlan p = make_plan(g, M);
    *n_rows = (RT_BLOCK / 64) * route_groups(g.n_levels) * ((M + RT_BLOCK * SC_ROUTE_UNR - 1) / (RT_BLOCK * SC_ROUTE_UNR));
    return p.w_end + 64;
}
#endif

static int check_plan(const ScatterPlan& plan, const GridLevels& g, uint32_t M) {
    MIPSF_REQUIRE(plan.n_bins <= SC_MAX_BINS && plan.n_bins <= 0xffffu, "grid too large: %u table slices", plan.n_bins);
    for (uint32_t l = 0; l < g.n_levels; ++l)
        MIPSF_REQUIRE(plan.n_slices[l] <= SC_MAX_NS, "level %u too large: %u slices", l, plan.n_slices[l]);
    MIPSF_REQUIRE(plan.w_end < (1ull << 32), "batch too large for 32-bit scratch offsets (M = %u)", M);
    MIPSF_REQUIRE(((uint64_t)M + SC_PART_MIN - 1) / SC_PART_MIN < (1u << 16), "batch too large (M = %u)", M);
    return 0;
}

// cw: the counter block; `clear`: it lives in caller scratch of unknown content (one more launch), else the caller keeps
// it between calls and every call leaves it ready (hashgrid_scatter_persist_kernel)
static int launch_route(const float* x, const float* dout, int layout, uint32_t* ws, uint32_t* cw, bool clear, uint32_t M,
                        const GridLevels& g, const ScatterPlan& plan, hipStream_t s) {
    if (clear) {
        const uint32_t nz = plan.w_nitems + 4;   // bin counts, queue head, tickets
        hipLaunchKernelGGL(scatter_zero_kernel, dim3((nz + 255) / 256), dim3(256), 0, s, cw, nz);
    }
    const uint32_t rb = route_groups(g.n_levels) * ((M + RT_BLOCK * SC_ROUTE_UNR - 1) / (RT_BLOCK * SC_ROUTE_UNR));
    if (layout == MIPSF_FEAT_AOS)
        hipLaunchKernelGGL(scatter_route_kernel<MIPSF_FEAT_AOS>, dim3(rb), dim3(RT_BLOCK), 0, s, x, dout, M, g, plan, ws, cw);
    else
        hipLaunchKernelGGL(scatter_route_kernel<MIPSF_FEAT_LEVEL_MAJOR>, dim3(rb), dim3(RT_BLOCK), 0, s, x, dout, M, g, plan, ws, cw);
    return check_launch("hashgrid_route");
}

// The routing half of the parameter-gradient scatter depends on the sample positions only: a caller may run it as soon
// as x exists (e.g. on a second stream next to the forward pass) and hand the scratch buffer to mipsf_hashgrid_bwd with MIPSF_HG_ROUTED.
int mipsf_hashgrid_route(const float* x, float* scratch, uint32_t M, const mipsf_grid_meta* meta, void* stream) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && scratch, "null pointer");
    const ScatterPlan plan = make_plan(g, M);
    if (int rc = check_plan(plan, g, M)) return rc;
    uint32_t* ws = reinterpret_cast<uint32_t*>(scratch);
    return launch_route(x, nullptr, MIPSF_FEAT_AOS, ws, ws, true, M, g, plan, (hipStream_t)stream);
}

static int hashgrid_bwd_impl(const float* x, const float* params, const float* dout, float* dparams, float* dx,
                             float* scratch, uint32_t* counters, uint32_t M, const mipsf_grid_meta* meta, int layout,
                             bool routed, void* stream, bool fresh = false) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && params && dout && scratch, "null pointer");
    MIPSF_REQUIRE(dparams || dx, "nothing to compute: dparams and dx are both null");
    MIPSF_REQUIRE(layout == MIPSF_FEAT_AOS || layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout %d", layout);
    hipStream_t s = (hipStream_t)stream;
    ScatterPlan plan = make_plan(g, M);
    plan.fresh = fresh ? 1u : 0u;
    if (int rc = check_plan(plan, g, M)) return rc;
    uint32_t* ws = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* cw = counters ? counters : ws;     // (without a kept block the counters sit at the front of the scratch)
    float* dxl = scratch + ((plan.w_end + 15) / 16) * 16;
    if (dparams) {   // a frozen grid (tracking) skips the scatter altogether
        if (!routed)
            if (int e = launch_route(x, MIPSF_SC_SKIP_ZERO ? dout : nullptr, layout, ws, cw, counters == nullptr, M, g, plan, s)) return e;
        const uint32_t lds_bytes = plan.max_slice * 16 + plan.n_bins * 4;      // the slice + the bin counts
        const int cus = device_cus();
        if (cus <= 0) return 3;
        // as many workgroups per CU as their slices fit into its 160 KB of LDS (and 32 waves), never more than there can be items
        uint32_t per_cu = (160u * 1024u) / (lds_bytes + 256u);
        per_cu = per_cu < 1u ? 1u : (per_cu > 2048u / SC_BLOCK ? 2048u / SC_BLOCK : per_cu);
        const uint32_t want = (uint32_t)cus * per_cu;
        const uint32_t blocks = plan.max_items < want ? plan.max_items : want;
#define SCATTER(LAY)                                                                                             \
    do {                                                                                                         \
        static uint32_t attr_bytes_dev[MAX_DEVICES] = {0};                                                       \
        uint32_t& attr_bytes = attr_bytes_dev[device_slot()];                                                    \
        if (lds_bytes > attr_bytes) {                                                                            \
            if (hipFuncSetAttribute((const void*)hashgrid_scatter_persist_kernel<LAY>,                           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {      \
                set_error("cannot raise dynamic LDS to %u bytes", lds_bytes);                                    \
                return 4;                                                                                        \
            }                                                                                                    \
            attr_bytes = lds_bytes;                                                                              \
        }                                                                                                        \
        hipLaunchKernelGGL((hashgrid_scatter_persist_kernel<LAY>), dim3(blocks), dim3(SC_BLOCK), lds_bytes, s,   \
                           x, dout, dparams, M, g, plan, ws, cw);                                                \
    } while (0)
        if (layout == MIPSF_FEAT_AOS) SCATTER(MIPSF_FEAT_AOS); else SCATTER(MIPSF_FEAT_LEVEL_MAJOR);
#undef SCATTER
        if (int e = check_launch("hashgrid_scatter")) return e;
        hipLaunchKernelGGL(hashgrid_scatter_reduce_kernel, dim3(plan.n_bins, (plan.max_slice + 255) / 256), dim3(256), 0, s,
                           dparams, g, plan, ws, cw);
        if (int e = check_launch("hashgrid_scatter_reduce")) return e;
    }
    if (dx) {
        uint32_t nchunk;
        const uint32_t nb = grid_blocks(M, g.n_levels, nchunk);
        const float2* table = reinterpret_cast<const float2*>(params);
        if (layout == MIPSF_FEAT_AOS)
            hipLaunchKernelGGL(hashgrid_dx_kernel<MIPSF_FEAT_AOS>, dim3(nb), dim3(HG_BLOCK), 0, s, x, table, dout, dxl, M, g, nchunk);
        else
            hipLaunchKernelGGL(hashgrid_dx_kernel<MIPSF_FEAT_LEVEL_MAJOR>, dim3(nb), dim3(HG_BLOCK), 0, s, x, table, dout, dxl, M, g, nchunk);
        if (int e = check_launch("hashgrid_dx")) return e;
        const uint64_t n3 = (uint64_t)M * 3;
        hipLaunchKernelGGL(hashgrid_dx_reduce_kernel, dim3((uint32_t)((n3 + 255) / 256)), dim3(256), 0, s, dxl, dx, n3, g.n_levels);
        if (int e = check_launch("hashgrid_dx_reduce")) return e;
    }
    return 0;
}

// ONE entry point for the family (round 5): every option is a field of the argument block (include/mipsf.h); unset fields
// are zero / NULL.  MIPSF_HG_DPARAMS_ZERO = the caller vouches that dparams is all zero on entry (a gradient buffer the
// optimiser cleared, a fresh torch.zeros): the table slices are stored instead of read-modify-written -- 36 MB less to read
// on the headline table, and a work item of the accumulate kernel no longer waits for its slice's old values
int mipsf_hashgrid_bwd(const mipsf_hashgrid_bwd_args* a, void* stream) {
    MIPSF_REQUIRE(a != nullptr, "null argument block");
    MIPSF_REQUIRE(a->struct_size == sizeof(mipsf_hashgrid_bwd_args), "mipsf_hashgrid_bwd_args: struct_size %u, this library expects %u",
                  a->struct_size, (unsigned)sizeof(mipsf_hashgrid_bwd_args));
    MIPSF_REQUIRE((a->flags & ~(uint32_t)(MIPSF_HG_DPARAMS_ZERO | MIPSF_HG_ROUTED)) == 0u, "unknown flags 0x%x", a->flags);
    return hashgrid_bwd_impl(a->x, a->params, a->dout, a->dparams, a->dx, a->scratch, a->counters, a->M, a->meta, a->feat_layout,
                             (a->flags & MIPSF_HG_ROUTED) != 0u, stream, (a->flags & MIPSF_HG_DPARAMS_ZERO) != 0u);
}

int mipsf_hashgrid_indices(const float* x, uint32_t* idx, uint32_t M, const mipsf_grid_meta* meta, void* stream) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && idx, "null pointer");
    uint32_t nchunk;
    const uint32_t nb = grid_blocks(M, g.n_levels, nchunk);
    hipLaunchKernelGGL(hashgrid_indices_kernel, dim3(nb), dim3(HG_BLOCK), 0, (hipStream_t)stream, x, idx, M, g,
                       nchunk);
    return check_launch("hashgrid_indices");
}

}  // extern "C"
