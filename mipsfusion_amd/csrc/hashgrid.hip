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

// ------------------------------------------------------------------------------ forward
template <int LAYOUT>
__global__ __launch_bounds__(HG_BLOCK) void hashgrid_fwd_kernel(const float* __restrict__ x,
                                                                const float2* __restrict__ table,
                                                                float* __restrict__ out, uint32_t M,
                                                                GridLevels g, uint32_t nchunk) {
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
    float2* dst = reinterpret_cast<float2*>(out + feat_index<LAYOUT>(i, level, M, g.n_levels));
    __builtin_nontemporal_store(a0, &dst->x);
    __builtin_nontemporal_store(a1, &dst->y);
}

// ----------------------------------------------------------------------------- backward
// dL/dparams is a scatter of 16 floats per (sample, level).  Global fp32 atomics top out at ~18 G atomics/s on
// MI355X whatever the access pattern (measured: 4.0 ms for the 67 M atomics of a 4096x64 batch), so the scatter
// target is moved ON CHIP: the 256 CUs own 40 MiB of LDS, more than the whole 34.4 MiB gradient table.
//   * a level's table is cut into slices of <= 20480 entries (160 KiB of float2);
//   * one workgroup owns one slice: it scans the samples of that level, recomputes the 8 corner indices and
//     accumulates the contributions that fall into its slice with LDS atomics (ds_add_f32);
//   * it then adds its slice to dparams with coalesced read-modify-writes -- no global atomics, no inter-workgroup
//     communication, nothing placement dependent;
//   * coarse levels (few slices, every sample hits, long same-cell runs => LDS conflicts) are additionally split
//     over the SAMPLES; such workgroups write partial slices to scratch and a reduce kernel folds them in.
// Redundant index computation (each sample is visited by every slice owner of its level) is ALU work the chip has
// to spare; HBM traffic stays at x + dL/dy + one read-modify-write of the table.
constexpr int SC_BLOCK = 1024;
constexpr uint32_t SC_MAX_SLICE = 20480;            // entries: 20480 * 8 B = 163840 B = all of a CU's LDS
constexpr uint32_t SC_MAX_SPLIT = 32;

struct ScatterPlan {
    uint32_t n_levels;
    uint32_t first_block[MIPSF_MAX_LEVELS + 1];   // blocks are ordered finest level first: order[k] = level
    // NB: every array is uint32_t on purpose.  With sub-dword arrays in a kernel-argument struct hipcc (ROCm 7.2)
    // folds `base + 2*level` into the SBASE of a scalar dword load; the hardware ignores SBASE's low two bits, so
    // odd levels silently read element [level-1].
    uint32_t order[MIPSF_MAX_LEVELS];
    uint32_t n_slices[MIPSF_MAX_LEVELS];
    uint32_t n_split[MIPSF_MAX_LEVELS];
    uint32_t slice_entries[MIPSF_MAX_LEVELS];
    uint32_t partial_off[MIPSF_MAX_LEVELS];       // float offset into the partial scratch (levels with n_split > 1)
    uint32_t total_blocks;
    uint32_t partial_floats;
    uint32_t split_entries;                       // sum of level sizes over split levels (reduce kernel extent)
};

static ScatterPlan make_plan(const GridLevels& g, uint32_t M) {
    ScatterPlan p = {};
    p.n_levels = g.n_levels;
    uint32_t blocks = 0, poff = 0, split_entries = 0;
    for (uint32_t k = 0; k < g.n_levels; ++k) {
        const uint32_t l = g.n_levels - 1 - k;
        const uint32_t size = g.offsets[l + 1] - g.offsets[l];
        const uint32_t ns = (size + SC_MAX_SLICE - 1) / SC_MAX_SLICE;
        const uint32_t se = (size + ns - 1) / ns;
        // Measured cost of one workgroup scanning all M samples of a level (units of a hashed-level workgroup at
        // M = 262144, ~300 us): hashed levels 1.0 (hits are scattered, ~2.5 hit-loop trips per wave iteration);
        // dense levels 1 + 33/n_slices (a sample has all 8 corners in the slice or none: 8 sparse trips, plus
        // same-address conflicts on coherent rays).  Split the samples so that no workgroup exceeds ~1 unit.
        const bool dense = !level_is_hashed(g.res[l], size);
        const double unit = (double)M / 262144.0 * (dense ? 1.0 + 33.0 / ns : 1.0);
        uint32_t split = (uint32_t)(unit + 0.999);
        if (split < 1) split = 1;
        if (split > SC_MAX_SPLIT) split = SC_MAX_SPLIT;
        p.order[k] = l;
        p.first_block[k] = blocks;
        p.n_slices[l] = ns;
        p.n_split[l] = split;
        p.slice_entries[l] = se;
        p.partial_off[l] = poff;
        if (split > 1) {
            poff += split * size * 2;
            split_entries += size;
        }
        blocks += ns * split;
    }
    p.first_block[g.n_levels] = blocks;
    p.total_blocks = blocks;
    p.partial_floats = poff;
    p.split_entries = split_entries;
    return p;
}

template <int LAYOUT>
__global__ __launch_bounds__(SC_BLOCK) void hashgrid_scatter_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ dout,
                                                                   float* __restrict__ dparams,
                                                                   float* __restrict__ partial, uint32_t M,
                                                                   GridLevels g, ScatterPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [slice entries][2]
    // ---- which (level, slice, sample split) is this workgroup?
    uint32_t k = 0;
    while (k + 1 < plan.n_levels && blockIdx.x >= plan.first_block[k + 1]) ++k;
    const uint32_t level = plan.order[k];
    const uint32_t rel = blockIdx.x - plan.first_block[k];
    const uint32_t n_split = plan.n_split[level];
    const uint32_t slice = rel / n_split, split = rel - slice * n_split;
    const uint32_t off = g.offsets[level];
    const uint32_t size = g.offsets[level + 1] - off;
    const uint32_t se = plan.slice_entries[level];
    const uint32_t begin = slice * se;
    const uint32_t count = begin + se <= size ? se : size - begin;
    const uint32_t res = g.res[level];
    const float scale = g.scale[level];

    for (uint32_t e = threadIdx.x; e < 2 * count; e += SC_BLOCK) acc[e] = 0.0f;
    __syncthreads();

    const uint32_t per = (M + n_split - 1) / n_split;
    const uint32_t s0 = split * per;
    const uint32_t s1 = s0 + per < M ? s0 + per : M;
    const int mode = level_mode(res, size);
    constexpr int UNR = 4;   // independent x loads in flight per thread: the scan is latency-bound otherwise
    for (uint32_t base = s0 + threadIdx.x; base < s1; base += UNR * SC_BLOCK) {
        float px[UNR][3];
        float2 pg[UNR];     // dL/dy is fetched up front too: a load issued inside the hit branch exposes a full
                            // L2 round trip per iteration (the dominant cost of the first version)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t i = base + u * SC_BLOCK;
            const uint32_t ii = i < s1 ? i : s1 - 1;
            px[u][0] = x[3 * (size_t)ii], px[u][1] = x[3 * (size_t)ii + 1], px[u][2] = x[3 * (size_t)ii + 2];
            pg[u] = *reinterpret_cast<const float2*>(dout + feat_index<LAYOUT>(ii, level, M, g.n_levels));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t i = base + u * SC_BLOCK;
            if (i >= s1) break;
            Cell cell;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float pos = fmaf(scale, px[u][d], 0.5f);
                const float fl = floorf(pos);
                cell.c[d] = (uint32_t)(int)fl;
                cell.f[d] = pos - fl;
            }
            uint32_t idx[8];
            corner_indices(mode, cell, res, size, idx);
            uint32_t hit = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                idx[c] -= begin;
                hit |= (idx[c] < count) ? (1u << c) : 0u;
            }
            if (hit) {
                // each lane walks ITS OWN hit list (mean 8/n_slices hits per lane), so the wave issues
                // max-over-lanes(hits) dense atomic pairs instead of 8 sparse ones: an LDS atomic instruction costs
                // ~18 cycles however few lanes are active
                const float2 gy = pg[u];
                do {
                    const int c = __ffs((int)hit) - 1;
                    hit &= hit - 1u;
                    uint32_t e = idx[0];
#pragma unroll
                    for (int k = 1; k < 8; ++k) e = (c == k) ? idx[k] : e;
                    float wgt = (c & 1) ? cell.f[0] : 1.0f - cell.f[0];
                    wgt = wgt * ((c & 2) ? cell.f[1] : 1.0f - cell.f[1]);
                    wgt = wgt * ((c & 4) ? cell.f[2] : 1.0f - cell.f[2]);
                    atomicAdd(&acc[2 * e], wgt * gy.x);
                    atomicAdd(&acc[2 * e + 1], wgt * gy.y);
                } while (hit);
            }
        }
    }
    __syncthreads();

    if (n_split == 1) {
        float2* dst = reinterpret_cast<float2*>(dparams) + off + begin;
        const float2* a2 = reinterpret_cast<const float2*>(acc);
        for (uint32_t e = threadIdx.x; e < count; e += SC_BLOCK) {
            const float2 v = a2[e];
            if (v.x != 0.0f || v.y != 0.0f) {
                float2 d = dst[e];
                d.x += v.x, d.y += v.y;
                dst[e] = d;
            }
        }
    } else {
        float2* dst = reinterpret_cast<float2*>(partial + plan.partial_off[level]) + (size_t)split * size + begin;
        const float2* a2 = reinterpret_cast<const float2*>(acc);
        for (uint32_t e = threadIdx.x; e < count; e += SC_BLOCK) dst[e] = a2[e];
    }
}

// folds the sample-split partial slices of the coarse levels into dparams
__global__ __launch_bounds__(256) void hashgrid_scatter_reduce_kernel(const float* __restrict__ partial,
                                                                      float* __restrict__ dparams, GridLevels g,
                                                                      ScatterPlan plan) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;     // float2 entry index over the split levels
    for (uint32_t l = 0; l < g.n_levels; ++l) {
        const uint32_t ns = plan.n_split[l];
        if (ns <= 1) continue;
        const uint32_t size = g.offsets[l + 1] - g.offsets[l];
        if (q < size) {
            const float2* p2 = reinterpret_cast<const float2*>(partial + plan.partial_off[l]);
            float2 a = make_float2(0.f, 0.f);
            for (uint32_t s = 0; s < ns; ++s) {
                const float2 v = p2[(size_t)s * size + q];
                a.x += v.x, a.y += v.y;
            }
            float2* d = reinterpret_cast<float2*>(dparams) + g.offsets[l] + q;
            float2 cur = *d;
            cur.x += a.x, cur.y += a.y;
            *d = cur;
            return;
        }
        q -= size;
    }
}

// dL/dx of one level per thread (tcnn kernel_grid_backward_input); written to per-level partials, no atomics
template <int LAYOUT>
__global__ __launch_bounds__(HG_BLOCK) void hashgrid_dx_kernel(const float* __restrict__ x,
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
    // dy_f/dx_d = scale * sum over the 4 corner pairs along d of w_other * (right - left)
    float gx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float s0 = 0.f, s1 = 0.f;
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

int mipsf_hashgrid_fwd(const float* x, const float* params, float* out, uint32_t M,
                       const mipsf_grid_meta* meta, int layout, void* stream) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && params && out, "null pointer");
    uint32_t nchunk;
    const uint32_t nb = grid_blocks(M, g.n_levels, nchunk);
    hipStream_t s = (hipStream_t)stream;
    const float2* table = reinterpret_cast<const float2*>(params);
    if (layout == MIPSF_FEAT_AOS)
        hipLaunchKernelGGL(hashgrid_fwd_kernel<MIPSF_FEAT_AOS>, dim3(nb), dim3(HG_BLOCK), 0, s, x, table, out, M, g,
                           nchunk);
    else if (layout == MIPSF_FEAT_LEVEL_MAJOR)
        hipLaunchKernelGGL(hashgrid_fwd_kernel<MIPSF_FEAT_LEVEL_MAJOR>, dim3(nb), dim3(HG_BLOCK), 0, s, x, table,
                           out, M, g, nchunk);
    else
        MIPSF_REQUIRE(false, "bad layout %d", layout);
    return check_launch("hashgrid_fwd");
}

uint64_t mipsf_hashgrid_bwd_scratch_floats(const mipsf_grid_meta* meta, uint32_t M, int need_dx) {
    GridLevels g;
    if (to_levels(meta, g)) return 0;
    const ScatterPlan p = make_plan(g, M);
    return (uint64_t)p.partial_floats + (need_dx ? (uint64_t)g.n_levels * M * 3 : 0) + 64;
}

int mipsf_hashgrid_bwd(const float* x, const float* params, const float* dout, float* dparams, float* dx,
                       float* scratch, uint32_t M, const mipsf_grid_meta* meta, int layout, void* stream) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && params && dout && scratch, "null pointer");
    MIPSF_REQUIRE(dparams || dx, "nothing to compute: dparams and dx are both null");
    MIPSF_REQUIRE(layout == MIPSF_FEAT_AOS || layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout %d", layout);
    hipStream_t s = (hipStream_t)stream;
    const ScatterPlan plan = make_plan(g, M);
    float* partial = scratch;
    float* dxl = scratch + ((plan.partial_floats + 15) / 16) * 16;
    uint32_t max_slice = 0;
    for (uint32_t l = 0; l < g.n_levels; ++l) max_slice = plan.slice_entries[l] > max_slice ? plan.slice_entries[l] : max_slice;
    const uint32_t lds_bytes = max_slice * 8;
#define SCATTER(LAY)                                                                                             \
    do {                                                                                                         \
        static uint32_t attr_bytes = 0;                                                                          \
        if (lds_bytes > attr_bytes) {                                                                            \
            if (hipFuncSetAttribute((const void*)hashgrid_scatter_kernel<LAY>,                                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {      \
                set_error("cannot raise dynamic LDS to %u bytes", lds_bytes);                                    \
                return 4;                                                                                        \
            }                                                                                                    \
            attr_bytes = lds_bytes;                                                                              \
        }                                                                                                        \
        hipLaunchKernelGGL((hashgrid_scatter_kernel<LAY>), dim3(plan.total_blocks), dim3(SC_BLOCK), lds_bytes, s, \
                           x, dout, dparams, partial, M, g, plan);                                               \
    } while (0)
    if (dparams) {   // a frozen grid (tracking) skips the scatter altogether
        if (layout == MIPSF_FEAT_AOS) SCATTER(MIPSF_FEAT_AOS); else SCATTER(MIPSF_FEAT_LEVEL_MAJOR);
        if (int e = check_launch("hashgrid_scatter")) return e;
    }
#undef SCATTER
    if (dparams && plan.split_entries) {
        hipLaunchKernelGGL(hashgrid_scatter_reduce_kernel, dim3((plan.split_entries + 255) / 256), dim3(256), 0, s,
                           partial, dparams, g, plan);
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
