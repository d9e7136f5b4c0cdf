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

// entry index of corner (cx,cy,cz) inside a level of `size` entries
template <bool HASHED>
__device__ __forceinline__ uint32_t corner_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res,
                                                 uint32_t size) {
    if (HASHED) {
        const uint32_t h = cx ^ (cy * P1) ^ (cz * P2);
        return h % size;   // size is a power of two for hashed levels in practice; % keeps it general
    } else {
        const uint32_t idx = cx + cy * res + cz * res * res;
        return idx < size ? idx : idx % size;
    }
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

template <bool HASHED>
__device__ __forceinline__ void corner_indices(const Cell& cell, uint32_t res, uint32_t size, uint32_t idx[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        idx[c] = corner_index<HASHED>(cell.c[0] + (c & 1), cell.c[1] + ((c >> 1) & 1), cell.c[2] + ((c >> 2) & 1),
                                      res, size);
    }
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
    if (level_is_hashed(res, size))
        corner_indices<true>(cell, res, size, idx);
    else
        corner_indices<false>(cell, res, size, idx);
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
template <int LAYOUT, bool NEED_DX>
__global__ __launch_bounds__(HG_BLOCK) void hashgrid_bwd_kernel(const float* __restrict__ x,
                                                                const float2* __restrict__ table,
                                                                const float* __restrict__ dout,
                                                                float* __restrict__ dparams,
                                                                float* __restrict__ dx, uint32_t M,
                                                                GridLevels g, uint32_t nchunk) {
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
    if (level_is_hashed(res, size))
        corner_indices<true>(cell, res, size, idx);
    else
        corner_indices<false>(cell, res, size, idx);
    float w[8];
    corner_weights(cell, w);
    const float2 gy = *reinterpret_cast<const float2*>(dout + feat_index<LAYOUT>(i, level, M, g.n_levels));

    float* lvl_grad = dparams + 2 * (size_t)off;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        unsafeAtomicAdd(lvl_grad + 2 * (size_t)idx[c], w[c] * gy.x);
        unsafeAtomicAdd(lvl_grad + 2 * (size_t)idx[c] + 1, w[c] * gy.y);
    }

    if (NEED_DX) {
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
                // the two dims other than d, in increasing order
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
#pragma unroll
        for (int d = 0; d < 3; ++d) unsafeAtomicAdd(dx + 3 * (size_t)i + d, gx[d]);
    }
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
    if (level_is_hashed(res, size))
        corner_indices<true>(cell, res, size, idx);
    else
        corner_indices<false>(cell, res, size, idx);
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

int mipsf_hashgrid_bwd(const float* x, const float* params, const float* dout, float* dparams, float* dx,
                       uint32_t M, const mipsf_grid_meta* meta, int layout, void* stream) {
    GridLevels g;
    if (int rc = to_levels(meta, g)) return rc;
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && params && dout && dparams, "null pointer");
    MIPSF_REQUIRE(layout == MIPSF_FEAT_AOS || layout == MIPSF_FEAT_LEVEL_MAJOR, "bad layout %d", layout);
    uint32_t nchunk;
    const uint32_t nb = grid_blocks(M, g.n_levels, nchunk);
    hipStream_t s = (hipStream_t)stream;
    const float2* table = reinterpret_cast<const float2*>(params);
#define LAUNCH_BWD(LAY, DX)                                                                                     \
    hipLaunchKernelGGL((hashgrid_bwd_kernel<LAY, DX>), dim3(nb), dim3(HG_BLOCK), 0, s, x, table, dout, dparams, \
                       dx, M, g, nchunk)
    if (layout == MIPSF_FEAT_AOS) {
        if (dx) LAUNCH_BWD(MIPSF_FEAT_AOS, true); else LAUNCH_BWD(MIPSF_FEAT_AOS, false);
    } else {
        if (dx) LAUNCH_BWD(MIPSF_FEAT_LEVEL_MAJOR, true); else LAUNCH_BWD(MIPSF_FEAT_LEVEL_MAJOR, false);
    }
#undef LAUNCH_BWD
    return check_launch("hashgrid_bwd");
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
