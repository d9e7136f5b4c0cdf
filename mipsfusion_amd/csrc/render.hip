// Per-ray kernels: depth-guided sample placement (+ fp64 normalisation), SDF-weight compositing with the
// training losses, and their gradients.  One 64-lane wavefront owns one ray, lanes stride over the S samples
// (S = 64 at the headline configuration = exactly one sample per lane); per-ray reductions are wave
// shuffles, the "first sign change" search is a ballot + ffs.  HBM-bound streaming kernels.
//
// Reference: model/scene_rep.py:58-103 (sdf2weights, raw2outputs), :156-179 (placement), :211-236 (losses);
// helper_functions/utils.py:21-49, 71-111 (get_masks, get_sdf_loss).
#include "pose_dev.h"
#include <cstring>
#include <cstddef>

namespace mipsf {

constexpr int RAYS_PER_BLOCK = 4;
constexpr int MAX_S = 256;

struct PlaceCfg {
    uint32_t n_uniform, n_near;
    int perturb;
    float trunc_total;   // trunc * sc_factor as fp32
};

// --------------------------------------------------------------------- sample placement
// The sorted concatenation of two already-sorted lists is a merge: every element's final slot is its own
// index plus the number of elements of the other list that precede it.  Values (not indices) are all the
// reference keeps from torch.sort (scene_rep.py:164), so tie order is irrelevant and the result is
// bit-identical to a sort.
// One ray's samples: rays_o / rays_d / target depth in registers (o, dv, d; has_target: a depth was given at all), the
// wave's two LDS rows (row: merged list, rb: this ray's depth-guided list), sa: the uniform list.
__device__ __forceinline__ void place_ray(const float (&o)[3], const float (&dv)[3], float d, bool has_target,
                                          const float* __restrict__ noise, const float* __restrict__ z_near_off,
                                          const float* __restrict__ z_near_nodepth, const PlaceCfg& pc, const NormCfg& nc,
                                          float* __restrict__ z_vals, float* __restrict__ xn, uint32_t* __restrict__ counts,
                                          float* row, float* rb, const float* sa, uint32_t n, uint32_t lane) {
    const uint32_t nu = pc.n_uniform, nn = pc.n_near, S = nu + nn;
    const bool has_depth = d > 0.f;   // rows with d <= 0 fall back to linspace(near, far) (scene_rep.py:160)
    for (uint32_t j = lane; j < nn; j += MIPSF_WAVE) rb[j] = has_depth ? z_near_off[j] + d : z_near_nodepth[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // merge by rank, ranks by binary search in the other (sorted) list
    for (uint32_t e = lane; e < S; e += MIPSF_WAVE) {
        float val;
        uint32_t lo = 0, hi;
        if (e < nu) {
            val = sa[e];
            hi = nn;                      // #{b : b < val}
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (rb[mid] < val) lo = mid + 1; else hi = mid;
            }
            row[e + lo] = val;
        } else {
            const uint32_t j = e - nu;
            val = rb[j];
            hi = nu;                      // #{a : a <= val}
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (sa[mid] <= val) lo = mid + 1; else hi = mid;
            }
            row[j + lo] = val;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    uint32_t n_front = 0, n_band = 0;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        float z = row[k];
        if (pc.perturb) {
            const float lo = k > 0 ? 0.5f * (z + row[k - 1]) : z;
            const float hi = k + 1 < S ? 0.5f * (row[k + 1] + z) : z;
            z = lo + (hi - lo) * noise[(size_t)n * S + k];
        }
        z_vals[(size_t)n * S + k] = z;
        const float px = o[0] + dv[0] * z, py = o[1] + dv[1] * z, pz = o[2] + dv[2] * z;
        float* out = xn + ((size_t)n * S + k) * 3;
        out[0] = normalise1(px, nc.sub[0], nc.div[0], nc.norm_factor);
        out[1] = normalise1(py, nc.sub[1], nc.div[1], nc.norm_factor);
        out[2] = normalise1(pz, nc.sub[2], nc.div[2], nc.norm_factor);
        if (has_target) {
            const bool front = z < d - pc.trunc_total;
            const bool back = z > d + pc.trunc_total;
            n_front += front ? 1u : 0u;
            n_band += (!front && !back && has_depth) ? 1u : 0u;
        }
    }
    if (has_target && counts) {
        // per-ray counts (summed by loss_finalize): two global atomics per ray on two shared words serialise at
        // ~12 ns each -- 100 us for 4096 rays -- so nothing is accumulated here
        const float f = wave_sum((float)n_front), b = wave_sum((float)n_band);   // <= 256 each: exact in fp32
        if (lane == 0) {
            counts[2 * n] = (uint32_t)f;
            counts[2 * n + 1] = (uint32_t)b;
        }
    }
}

__global__ __launch_bounds__(RAYS_PER_BLOCK * MIPSF_WAVE) void sample_rays_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ target_d,
    const float* __restrict__ noise, const float* __restrict__ z_uniform, const float* __restrict__ z_near_off,
    const float* __restrict__ z_near_nodepth, PlaceCfg pc, NormCfg nc, float* __restrict__ z_vals,
    float* __restrict__ xn, uint32_t* __restrict__ counts, uint32_t N) {
    __shared__ float zs[RAYS_PER_BLOCK][MAX_S];
    __shared__ float sb[RAYS_PER_BLOCK][MAX_S];   // this ray's depth-guided list
    __shared__ float sa[MAX_S];                   // the uniform list (shared by the block's rays)
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    const uint32_t n = blockIdx.x * RAYS_PER_BLOCK + w;
    for (uint32_t e = threadIdx.x; e < pc.n_uniform; e += RAYS_PER_BLOCK * MIPSF_WAVE) sa[e] = z_uniform[e];
    __syncthreads();
    if (n >= N) return;   // whole wave exits together; no block-level barrier is used below
    const float d = target_d ? target_d[n] : 0.f;
    const float o[3] = {rays_o[3 * n], rays_o[3 * n + 1], rays_o[3 * n + 2]};
    const float dv[3] = {rays_d[3 * n], rays_d[3 * n + 1], rays_d[3 * n + 2]};
    place_ray(o, dv, d, target_d != nullptr, noise, z_near_off, z_near_nodepth, pc, nc, z_vals, xn, counts, zs[w], sb[w], sa, n,
              lane);
}

// Row gather of the ray table + ray construction from the pose parameters + sample placement in ONE launch (they were
// gather_pose_rays_fwd_kernel + sample_rays_kernel, a 5 us launch each in every iteration; rays_o / rays_d never leave
// the registers).  Same arithmetic and NaN conventions as the two kernels.  Wave per ray.
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(RAYS_PER_BLOCK * MIPSF_WAVE) void gather_pose_place_kernel(
    const float* __restrict__ db, uint64_t n_rows, const int64_t* __restrict__ idx, const float* __restrict__ fixed,
    const float* __restrict__ rot, const float* __restrict__ trans, int F, int K, const int64_t* __restrict__ owner,
    const float* __restrict__ noise, const float* __restrict__ z_uniform, const float* __restrict__ z_near_off,
    const float* __restrict__ z_near_nodepth, PlaceCfg pc, NormCfg nc, float* __restrict__ d_cam, float* __restrict__ rgb,
    float* __restrict__ depth, float* __restrict__ z_vals, float* __restrict__ xn, uint32_t* __restrict__ counts,
    uint32_t N) {
    __shared__ float zs[RAYS_PER_BLOCK][MAX_S];
    __shared__ float sb[RAYS_PER_BLOCK][MAX_S];
    __shared__ float sa[MAX_S];
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    const uint32_t n = blockIdx.x * RAYS_PER_BLOCK + w;
    for (uint32_t e = threadIdx.x; e < pc.n_uniform; e += RAYS_PER_BLOCK * MIPSF_WAVE) sa[e] = z_uniform[e];
    __syncthreads();
    if (n >= N) return;
    int64_t r = idx[n];
    if (r < 0) r += (int64_t)n_rows;
    const bool row_ok = r >= 0 && (uint64_t)r < n_rows;
    const float* src = db + 7 * (size_t)(row_ok ? r : 0);
    float v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = row_ok ? src[k] : __builtin_nanf("");      // (wave-uniform address: one request)
    if (lane == 0) {
        d_cam[3 * (size_t)n] = v[0], d_cam[3 * (size_t)n + 1] = v[1], d_cam[3 * (size_t)n + 2] = v[2];
        rgb[3 * (size_t)n] = v[3], rgb[3 * (size_t)n + 1] = v[4], rgb[3 * (size_t)n + 2] = v[5];
        depth[n] = v[6];
    }
    int64_t p = owner[n];
    if (p < 0) p += F + K;
    const bool in_range = p >= 0 && p < F + K;
    const Mat34 m = load_pose(fixed, rot, trans, F, in_range ? (int)p : 0);
    const float dx = in_range ? v[0] : __builtin_nanf(""), dy = v[1], dz = v[2];
    float o[3], dv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dv[j] = (dx * m.r[3 * j] + dy * m.r[3 * j + 1]) + dz * m.r[3 * j + 2];
        o[j] = m.t[j];
    }
    place_ray(o, dv, v[6], true, noise, z_near_off, z_near_nodepth, pc, nc, z_vals, xn, counts, zs[w], sb[w], sa, n, lane);
}

// ------------------------------------------------------------------- compositing helpers
struct RenderCfg {
    float trunc;          // training.trunc
    float band;           // fp32(sc_factor * trunc): z < z_min + band
    float trunc_total;    // fp32(trunc * sc_factor): loss truncation
    float depth_trunc;
    int rgb_missing_nonzero;
    float emd_w;
};

struct RayState {   // per-lane, for up to MAX_S / 64 samples per lane
    float z_min;
    float usum;
};

// first k in [0, S-1) with s[k] * s[k+1] < 0, else 0 (torch.argmax of an all-zero row)
__device__ __forceinline__ uint32_t first_crossing(const float* __restrict__ srow, uint32_t S, uint32_t lane) {
    uint32_t found = 0xFFFFFFFFu;
    for (uint32_t base = 0; base + 1 < S; base += MIPSF_WAVE) {
        const uint32_t k = base + lane;
        const bool hit = (k + 1 < S) && (srow[k] * srow[k + 1] < 0.0f);
        const unsigned long long m = __ballot(hit);
        if (m != 0ull) {
            found = base + (uint32_t)(__ffsll((long long)m) - 1);
            break;
        }
    }
    return found == 0xFFFFFFFFu ? 0u : found;
}

// ------------------------------------------------------------------------ forward
// partial[n*8 + {0..5}] = {rgb_sq, depth_sq(valid), fs_sq, sdf_sq, fs_emd, sdf_emd}; [6] = valid flag
struct LossFinalize {           // FUSED: the last workgroup of render_fwd_kernel finishes the losses (ticket != null)
    const uint32_t* counts;
    uint32_t* ticket;
    float* losses;
    const float* loss_weights;
    float* loss_total;
    double* sums_out;           // != null: the nine sums of THIS batch are left here and the losses are NOT finished (a share of
};                              // a ray-data-parallel batch: mipsf_render_fwd (sums); the sums of all shares go to mipsf_loss_finalize_sums)

MIPSF_SINGLE_FP32 __device__ void finalize_losses(const double (&t)[9], float emd_w, uint32_t N, uint32_t S, float* __restrict__ losses,
                                const float* __restrict__ loss_weights, float* __restrict__ loss_total);

template <bool TRAIN, bool FUSED>
__device__ __forceinline__ void render_fwd_ray(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ target_rgb,
    const float* __restrict__ target_d, const RenderCfg& rc, float* __restrict__ rgb_out, float* __restrict__ depth_out,
    float* __restrict__ var_out, float* __restrict__ disp_out, float* __restrict__ acc_out,
    float* __restrict__ weights_out, float* __restrict__ partial, uint32_t N, uint32_t S, float* srow, uint32_t n,
    uint32_t lane, float (&row)[7]) {
    const float* rraw = raw + (size_t)n * S * 10;
    const float* rz = z_vals + (size_t)n * S;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) srow[k] = rraw[k * 10 + 3];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t kc = first_crossing(srow, S, lane);
    const float z_min = rz[kc];
    const float z_cut = z_min + rc.band;

    float usum = 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float q = srow[k] / rc.trunc;
        const float u = sigmoidf_(q) * sigmoidf_(-q);
        usum += (rz[k] < z_cut) ? u : 0.f;
    }
    usum = wave_sum(usum);
    const float inv = 1.0f / (usum + 1e-8f);

    float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f, a_w = 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float q = srow[k] / rc.trunc;
        const float u = (rz[k] < z_cut) ? sigmoidf_(q) * sigmoidf_(-q) : 0.f;
        const float wn = u * inv;
        if (weights_out) weights_out[(size_t)n * S + k] = wn;
        a_r += wn * sigmoidf_(rraw[k * 10 + 0]);
        a_g += wn * sigmoidf_(rraw[k * 10 + 1]);
        a_b += wn * sigmoidf_(rraw[k * 10 + 2]);
        a_d += wn * rz[k];
        a_w += wn;
    }
    a_r = wave_sum(a_r), a_g = wave_sum(a_g), a_b = wave_sum(a_b), a_d = wave_sum(a_d), a_w = wave_sum(a_w);
    float a_v = 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float q = srow[k] / rc.trunc;
        const float u = (rz[k] < z_cut) ? sigmoidf_(q) * sigmoidf_(-q) : 0.f;
        const float t = rz[k] - a_d;
        a_v += (u * inv) * (t * t);
    }
    a_v = wave_sum(a_v);
    if (lane == 0) {
        rgb_out[3 * n] = a_r, rgb_out[3 * n + 1] = a_g, rgb_out[3 * n + 2] = a_b;
        depth_out[n] = a_d;
        if (var_out) var_out[n] = a_v;
        if (disp_out) disp_out[n] = 1.0f / fmaxf(1e-10f, a_d / a_w);
        if (acc_out) acc_out[n] = a_w;
    }
    if (!TRAIN) return;

    const float d = target_d[n];
    const bool valid = (d > 0.f) && (d < rc.depth_trunc);
    const float cw = (valid || rc.rgb_missing_nonzero) ? 1.f : 0.f;
    float p_fs = 0.f, p_sd = 0.f, p_fe = 0.f, p_se = 0.f;
    const float T = rc.trunc_total;
    const bool has_depth = d > 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float z = rz[k], s = srow[k];
        const bool front = z < d - T;
        const bool back = z > d + T;
        const float fm = front ? 1.f : 0.f;
        const float bm = (!front && !back && has_depth) ? 1.f : 0.f;
        const float ef = s * fm - fm;
        p_fs += ef * ef;
        const float es = (z + s * T) * bm - d * bm;
        p_sd += es * es;
        if (rc.emd_w > 0.f) {
            const float gt = (((d - z) + T) / (2.f * T)) * 4.f;
            float fe = 0.f, se = 0.f;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const float p = rraw[k * 10 + 5 + c];
                fe += p * (float)(4 - c) * fm;
                se += fabsf(gt - (float)c) * bm * p;
            }
            p_fe += fe;
            p_se += se;
        }
    }
    p_fs = wave_sum(p_fs), p_sd = wave_sum(p_sd), p_fe = wave_sum(p_fe), p_se = wave_sum(p_se);
    float* p = partial + (size_t)n * 8;
    const float e0 = a_r * cw - target_rgb[3 * n] * cw;
    const float e1 = a_g * cw - target_rgb[3 * n + 1] * cw;
    const float e2 = a_b * cw - target_rgb[3 * n + 2] * cw;
    const float ed = a_d - d;
    const float v0 = e0 * e0 + e1 * e1 + e2 * e2, v1 = valid ? ed * ed : 0.f, v6 = valid ? 1.f : 0.f;
    if (!FUSED) {
        if (lane == 0) {
            p[0] = v0, p[1] = v1;
            p[2] = p_fs, p[3] = p_sd, p[4] = p_fe, p[5] = p_se;
            p[6] = v6, p[7] = 0.f;
        }
    } else {        // (every value is wave-uniform after the reductions above)
        row[0] = v0, row[1] = v1, row[2] = p_fs, row[3] = p_sd, row[4] = p_fe, row[5] = p_se, row[6] = v6;
    }
}

// The hand-off of the one-launch training forward (see render_fwd_kernel): this workgroup's rays' loss rows (`row`, wave-uniform
// per ray) and their front / band counts are added in fp64 and written through as ITS row of `partial`; the last workgroup through
// the ticket adds all rows in a fixed order and finishes the losses.  Called by every thread of the workgroup.
template <int RPB>
__device__ __forceinline__ void render_fused_tail(const float (&row)[7], uint32_t n, uint32_t w, uint32_t lane, uint32_t N, uint32_t S,
                                                  const RenderCfg& rc, float* __restrict__ partial, const LossFinalize& fin) {
#ifdef RT_ABL_NO_TAIL      // ablation (no losses): what the hand-off costs
    return;
#endif
    __shared__ bool is_last;
    __shared__ double red[RPB][9];
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 7; ++j) red[w][j] = (double)row[j];
        const uint2 c = n < N ? reinterpret_cast<const uint2*>(fin.counts)[n] : make_uint2(0u, 0u);
        red[w][7] = (double)c.x, red[w][8] = (double)c.y;          // integers: exact
    }
    __syncthreads();
    double* rows = reinterpret_cast<double*>(partial);             // [gridDim.x][9]
    if (threadIdx.x < 9) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < RPB; ++q) t += red[q][threadIdx.x];
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(rows + 9 * (size_t)blockIdx.x + threadIdx.x),
                           (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the row is acknowledged before the ticket is taken
    }
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    // thread (g, j): column j of the rows g, g + G, g + 2G, ... -- every load of a thread in flight at once
    constexpr uint32_t T = RPB * MIPSF_WAVE;
    constexpr uint32_t G = T / 9;
    const uint32_t j = threadIdx.x % 9, gidx = threadIdx.x / 9;
    double acc = 0.0;
    if (gidx < G) {
        constexpr int U = 8;
        for (uint32_t b0 = gidx; b0 < gridDim.x; b0 += U * G) {
            unsigned long long q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t b = b0 + u * G;
                q[u] = b < gridDim.x ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(rows + 9 * (size_t)b + j),
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += __longlong_as_double((long long)q[u]);
        }
    }
    __shared__ double red2[T];
    red2[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 9) {
        double t = 0.0;
        for (uint32_t g2 = 0; g2 < G; ++g2) t += red2[g2 * 9 + threadIdx.x];
        red[0][threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) t[c] = red[0][c];
        if (fin.sums_out != nullptr) {
#pragma unroll
            for (int c = 0; c < 9; ++c) fin.sums_out[c] = t[c];
        } else {
            finalize_losses(t, rc.emd_w, N, S, fin.losses, fin.loss_weights, fin.loss_total);
        }
        __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// partial[n*8 + {0..5}] = {rgb_sq, depth_sq(valid), fs_sq, sdf_sq, fs_emd, sdf_emd}; [6] = valid flag
// FUSED (training, fin.ticket given): ONE launch.  Every workgroup adds its four rays' rows (and their front / band
// counts) in fp64, writes the 9 sums through as ITS row of `partial` ([workgroups][9] doubles), waits for the stores'
// acknowledgement (s_waitcnt vmcnt(0)) and takes a device-scope ticket; the LAST workgroup adds all rows in a fixed order
// (so the result does not depend on which workgroup is last), finishes the losses and leaves the ticket at zero.  The same
// hand-off as pose_rays_bwd_kernel (csrc/pose.hip: sc1 stores + vmcnt(0) + ticket, sc1 loads on the reading side).
// 16 rays per workgroup in this form: the tickets are same-address device atomics (~90 per microsecond): the 1024 workgroups
// of the 4-ray form spent 12 us on them (30 us instead of the 21 us of the two launches), 256 workgroups spend 3.
template <bool TRAIN, bool FUSED, int RPB>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(RPB * MIPSF_WAVE) void render_fwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ target_rgb,
    const float* __restrict__ target_d, RenderCfg rc, float* __restrict__ rgb_out, float* __restrict__ depth_out,
    float* __restrict__ var_out, float* __restrict__ disp_out, float* __restrict__ acc_out,
    float* __restrict__ weights_out, float* __restrict__ partial, uint32_t N, uint32_t S, LossFinalize fin) {
    __shared__ float ssdf[RPB][MAX_S];
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    const uint32_t n = blockIdx.x * RPB + w;
    float row[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n < N)
        render_fwd_ray<TRAIN, FUSED>(raw, z_vals, target_rgb, target_d, rc, rgb_out, depth_out, var_out, disp_out, acc_out,
                                     weights_out, partial, N, S, ssdf[w], n, lane, row);
    if (!FUSED) return;
    render_fused_tail<RPB>(row, n, w, lane, N, S, rc, partial, fin);
}

// ------------------------------------------------------------ training forward (+ the backward of the objective) in ONE pass
// The one-launch training forward again, for S <= 128, with the ray's data read ONCE: its row of `raw` (S x 10 floats, contiguous)
// is staged in LDS by coalesced loads -- render_fwd_ray / render_bwd_kernel read it component by component, 40 bytes from lane
// to lane, in every one of their passes, and fetch z[first crossing] from memory behind the ballot -- its depths sit in registers
// (two samples per lane), every sigmoid is evaluated once.  Same expressions in the same order: the same bits as
// render_fwd_kernel<true, true, 16>.
// DRAW: d objective / d raw for an objective gradient of exactly 1 (g_total = 1, no other gradient: what `loss.backward()` on
// render_fwd's loss_total means) is written as well -- render_bwd_kernel's expressions, bit for bit.  What that kernel takes
// from the finished losses (fs_weight, sdf_weight, n_valid) depends on the sample depths and the target depths only: every
// workgroup adds the per-ray counts of ALL rays itself (integers: exact in any order; 48 KB from L2, under the staging loads).
constexpr int RT_RPB = 16;          // rays per workgroup (the ticket traffic: see render_fwd_kernel)
constexpr uint32_t RT_MAX_S = 2 * MIPSF_WAVE;          // RT_KMAX = samples per lane: 1 or 2
// sum of an integer over the wave's lanes (DPP: vector pipe, no LDS traffic; integers add exactly in any order)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1, 2, 4, 8: lane 15 of a row holds the row's sum
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
template <bool DRAW, int RT_KMAX>
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(RT_RPB * MIPSF_WAVE) void render_train_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ target_rgb,
    const float* __restrict__ target_d, RenderCfg rc, float* __restrict__ rgb_out, float* __restrict__ depth_out,
    float* __restrict__ var_out, float* __restrict__ disp_out, float* __restrict__ acc_out,
    float* __restrict__ weights_out, float* __restrict__ partial, uint32_t N, uint32_t S, LossFinalize fin,
    float* __restrict__ draw) {
    extern __shared__ float rt_rows[];                  // [RT_RPB][S * 10]
    __shared__ uint32_t cred[RT_RPB][3];
    constexpr uint32_t T = RT_RPB * MIPSF_WAVE;
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    const uint32_t n = blockIdx.x * RT_RPB + w;
    const bool live = n < N;
    const uint32_t row_words = S * 10u;
    float* sraw = rt_rows + (size_t)w * row_words;
    const uint32_t nn = live ? n : N - 1;
    const float* rraw = raw + (size_t)nn * row_words;
    const float* rz = z_vals + (size_t)nn * S;

    // ---- every load of the ray in flight at once (the counts of all rays first: they are reduced while the row arrives)
    uint32_t cf = 0, cb = 0, cv = 0;
    if (DRAW) {
        for (uint32_t m0 = threadIdx.x; m0 < N; m0 += 4 * T) {
            uint2 c[4];
            float dd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t m = m0 + (uint32_t)u * T;
                const uint32_t mm = m < N ? m : N - 1;
                c[u] = reinterpret_cast<const uint2*>(fin.counts)[mm];
                dd[u] = target_d[mm];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (m0 + (uint32_t)u * T >= N) break;
                cf += c[u].x, cb += c[u].y;
                cv += ((dd[u] > 0.f) && (dd[u] < rc.depth_trunc)) ? 1u : 0u;
            }
        }
    }
    float stage[RT_KMAX * 10];
#pragma unroll
    for (int q = 0; q < RT_KMAX * 10; ++q) {
        const uint32_t i = lane + (uint32_t)q * MIPSF_WAVE;
        stage[q] = i < row_words ? rraw[i] : 0.0f;
    }
    float zz[RT_KMAX];
    bool in[RT_KMAX];
#pragma unroll
    for (int j = 0; j < RT_KMAX; ++j) {
        const uint32_t k = lane + (uint32_t)j * MIPSF_WAVE;
        in[j] = k < S;
        zz[j] = in[j] ? rz[k] : 0.0f;
    }
    const float d = target_d[nn];
    const float t_r = target_rgb[3 * nn], t_g = target_rgb[3 * nn + 1], t_b = target_rgb[3 * nn + 2];
    if (DRAW) {
        cf = wave_sum_u32(cf), cb = wave_sum_u32(cb), cv = wave_sum_u32(cv);
        if (lane == 0) cred[w][0] = cf, cred[w][1] = cb, cred[w][2] = cv;
    }
#pragma unroll
    for (int q = 0; q < RT_KMAX * 10; ++q) {
        const uint32_t i = lane + (uint32_t)q * MIPSF_WAVE;
        if (i < row_words) sraw[i] = stage[q];
    }
    uint32_t n_front = 0, n_band = 0, n_valid = 0;        // of the whole batch (< 2^32: M = N S is a 32-bit count)
    if (DRAW) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RT_RPB; ++q) n_front += cred[q][0], n_band += cred[q][1], n_valid += cred[q][2];
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    float row[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        float sv[RT_KMAX], u_[RT_KMAX], sg[RT_KMAX], c0[RT_KMAX], c1[RT_KMAX], c2[RT_KMAX];
        bool keep[RT_KMAX];
        // first k in [0, S - 1) with s[k] * s[k + 1] < 0, else 0 (first_crossing)
        uint32_t kc = 0;
        bool found = false;
#pragma unroll
        for (int j = 0; j < RT_KMAX; ++j) {
            const uint32_t k = lane + (uint32_t)j * MIPSF_WAVE;
            sv[j] = in[j] ? sraw[k * 10 + 3] : 0.0f;
            const float nxt = (k + 1 < S) ? sraw[(k + 1) * 10 + 3] : 0.0f;
            const bool hit = (k + 1 < S) && (sv[j] * nxt < 0.0f);
            const unsigned long long m = __ballot(hit);
            if (!found && m != 0ull) kc = (uint32_t)j * MIPSF_WAVE + (uint32_t)(__ffsll((long long)m) - 1), found = true;
        }
        float zsel = zz[0];
#pragma unroll
        for (int j = 1; j < RT_KMAX; ++j) zsel = (kc / MIPSF_WAVE == (uint32_t)j) ? zz[j] : zsel;
        const float z_min = __shfl(zsel, (int)(kc & (MIPSF_WAVE - 1)), 64);
        const float z_cut = z_min + rc.band;

        float usum = 0.f;
#pragma unroll
        for (int j = 0; j < RT_KMAX; ++j) {
            if (!in[j]) continue;
            const float q = sv[j] / rc.trunc;
            sg[j] = sigmoidf_(q);
            u_[j] = sg[j] * sigmoidf_(-q);
            keep[j] = zz[j] < z_cut;
            usum += keep[j] ? u_[j] : 0.f;
        }
        usum = wave_sum(usum);
        const float inv = 1.0f / (usum + 1e-8f);

        float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f, a_w = 0.f;
#pragma unroll
        for (int j = 0; j < RT_KMAX; ++j) {
            if (!in[j]) continue;
            const uint32_t k = lane + (uint32_t)j * MIPSF_WAVE;
            const float u = keep[j] ? u_[j] : 0.f;
            const float wn = u * inv;
            if (weights_out) weights_out[(size_t)n * S + k] = wn;
            c0[j] = sigmoidf_(sraw[k * 10 + 0]), c1[j] = sigmoidf_(sraw[k * 10 + 1]), c2[j] = sigmoidf_(sraw[k * 10 + 2]);
            a_r += wn * c0[j];
            a_g += wn * c1[j];
            a_b += wn * c2[j];
            a_d += wn * zz[j];
            a_w += wn;
        }
        a_r = wave_sum(a_r), a_g = wave_sum(a_g), a_b = wave_sum(a_b), a_d = wave_sum(a_d), a_w = wave_sum(a_w);
        float a_v = 0.f;
#pragma unroll
        for (int j = 0; j < RT_KMAX; ++j) {
            if (!in[j]) continue;
            const float u = keep[j] ? u_[j] : 0.f;
            const float t = zz[j] - a_d;
            a_v += (u * inv) * (t * t);
        }
        a_v = wave_sum(a_v);
        if (lane == 0) {
            rgb_out[3 * n] = a_r, rgb_out[3 * n + 1] = a_g, rgb_out[3 * n + 2] = a_b;
            depth_out[n] = a_d;
            if (var_out) var_out[n] = a_v;
            if (disp_out) disp_out[n] = 1.0f / fmaxf(1e-10f, a_d / a_w);
            if (acc_out) acc_out[n] = a_w;
        }

        // ---- the losses' per-ray sums (render_fwd_ray)
        const bool valid = (d > 0.f) && (d < rc.depth_trunc);
        const float cw = (valid || rc.rgb_missing_nonzero) ? 1.f : 0.f;
        float p_fs = 0.f, p_sd = 0.f, p_fe = 0.f, p_se = 0.f;
        const float TT = rc.trunc_total;
        const bool has_depth = d > 0.f;
#pragma unroll
        for (int j = 0; j < RT_KMAX; ++j) {
            if (!in[j]) continue;
            const uint32_t k = lane + (uint32_t)j * MIPSF_WAVE;
            const float z = zz[j], s = sv[j];
            const bool front = z < d - TT;
            const bool back = z > d + TT;
            const float fm = front ? 1.f : 0.f;
            const float bm = (!front && !back && has_depth) ? 1.f : 0.f;
            const float ef = s * fm - fm;
            p_fs += ef * ef;
            const float es = (z + s * TT) * bm - d * bm;
            p_sd += es * es;
            if (rc.emd_w > 0.f) {
                const float gt = (((d - z) + TT) / (2.f * TT)) * 4.f;
                float fe = 0.f, se = 0.f;
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const float p = sraw[k * 10 + 5 + c];
                    fe += p * (float)(4 - c) * fm;
                    se += fabsf(gt - (float)c) * bm * p;
                }
                p_fe += fe;
                p_se += se;
            }
        }
        p_fs = wave_sum(p_fs), p_sd = wave_sum(p_sd), p_fe = wave_sum(p_fe), p_se = wave_sum(p_se);
        const float e0 = a_r * cw - t_r * cw;
        const float e1 = a_g * cw - t_g * cw;
        const float e2 = a_b * cw - t_b * cw;
        const float ed = a_d - d;
        row[0] = e0 * e0 + e1 * e1 + e2 * e2, row[1] = valid ? ed * ed : 0.f, row[2] = p_fs, row[3] = p_sd, row[4] = p_fe, row[5] = p_se;
        row[6] = valid ? 1.f : 0.f;

        if (DRAW) {
            // ---- render_bwd_kernel with g_losses = g_rgb = g_depth = null, g_total = 1, N_norm = N
            float gl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) gl[k] = 0.f + 1.0f * fin.loss_weights[k];
            const float gR = gl[0], gD = gl[1], gS = gl[2], gF = gl[3];
            const float nf = (float)n_front, nb = (float)n_band;
            const float total = nf + nb;
            const float fs_w = 1.0f - nf / total, sdf_w = 1.0f - nb / total;      // finalize_losses
            const float NS = (float)N * (float)S;
            float G_r = 0.f, G_g = 0.f, G_b = 0.f, G_d = 0.f;
            const float k_rgb = gR * 2.f * cw * cw / (3.f * (float)N);
            G_r += k_rgb * (a_r - t_r);
            G_g += k_rgb * (a_g - t_g);
            G_b += k_rgb * (a_b - t_b);
            if (valid) G_d += gD * 2.f * (a_d - d) / (float)n_valid;
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < RT_KMAX; ++j) {
                if (!in[j]) continue;
                const float wn = (keep[j] ? u_[j] : 0.f) * inv;
                const float Gk = G_r * c0[j] + G_g * c1[j] + G_b * c2[j] + G_d * zz[j];
                dot += Gk * wn;
            }
            dot = wave_sum(dot);
#pragma unroll
            for (int j = 0; j < RT_KMAX; ++j) {
                if (!in[j]) continue;
                const uint32_t k = lane + (uint32_t)j * MIPSF_WAVE;
                const float z = zz[j], s = sv[j];
                const float a = u_[j];
                const float wn = (keep[j] ? a : 0.f) * inv;
                const float Gk = G_r * c0[j] + G_g * c1[j] + G_b * c2[j] + G_d * z;
                float ds = keep[j] ? (Gk - dot) * inv * (a * (1.f - 2.f * sg[j]) / rc.trunc) : 0.f;
                float dp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                const bool front = z < d - TT;
                const bool back = z > d + TT;
                const float fm = front ? 1.f : 0.f;
                const float bm = (!front && !back && has_depth) ? 1.f : 0.f;
                ds += gF * fs_w * (2.f / NS) * fm * (s * fm - fm);
                ds += gS * sdf_w * (2.f / NS) * (bm * TT) * ((z + s * TT) * bm - d * bm);
                if (rc.emd_w > 0.f) {
                    const float gt = (((d - z) + TT) / (2.f * TT)) * 4.f;
                    const float kf = gF * rc.emd_w / (250.f * NS), ks = gS * rc.emd_w / (5000.f * NS);
#pragma unroll
                    for (int c = 0; c < 5; ++c) dp[c] = kf * fm * (float)(4 - c) + ks * bm * fabsf(gt - (float)c);
                }
                float* o = sraw + k * 10;          // this sample's own words: nobody else reads them any more
                o[0] = G_r * wn * c0[j] * (1.f - c0[j]);
                o[1] = G_g * wn * c1[j] * (1.f - c1[j]);
                o[2] = G_b * wn * c2[j] * (1.f - c2[j]);
                o[3] = ds;
                o[4] = 0.f;
#pragma unroll
                for (int c = 0; c < 5; ++c) o[5 + c] = dp[c];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float* rdr = draw + (size_t)n * row_words;
#pragma unroll
            for (int q = 0; q < RT_KMAX * 10; ++q) {
                const uint32_t i = lane + (uint32_t)q * MIPSF_WAVE;
                if (i < row_words) rdr[i] = sraw[i];
            }
        }
    }
    render_fused_tail<RT_RPB>(row, n, w, lane, N, S, rc, partial, fin);
}

// losses[8] = {rgb_loss, depth_loss, sdf_loss, fs_loss, psnr, fs_weight, sdf_weight, n_valid}
constexpr int LF_BLOCK = 1024;
__global__ __launch_bounds__(LF_BLOCK) void loss_finalize_kernel(const float* __restrict__ partial,
                                                            const uint32_t* __restrict__ counts, float emd_w,
                                                            float* __restrict__ losses, uint32_t N, uint32_t S,
                                                            const float* __restrict__ loss_weights,
                                                            float* __restrict__ loss_total) {
    __shared__ double red[LF_BLOCK / MIPSF_WAVE][9];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // four rays' loads in flight per thread (4096 rays = 4 per thread: one memory round trip instead of four); the
    // additions keep their order
    for (uint32_t n0 = threadIdx.x; n0 < N; n0 += 4 * LF_BLOCK) {
        float4 p0[4], p1[4];
        uint2 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t n = n0 + u * LF_BLOCK;
            const uint32_t m = n < N ? n : N - 1;
            p0[u] = reinterpret_cast<const float4*>(partial)[2 * (size_t)m];
            p1[u] = reinterpret_cast<const float4*>(partial)[2 * (size_t)m + 1];
            c[u] = reinterpret_cast<const uint2*>(counts)[m];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (n0 + u * LF_BLOCK >= N) break;
            acc[0] += (double)p0[u].x, acc[1] += (double)p0[u].y, acc[2] += (double)p0[u].z, acc[3] += (double)p0[u].w;
            acc[4] += (double)p1[u].x, acc[5] += (double)p1[u].y, acc[6] += (double)p1[u].z;
            acc[7] += (double)c[u].x;             // integers < 2^53: exact
            acc[8] += (double)c[u].y;
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = wave_sum_d(acc[j]);
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j) red[w][j] = acc[j];
    }
    __syncthreads();
    if (w != 0) return;
    // second stage in the first wave (lane k holds wave k's partial): a single thread walking the 16 x 9 doubles
    // through LDS was a third of this kernel's 12 us
    double t[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] = wave_sum_d(lane < LF_BLOCK / MIPSF_WAVE ? red[lane][j] : 0.0);
    if (lane == 0) finalize_losses(t, emd_w, N, S, losses, loss_weights, loss_total);
}

MIPSF_SINGLE_FP32 __device__ void finalize_losses(const double (&t)[9], float emd_w, uint32_t N, uint32_t S, float* __restrict__ losses,
                                const float* __restrict__ loss_weights, float* __restrict__ loss_total) {
    const double NS = (double)N * (double)S;
    const float n_front = (float)t[7], n_band = (float)t[8];
    const float total = n_front + n_band;
    const float fs_w = 1.0f - n_front / total;     // 0/0 -> NaN exactly like the reference
    const float sdf_w = 1.0f - n_band / total;
    const float rgb_loss = (float)(t[0] / (3.0 * (double)N));
    const float depth_loss = (float)(t[1] / t[6]);   // no valid depth -> 0/0 = NaN (mse of an empty tensor)
    float fs = (float)(t[2] / NS) * fs_w;
    float sd = (float)(t[3] / NS) * sdf_w;
    if (emd_w > 0.f) {
        fs = fs + ((float)(t[4] / NS) / 250.f) * emd_w;
        sd = sd + ((float)(t[5] / NS) / 5000.f) * emd_w;
    }
    losses[0] = rgb_loss;
    losses[1] = depth_loss;
    losses[2] = sd;
    losses[3] = fs;
    losses[4] = -10.f * logf(rgb_loss) / logf(10.f);
    losses[5] = fs_w;
    losses[6] = sdf_w;
    losses[7] = (float)t[6];
    // the training objective itself (MIPSFusion.get_loss_from_ret, mipsfusion.py:142-152): the same products added left
    // to right in fp32 -- saves the caller a dot product forward and a scaling pass backward (5 us launches each)
    if (loss_total) {
        float tot = 0.0f;
        tot = tot + loss_weights[0] * rgb_loss;
        tot = tot + loss_weights[1] * depth_loss;
        tot = tot + loss_weights[2] * sd;
        tot = tot + loss_weights[3] * fs;
        loss_total[0] = tot;
    }
}

// the losses of a batch whose nine sums were formed elsewhere (the all-reduced sums of a ray-data-parallel batch's shares)
MIPSF_SINGLE_FP32 __global__ void loss_finalize_sums_kernel(const double* __restrict__ sums, float emd_w, uint32_t N, uint32_t S,
                                          float* __restrict__ losses, const float* __restrict__ loss_weights,
                                          float* __restrict__ loss_total) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double t[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) t[c] = sums[c];
    finalize_losses(t, emd_w, N, S, losses, loss_weights, loss_total);
}

// ------------------------------------------------------------------------ backward
// N_norm: the ray count the losses were normalised by (= N unless this launch differentiates a SHARE of a larger batch)
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(RAYS_PER_BLOCK * MIPSF_WAVE) void render_bwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ target_rgb,
    const float* __restrict__ target_d, const float* __restrict__ losses, RenderCfg rc, int train,
    const float* __restrict__ g_losses, const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
    float* __restrict__ draw, uint32_t N, uint32_t S, const float* __restrict__ g_total,
    const float* __restrict__ loss_weights, uint32_t N_norm, int keep_if_unit) {
    // draw already holds the gradient for an objective gradient of exactly 1 (render_train_kernel<true, .>): nothing to do then
    if (keep_if_unit && g_total[0] == 1.0f) return;
    __shared__ float ssdf[RAYS_PER_BLOCK][MAX_S];
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    const uint32_t n = blockIdx.x * RAYS_PER_BLOCK + w;
    if (n >= N) return;
    const float* rraw = raw + (size_t)n * S * 10;
    const float* rz = z_vals + (size_t)n * S;
    float* rdr = draw + (size_t)n * S * 10;
    float* srow = ssdf[w];
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) srow[k] = rraw[k * 10 + 3];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t kc = first_crossing(srow, S, lane);
    const float z_cut = rz[kc] + rc.band;

    // recompute the forward reductions
    float usum = 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float q = srow[k] / rc.trunc;
        usum += (rz[k] < z_cut) ? sigmoidf_(q) * sigmoidf_(-q) : 0.f;
    }
    usum = wave_sum(usum);
    const float inv = 1.0f / (usum + 1e-8f);
    float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float q = srow[k] / rc.trunc;
        const float wn = ((rz[k] < z_cut) ? sigmoidf_(q) * sigmoidf_(-q) : 0.f) * inv;
        a_r += wn * sigmoidf_(rraw[k * 10 + 0]);
        a_g += wn * sigmoidf_(rraw[k * 10 + 1]);
        a_b += wn * sigmoidf_(rraw[k * 10 + 2]);
        a_d += wn * rz[k];
    }
    a_r = wave_sum(a_r), a_g = wave_sum(a_g), a_b = wave_sum(a_b), a_d = wave_sum(a_d);

    // gradients reaching the rendered maps
    float G_r = g_rgb ? g_rgb[3 * n] : 0.f, G_g = g_rgb ? g_rgb[3 * n + 1] : 0.f, G_b = g_rgb ? g_rgb[3 * n + 2] : 0.f;
    float G_d = g_depth ? g_depth[n] : 0.f;
    float d = 0.f, gS = 0.f, gF = 0.f, fs_w = 0.f, sdf_w = 0.f;
    const float NS = (float)N_norm * (float)S;
    if (train) {
        d = target_d[n];
        const bool valid = (d > 0.f) && (d < rc.depth_trunc);
        const float cw = (valid || rc.rgb_missing_nonzero) ? 1.f : 0.f;
        // d objective / d {rgb, depth, sdf, fs}_loss: given directly and / or as (d objective / d total) x weights
        float gl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) gl[k] = (g_losses ? g_losses[k] : 0.f) + (g_total ? g_total[0] * loss_weights[k] : 0.f);
        const float gR = gl[0], gD = gl[1];
        gS = gl[2], gF = gl[3];
        fs_w = losses[5], sdf_w = losses[6];
        const float k_rgb = gR * 2.f * cw * cw / (3.f * (float)N_norm);
        G_r += k_rgb * (a_r - target_rgb[3 * n]);
        G_g += k_rgb * (a_g - target_rgb[3 * n + 1]);
        G_b += k_rgb * (a_b - target_rgb[3 * n + 2]);
        if (valid) G_d += gD * 2.f * (a_d - d) / losses[7];
    }
    // dot = sum_k G_k * wn_k
    float dot = 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float q = srow[k] / rc.trunc;
        const float wn = ((rz[k] < z_cut) ? sigmoidf_(q) * sigmoidf_(-q) : 0.f) * inv;
        const float Gk = G_r * sigmoidf_(rraw[k * 10 + 0]) + G_g * sigmoidf_(rraw[k * 10 + 1]) +
                         G_b * sigmoidf_(rraw[k * 10 + 2]) + G_d * rz[k];
        dot += Gk * wn;
    }
    dot = wave_sum(dot);

    const float T = rc.trunc_total;
    const bool has_depth = d > 0.f;
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float z = rz[k], s = srow[k];
        const float q = s / rc.trunc;
        const float sg = sigmoidf_(q);
        const bool keep = z < z_cut;
        const float a = sg * sigmoidf_(-q);
        const float wn = (keep ? a : 0.f) * inv;
        const float c0 = sigmoidf_(rraw[k * 10 + 0]), c1 = sigmoidf_(rraw[k * 10 + 1]), c2 = sigmoidf_(rraw[k * 10 + 2]);
        const float Gk = G_r * c0 + G_g * c1 + G_b * c2 + G_d * z;
        float ds = keep ? (Gk - dot) * inv * (a * (1.f - 2.f * sg) / rc.trunc) : 0.f;
        float dp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (train) {
            const bool front = z < d - T;
            const bool back = z > d + T;
            const float fm = front ? 1.f : 0.f;
            const float bm = (!front && !back && has_depth) ? 1.f : 0.f;
            ds += gF * fs_w * (2.f / NS) * fm * (s * fm - fm);
            ds += gS * sdf_w * (2.f / NS) * (bm * T) * ((z + s * T) * bm - d * bm);
            if (rc.emd_w > 0.f) {
                const float gt = (((d - z) + T) / (2.f * T)) * 4.f;
                const float kf = gF * rc.emd_w / (250.f * NS), ks = gS * rc.emd_w / (5000.f * NS);
#pragma unroll
                for (int c = 0; c < 5; ++c) dp[c] = kf * fm * (float)(4 - c) + ks * bm * fabsf(gt - (float)c);
            }
        }
        float* o = rdr + k * 10;
        o[0] = G_r * wn * c0 * (1.f - c0);
        o[1] = G_g * wn * c1 * (1.f - c1);
        o[2] = G_b * wn * c2 * (1.f - c2);
        o[3] = ds;
        o[4] = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) o[5 + c] = dp[c];
    }
}

// d(xn) -> d(rays_o), d(rays_d): pts = o + d*z (scene_rep.py:179) then fp64 normalisation (:140/:142)
__global__ __launch_bounds__(RAYS_PER_BLOCK * MIPSF_WAVE) void rays_bwd_kernel(const float* __restrict__ dxn,
                                                                              const float* __restrict__ z_vals,
                                                                              NormCfg nc, float* __restrict__ d_o,
                                                                              float* __restrict__ d_d, uint32_t N,
                                                                              uint32_t S) {
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    const uint32_t n = blockIdx.x * RAYS_PER_BLOCK + w;
    if (n >= N) return;
    float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
    for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
        const float z = z_vals[(size_t)n * S + k];
        const float* g = dxn + ((size_t)n * S + k) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float gp = (float)(((double)g[d] / nc.norm_factor) / nc.div[d]);
            so[d] += gp;
            sd[d] += gp * z;
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        so[d] = wave_sum(so[d]);
        sd[d] = wave_sum(sd[d]);
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            d_o[3 * n + d] = so[d];
            d_d[3 * n + d] = sd[d];
        }
    }
}

// d(xn) -> pose gradients in ONE launch (rays_bwd_kernel + pose_rays_bwd_kernel: the per-ray {d o, d d} stay in
// registers).  16 rays per workgroup: 256 tickets for 4096 rays (same-address device atomics retire at ~90 per us).
constexpr int PPB = 16;
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(PPB * MIPSF_WAVE) void place_pose_bwd_kernel(
    const float* __restrict__ dxn, const float* __restrict__ z_vals, NormCfg nc, const float* __restrict__ d_cam,
    const int64_t* __restrict__ owner, const float* __restrict__ rot, int F, int K, float* __restrict__ part,
    uint32_t* __restrict__ ticket, float* __restrict__ d_rot, float* __restrict__ d_trans, uint32_t N, uint32_t S,
    int accumulate) {
    __shared__ float sacc[PR_MAX_POSES * 12];
    __shared__ float wv[PPB][12];
    __shared__ int wp[PPB];
    __shared__ bool is_last;
    const int P = F + K;
    for (int q = threadIdx.x; q < P * 12; q += PPB * MIPSF_WAVE) sacc[q] = 0.f;
    __syncthreads();
    const uint32_t w = threadIdx.x / MIPSF_WAVE, lane = threadIdx.x & (MIPSF_WAVE - 1);
    const uint32_t n = blockIdx.x * PPB + w;
    if (n < N) {
        float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
        for (uint32_t k = lane; k < S; k += MIPSF_WAVE) {
            const float z = z_vals[(size_t)n * S + k];
            const float* g = dxn + ((size_t)n * S + k) * 3;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float gp = (float)(((double)g[d] / nc.norm_factor) / nc.div[d]);
                so[d] += gp;
                sd[d] += gp * z;
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            so[d] = wave_sum(so[d]);
            sd[d] = wave_sum(sd[d]);
        }
        int64_t p = owner[n];
        if (p < 0) p += P;
        if (p < 0 || p >= P) p = 0;              // (the forward already produced NaN rays for this owner)
        const float dx = d_cam[3 * (size_t)n], dy = d_cam[3 * (size_t)n + 1], dz = d_cam[3 * (size_t)n + 2];
        if (lane < 12) {                         // lane q: entry q of this ray's {dR (9), dt (3)}, left in the wave's own slot
            const int j = lane < 9 ? (int)lane / 3 : (int)lane - 9;
            const float gd = j == 0 ? sd[0] : (j == 1 ? sd[1] : sd[2]);
            const float go = j == 0 ? so[0] : (j == 1 ? so[1] : so[2]);
            const int c = (int)lane % 3;
            const float dc = c == 0 ? dx : (c == 1 ? dy : dz);
            wv[w][lane] = lane < 9 ? gd * dc : go;
        }
        if (lane == 0) wp[w] = (int)p;
    } else if (lane == 0) {
        wp[w] = -1;
    }
    // the rays' entries are added pose row by pose row in RAY order by one thread per entry (LDS float atomics from 16 waves
    // arrive in any order: two identical launches used to differ in the last bit of the pose gradients)
    __syncthreads();
    if (threadIdx.x < 12) {
        for (int ww = 0; ww < PPB; ++ww)
            if (wp[ww] >= 0) sacc[wp[ww] * 12 + (int)threadIdx.x] += wv[ww][threadIdx.x];
    }
    pose_block_finish<PPB * MIPSF_WAVE>(sacc, &is_last, P, F, K, part, ticket, rot, d_rot, d_trans, accumulate != 0);
}

static RenderCfg to_render_cfg(const mipsf_render_cfg& c) {
    RenderCfg r;
    r.trunc = c.trunc;
    r.band = (float)((double)c.sc_factor * (double)c.trunc);
    r.trunc_total = (float)((double)c.trunc * (double)c.sc_factor);
    r.depth_trunc = c.depth_trunc;
    r.rgb_missing_nonzero = c.rgb_missing_nonzero;
    r.emd_w = c.emd_w;
    return r;
}

}  // namespace mipsf

using namespace mipsf;

extern "C" {

int mipsf_sample_rays(const float* rays_o, const float* rays_d, const float* target_d, const float* noise,
                      const float* z_uniform, const float* z_near_offsets, const float* z_near_nodepth,
                      const mipsf_render_cfg* cfg, float* z_vals, float* xn, uint32_t* counts, uint32_t N,
                      void* stream) {
    if (N == 0) return 0;
    MIPSF_REQUIRE(cfg && rays_o && rays_d && z_vals && xn, "null pointer");
    MIPSF_REQUIRE(z_uniform || cfg->n_uniform == 0, "null pointer: z_uniform with n_uniform = %u", cfg->n_uniform);
    const uint32_t S = cfg->n_uniform + cfg->n_near;
    MIPSF_REQUIRE(S >= 1 && S <= MAX_S, "samples per ray %u outside [1,%d]", S, MAX_S);
    MIPSF_REQUIRE(cfg->n_near == 0 || (target_d && z_near_offsets && z_near_nodepth),
                  "depth-guided samples need target_d and the near tables");
    MIPSF_REQUIRE(!cfg->perturb || noise, "perturb needs the noise tensor");
    PlaceCfg pc;
    pc.n_uniform = cfg->n_uniform;
    pc.n_near = cfg->n_near;
    pc.perturb = cfg->perturb;
    pc.trunc_total = (float)((double)cfg->trunc * (double)cfg->sc_factor);
    hipLaunchKernelGGL(sample_rays_kernel, dim3((N + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK),
                       dim3(RAYS_PER_BLOCK * MIPSF_WAVE), 0, (hipStream_t)stream, rays_o, rays_d, target_d, noise,
                       z_uniform, z_near_offsets, z_near_nodepth, pc, make_norm(*cfg), z_vals, xn, counts, N);
    return check_launch("sample_rays");
}

int mipsf_gather_pose_place_fwd(const float* db, uint64_t n_rows, const int64_t* idx, const float* fixed_poses,
                                const float* rot, const float* trans, uint32_t F, uint32_t K, const int64_t* owner,
                                const float* noise, const float* z_uniform, const float* z_near_offsets,
                                const float* z_near_nodepth, const mipsf_render_cfg* cfg, float* d_cam, float* rgb,
                                float* depth, float* z_vals, float* xn, uint32_t* counts, uint32_t N, void* stream) {
    if (N == 0) return 0;
    MIPSF_REQUIRE(cfg && db && idx && owner && d_cam && rgb && depth && z_vals && xn, "null pointer");
    MIPSF_REQUIRE(z_uniform || cfg->n_uniform == 0, "null pointer: z_uniform with n_uniform = %u", cfg->n_uniform);
    MIPSF_REQUIRE((F == 0 || fixed_poses) && (K == 0 || (rot && trans)), "null pose pointer");
    MIPSF_REQUIRE(F + K >= 1 && F + K <= (uint32_t)PR_MAX_POSES, "number of poses %u outside [1,%d]", F + K, PR_MAX_POSES);
    const uint32_t S = cfg->n_uniform + cfg->n_near;
    MIPSF_REQUIRE(S >= 1 && S <= MAX_S, "samples per ray %u outside [1,%d]", S, MAX_S);
    MIPSF_REQUIRE(cfg->n_near == 0 || (z_near_offsets && z_near_nodepth), "depth-guided samples need the near tables");
    MIPSF_REQUIRE(!cfg->perturb || noise, "perturb needs the noise tensor");
    PlaceCfg pc;
    pc.n_uniform = cfg->n_uniform;
    pc.n_near = cfg->n_near;
    pc.perturb = cfg->perturb;
    pc.trunc_total = (float)((double)cfg->trunc * (double)cfg->sc_factor);
    hipLaunchKernelGGL(gather_pose_place_kernel, dim3((N + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK),
                       dim3(RAYS_PER_BLOCK * MIPSF_WAVE), 0, (hipStream_t)stream, db, n_rows, idx, fixed_poses, rot, trans,
                       (int)F, (int)K, owner, noise, z_uniform, z_near_offsets, z_near_nodepth, pc, make_norm(*cfg), d_cam, rgb,
                       depth, z_vals, xn, counts, N);
    return check_launch("gather_pose_place");
}

}  // extern "C"
namespace mipsf {
uint64_t place_pose_scratch_floats(uint32_t F, uint32_t K, uint32_t N) {
    return 1ull + 12ull * (F + K) * ((N + PPB - 1) / PPB);
}
uint64_t render_partial_floats(uint32_t N) {
    // per-ray rows of the two-launch form (8 floats per ray) or one row of nine doubles per 16-ray workgroup of the fused form
    const uint64_t rows = 8ull * N, fused = 18ull * (((uint64_t)N + 15) / 16);
    return rows > fused ? rows : fused;
}
}  // namespace mipsf
extern "C" {

int mipsf_place_pose_bwd(const float* dxn, const float* z_vals, const mipsf_render_cfg* cfg, const float* rot, uint32_t F,
                         uint32_t K, const int64_t* owner, const float* d_cam, float* d_rot, float* d_trans, float* scratch,
                         uint32_t N, uint32_t S, int accumulate, void* stream) {
    MIPSF_REQUIRE(K >= 1, "no optimisable pose");
    MIPSF_REQUIRE(cfg && dxn && z_vals && rot && owner && d_cam && d_rot && d_trans && scratch, "null pointer");
    MIPSF_REQUIRE(F + K <= (uint32_t)PR_MAX_POSES, "number of poses %u above %d", F + K, PR_MAX_POSES);
    if (N == 0) return 0;
    // scratch[0] = ticket (zero on entry, zero again on return), then one row of partials per workgroup
    hipLaunchKernelGGL(place_pose_bwd_kernel, dim3((N + PPB - 1) / PPB), dim3(PPB * MIPSF_WAVE), 0, (hipStream_t)stream, dxn,
                       z_vals, make_norm(*cfg), d_cam, owner, rot, (int)F, (int)K, scratch + 1,
                       reinterpret_cast<uint32_t*>(scratch), d_rot, d_trans, N, S, accumulate);
    return check_launch("place_pose_bwd");
}

// render_train_kernel (S <= RT_MAX_S): RT_RPB x S x 40 bytes of dynamic LDS beside ~9.5 KB of static LDS (together above 64 KB
// from S = 88, 90 KB at S = 128: inside gfx950's 160 KB).  Returns -1 -- nothing launched -- when the device does not grant the
// dynamic size (a build for another ARCH): the caller then takes the kernel without the LDS row.
static int launch_render_train(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                               const RenderCfg& rc, float* rgb, float* depth, float* depth_var, float* disp, float* acc,
                               float* weights, float* partial, uint32_t N, uint32_t S, const LossFinalize& fin, float* draw,
                               hipStream_t s) {
    const uint32_t lds = (uint32_t)RT_RPB * S * 40u;
    const int two = S > MIPSF_WAVE ? 1 : 0;       // samples per lane - 1
    const void* fn[2][2] = {{(const void*)render_train_kernel<false, 1>, (const void*)render_train_kernel<false, 2>},
                            {(const void*)render_train_kernel<true, 1>, (const void*)render_train_kernel<true, 2>}};
    static uint32_t attr_dev[MAX_DEVICES][2] = {};
    uint32_t& attr = attr_dev[device_slot()][draw ? 1 : 0];
    if (lds > 48u * 1024u && lds > attr) {        // (only the two-samples-per-lane kernels get there)
        if (hipFuncSetAttribute(fn[draw ? 1 : 0][1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        attr = lds;       // (two threads racing here both set the attribute to a size that covers their launch: benign)
    }
    const dim3 grid((N + RT_RPB - 1) / RT_RPB), block(RT_RPB * MIPSF_WAVE);
#define RT_LAUNCH(D, K)                                                                                                         \
    hipLaunchKernelGGL((render_train_kernel<D, K>), grid, block, lds, s, raw, z_vals, target_rgb, target_d, rc, rgb, depth, depth_var, \
                       disp, acc, weights, partial, N, S, fin, draw)
    if (draw) { if (two) RT_LAUNCH(true, 2); else RT_LAUNCH(true, 1); }
    else { if (two) RT_LAUNCH(false, 2); else RT_LAUNCH(false, 1); }
#undef RT_LAUNCH
    return 0;
}

static int render_fwd_sums(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                           const uint32_t* counts, const mipsf_render_cfg* cfg, float* rgb, float* depth, float* depth_var,
                           float* disp, float* acc, float* weights, float* partial, double* sums, uint32_t* ticket, uint32_t N,
                           uint32_t S, void* stream);

// ONE entry point for the forward (round 5; include/mipsf.h): evaluation, training in two launches, training in one launch
// (ticket), the training objective formed in the same launch (loss_weights / loss_total), a share of a ray-data-parallel
// batch (sums).
int mipsf_render_fwd(const mipsf_render_fwd_args* a_in, void* stream) {
    MIPSF_REQUIRE(a_in != nullptr, "null argument block");
    // (a block that ends before `draw` -- the first form of this ABI version -- is accepted: the field reads as null)
    MIPSF_REQUIRE(a_in->struct_size == sizeof(mipsf_render_fwd_args) || a_in->struct_size == offsetof(mipsf_render_fwd_args, draw),
                  "mipsf_render_fwd_args: struct_size %u, this library expects %u", a_in->struct_size,
                  (unsigned)sizeof(mipsf_render_fwd_args));
    mipsf_render_fwd_args a_copy = {};
    memcpy(&a_copy, a_in, a_in->struct_size);
    const mipsf_render_fwd_args* a = &a_copy;
    const float* raw = a->raw; const float* z_vals = a->z_vals; const float* target_rgb = a->target_rgb; const float* target_d = a->target_d;
    const uint32_t* counts = a->counts; const mipsf_render_cfg* cfg = a->cfg; float* rgb = a->rgb; float* depth = a->depth;
    float* depth_var = a->depth_var; float* disp = a->disp; float* acc = a->acc; float* weights = a->weights; float* losses = a->losses;
    float* partial = a->partial; const float* loss_weights = a->loss_weights; float* loss_total = a->loss_total;
    uint32_t* ticket = a->ticket; const uint32_t N = a->N, S = a->S;
    if (a->sums != nullptr) {
        MIPSF_REQUIRE(losses == nullptr && loss_weights == nullptr && loss_total == nullptr && a->draw == nullptr,
                      "sums: the losses of a share are finished by mipsf_loss_finalize_sums, not here");
        return render_fwd_sums(raw, z_vals, target_rgb, target_d, counts, cfg, rgb, depth, depth_var, disp, acc, weights, partial,
                               a->sums, ticket, N, S, stream);
    }
    MIPSF_REQUIRE((loss_weights == nullptr) == (loss_total == nullptr), "loss_weights and loss_total come together");
    MIPSF_REQUIRE(loss_total == nullptr || losses != nullptr, "loss_total needs the training mode (losses)");
    if (N == 0) return 0;
    MIPSF_REQUIRE(cfg && raw && z_vals && rgb && depth, "null pointer");
    MIPSF_REQUIRE(S >= 1 && S <= MAX_S, "samples per ray %u outside [1,%d]", S, MAX_S);
    const RenderCfg rc = to_render_cfg(*cfg);
    const dim3 grid((N + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK), block(RAYS_PER_BLOCK * MIPSF_WAVE);
    hipStream_t s = (hipStream_t)stream;
    const LossFinalize fin = {counts, ticket, losses, loss_weights, loss_total, nullptr};
    if (losses) {
        MIPSF_REQUIRE(target_rgb && target_d && counts && partial, "training mode needs targets, counts, partial");
        MIPSF_REQUIRE(a->draw == nullptr || (ticket && loss_total && S <= RT_MAX_S),
                      "draw: the one-launch form with the objective (ticket, loss_weights, loss_total) and S <= %u", RT_MAX_S);
        if (ticket && S <= RT_MAX_S) {      // one launch, the ray read once; with `draw` also the backward of the objective
            const int e = launch_render_train(raw, z_vals, target_rgb, target_d, rc, rgb, depth, depth_var, disp, acc, weights, partial,
                                              N, S, fin, a->draw, s);
            if (e > 0) return e;
            if (e == 0) return check_launch("render_fwd");
            MIPSF_REQUIRE(a->draw == nullptr, "draw: this device does not grant render_train_kernel's %u bytes of LDS", RT_RPB * S * 40u);
        }
        if (ticket) {       // one launch: the last workgroup finishes the losses
            constexpr int RPB = 16;
            hipLaunchKernelGGL((render_fwd_kernel<true, true, RPB>), dim3((N + RPB - 1) / RPB), dim3(RPB * MIPSF_WAVE), 0, s, raw,
                               z_vals, target_rgb, target_d, rc, rgb, depth, depth_var, disp, acc, weights, partial, N, S, fin);
            return check_launch("render_fwd");
        }
        hipLaunchKernelGGL((render_fwd_kernel<true, false, RAYS_PER_BLOCK>), grid, block, 0, s, raw, z_vals, target_rgb, target_d, rc, rgb,
                           depth, depth_var, disp, acc, weights, partial, N, S, fin);
        if (int e = check_launch("render_fwd")) return e;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LF_BLOCK), 0, s, partial, counts, rc.emd_w, losses, N, S,
                           loss_weights, loss_total);
        return check_launch("loss_finalize");
    }
    hipLaunchKernelGGL((render_fwd_kernel<false, false, RAYS_PER_BLOCK>), grid, block, 0, s, raw, z_vals, target_rgb, target_d, rc, rgb, depth,
                       depth_var, disp, acc, weights, partial, N, S, fin);
    return check_launch("render_fwd");
}

// A SHARE of a batch (ray-data-parallel training): the per-ray maps of this share and the nine fp64 sums its losses are made
// of -- {rgb_sq, depth_sq(valid), fs_sq, sdf_sq, fs_emd, sdf_emd, n_valid, n_front, n_band} -- in sums[9] (device).  The
// caller adds the shares' sums (an all-reduce of 72 bytes) and finishes the losses with mipsf_loss_finalize_sums.
static int render_fwd_sums(const float* raw, const float* z_vals, const float* target_rgb, const float* target_d,
                           const uint32_t* counts, const mipsf_render_cfg* cfg, float* rgb, float* depth, float* depth_var,
                           float* disp, float* acc, float* weights, float* partial, double* sums, uint32_t* ticket, uint32_t N,
                           uint32_t S, void* stream) {
    MIPSF_REQUIRE(cfg && raw && z_vals && rgb && depth && target_rgb && target_d && counts && partial && sums && ticket, "null pointer");
    MIPSF_REQUIRE(S >= 1 && S <= MAX_S, "samples per ray %u outside [1,%d]", S, MAX_S);
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (hipMemsetAsync(sums, 0, 9 * sizeof(double), s) != hipSuccess) return 4;
        return 0;
    }
    const LossFinalize fin = {counts, ticket, nullptr, nullptr, nullptr, sums};
    if (S <= RT_MAX_S) {
        const int e = launch_render_train(raw, z_vals, target_rgb, target_d, to_render_cfg(*cfg), rgb, depth, depth_var, disp, acc, weights,
                                          partial, N, S, fin, nullptr, s);
        if (e > 0) return e;
        if (e == 0) return check_launch("render_fwd_sums");
    }
    constexpr int RPB = 16;
    hipLaunchKernelGGL((render_fwd_kernel<true, true, RPB>), dim3((N + RPB - 1) / RPB), dim3(RPB * MIPSF_WAVE), 0, s, raw, z_vals,
                       target_rgb, target_d, to_render_cfg(*cfg), rgb, depth, depth_var, disp, acc, weights, partial, N, S, fin);
    return check_launch("render_fwd_sums");
}

// losses[8] (and loss_total, as in mipsf_render_fwd) of a batch of N_total rays from its nine sums
int mipsf_loss_finalize_sums(const double* sums, const mipsf_render_cfg* cfg, uint32_t N_total, uint32_t S, float* losses,
                             const float* loss_weights, float* loss_total, void* stream) {
    MIPSF_REQUIRE(sums && cfg && losses, "null pointer");
    MIPSF_REQUIRE((loss_weights == nullptr) == (loss_total == nullptr), "loss_weights and loss_total come together");
    hipLaunchKernelGGL(loss_finalize_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, to_render_cfg(*cfg).emd_w,
                       N_total, S, losses, loss_weights, loss_total);
    return check_launch("loss_finalize_sums");
}

int mipsf_render_bwd(const mipsf_render_bwd_args* a_in, void* stream) {
    MIPSF_REQUIRE(a_in != nullptr, "null argument block");
    MIPSF_REQUIRE(a_in->struct_size == sizeof(mipsf_render_bwd_args) || a_in->struct_size == offsetof(mipsf_render_bwd_args, flags),
                  "mipsf_render_bwd_args: struct_size %u, this library expects %u", a_in->struct_size,
                  (unsigned)sizeof(mipsf_render_bwd_args));
    mipsf_render_bwd_args a_copy = {};
    memcpy(&a_copy, a_in, a_in->struct_size);
    const mipsf_render_bwd_args* a = &a_copy;
    const float* raw = a->raw; const float* z_vals = a->z_vals; const float* target_rgb = a->target_rgb; const float* target_d = a->target_d;
    const float* losses = a->losses; const mipsf_render_cfg* cfg = a->cfg; const float* g_losses = a->g_losses;
    const float* g_total = a->g_total; const float* loss_weights = a->loss_weights; const float* g_rgb = a->g_rgb;
    const float* g_depth = a->g_depth; float* draw = a->draw; const uint32_t N = a->N, S = a->S, N_norm = a->N_norm ? a->N_norm : a->N;
    if (N == 0) return 0;
    MIPSF_REQUIRE(N_norm >= N, "N_norm = %u: the normalising ray count cannot be below this launch's %u rays", N_norm, N);
    MIPSF_REQUIRE(cfg && raw && z_vals && draw, "null pointer");
    MIPSF_REQUIRE(S >= 1 && S <= MAX_S, "samples per ray %u outside [1,%d]", S, MAX_S);
    MIPSF_REQUIRE(g_total == nullptr || loss_weights != nullptr, "g_total needs the loss weights");
    const int train = g_losses != nullptr || g_total != nullptr;
    MIPSF_REQUIRE(!train || (target_rgb && target_d && losses), "training backward needs targets and losses");
    const int keep_if_unit = (a->flags & MIPSF_RENDER_BWD_KEEP_IF_UNIT) != 0;
    MIPSF_REQUIRE(!keep_if_unit || (g_total && !g_losses && !g_rgb && !g_depth && N_norm == N),
                  "KEEP_IF_UNIT: draw holds mipsf_render_fwd's gradient for g_total = 1 -- g_total must be the only gradient");
    hipLaunchKernelGGL(render_bwd_kernel, dim3((N + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK),
                       dim3(RAYS_PER_BLOCK * MIPSF_WAVE), 0, (hipStream_t)stream, raw, z_vals, target_rgb, target_d,
                       losses, to_render_cfg(*cfg), train, g_losses, g_rgb, g_depth, draw, N, S, g_total, loss_weights, N_norm, keep_if_unit);
    return check_launch("render_bwd");
}

int mipsf_rays_bwd(const float* dxn, const float* z_vals, const mipsf_render_cfg* cfg, float* d_rays_o,
                   float* d_rays_d, uint32_t N, uint32_t S, void* stream) {
    if (N == 0) return 0;
    MIPSF_REQUIRE(cfg && dxn && z_vals && d_rays_o && d_rays_d, "null pointer");
    hipLaunchKernelGGL(rays_bwd_kernel, dim3((N + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK),
                       dim3(RAYS_PER_BLOCK * MIPSF_WAVE), 0, (hipStream_t)stream, dxn, z_vals, make_norm(*cfg),
                       d_rays_o, d_rays_d, N, S);
    return check_launch("rays_bwd");
}

}  // extern "C"
