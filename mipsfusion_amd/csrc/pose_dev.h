// Pose helpers shared by pose.hip (ray construction) and render.hip (the fused gather + pose + placement kernels).
#pragma once
#include "common.h"

namespace mipsf {

constexpr int PR_BLOCK = 256;
constexpr int PR_MAX_POSES = 64;

struct Mat34 {
    float r[9];
    float t[3];
};

__device__ __forceinline__ Mat34 load_pose(const float* __restrict__ fixed, const float* __restrict__ rot,
                                           const float* __restrict__ trans, int F, int p) {
    Mat34 m;
    if (p < F) {
        const float* s = fixed + 16 * p;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int i = 0; i < 3; ++i) m.r[3 * j + i] = s[4 * j + i];
            m.t[j] = s[4 * j + 3];
        }
    } else {
        const float* q = rot + 4 * (p - F);
        const float w = q[0], x = q[1], y = q[2], z = q[3];
        const float s = 2.0f / (w * w + x * x + y * y + z * z);
        m.r[0] = 1 - s * (y * y + z * z), m.r[1] = s * (x * y - z * w), m.r[2] = s * (x * z + y * w);
        m.r[3] = s * (x * y + z * w), m.r[4] = 1 - s * (x * x + z * z), m.r[5] = s * (y * z - x * w);
        m.r[6] = s * (x * z - y * w), m.r[7] = s * (y * z + x * w), m.r[8] = 1 - s * (x * x + y * y);
        const float* t = trans + 3 * (p - F);
        m.t[0] = t[0], m.t[1] = t[1], m.t[2] = t[2];
    }
    return m;
}

// chain through R(q) = I + s A(q), s = 2/|q|^2 (pytorch3d quaternion_to_matrix, not assuming unit norm);
// G = {dR (9, row-major), dt (3)} of optimisable pose k
__device__ __forceinline__ void pose_chain(const float* __restrict__ rot, const float* G, int k,
                                           float* __restrict__ d_rot, float* __restrict__ d_trans, bool accumulate = false) {
    const float w = rot[4 * k], x = rot[4 * k + 1], y = rot[4 * k + 2], z = rot[4 * k + 3];
    const float n = w * w + x * x + y * y + z * z;
    const float s = 2.0f / n;
    const float A[9] = {-(y * y + z * z), x * y - z * w, x * z + y * w, x * y + z * w, -(x * x + z * z),
                        y * z - x * w,    x * z - y * w, y * z + x * w, -(x * x + y * y)};
    float GA = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) GA += G[i] * A[i];
    const float dAw = (-z * G[1] + y * G[2]) + (z * G[3] - x * G[5]) + (-y * G[6] + x * G[7]);
    const float dAx = (y * G[1] + z * G[2]) + (y * G[3] - 2 * x * G[4] - w * G[5]) + (z * G[6] + w * G[7] - 2 * x * G[8]);
    const float dAy = (-2 * y * G[0] + x * G[1] + w * G[2]) + (x * G[3] + z * G[5]) + (-w * G[6] + z * G[7] - 2 * y * G[8]);
    const float dAz = (-2 * z * G[0] - w * G[1] + x * G[2]) + (w * G[3] - 2 * z * G[4] + y * G[5]) + (x * G[6] + y * G[7]);
    const float c = s * GA * 2.0f / n;
    const float r0 = s * dAw - c * w, r1 = s * dAx - c * x, r2 = s * dAy - c * y, r3 = s * dAz - c * z;
    if (accumulate) {       // `.grad +=` semantics: the caller passed the parameters' gradient buffers themselves
        d_rot[4 * k] += r0, d_rot[4 * k + 1] += r1, d_rot[4 * k + 2] += r2, d_rot[4 * k + 3] += r3;
        d_trans[3 * k] += G[9], d_trans[3 * k + 1] += G[10], d_trans[3 * k + 2] += G[11];
    } else {
        d_rot[4 * k] = r0, d_rot[4 * k + 1] = r1, d_rot[4 * k + 2] = r2, d_rot[4 * k + 3] = r3;
        d_trans[3 * k] = G[9], d_trans[3 * k + 1] = G[10], d_trans[3 * k + 2] = G[11];
    }
}

// The end of a pose-gradient kernel: this workgroup's per-pose partial {dR, dt} rows (sacc, LDS, P x 12) go to its row of
// `part`; the LAST workgroup through the ticket sums the rows in workgroup order, runs the quaternion chain and puts the
// ticket back to zero.  T = threads of the workgroup; called by every thread.
template <int T>
__device__ __forceinline__ void pose_block_finish(float* sacc, bool* is_last, int P, int F, int K, float* __restrict__ part,
                                                  uint32_t* __restrict__ ticket, const float* __restrict__ rot,
                                                  float* __restrict__ d_rot, float* __restrict__ d_trans, bool accumulate) {
#ifdef RT_ABL_NO_TAIL
    return;
#endif
    __syncthreads();
    for (int q = threadIdx.x; q < P * 12; q += T)
        __hip_atomic_store(&part[(size_t)blockIdx.x * (P * 12) + q], sacc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Hand-off form: "sc1 write-through stores + s_waitcnt vmcnt(0) + flag" on the producer side, sc1 loads on the
    // consumer side (MI355X_MICROARCH.md, inter-workgroup visibility: `__hip_atomic_store/load(relaxed, agent)` lower to
    // `global_store / global_load ... sc1`, which bypass the CU's L1 and leave no dirty line in the XCD's L2, so there
    // is nothing for an agent-scope release to write back and nothing stale for an acquire to invalidate; "sc1 loads may
    // replace the acquire only when the producer stored sc1" -- both sides do).  What the ticket needs is that this
    // workgroup's row stores have been ACKNOWLEDGED before its increment is issued: every wave waits for its own stores
    // (the explicit asm: the compiler may drop a fence's wait when it believes the counter is empty, and inline asm is
    // invisible to that pass), the barrier collects the waves, one lane takes the ticket with a device-scope atomic.
    // An agent-scope __threadfence() instead writes back and invalidates the XCD's whole L2 on this multi-XCD part (the
    // same pattern took a scatter kernel from 61 to 690 us; this kernel: 15 -> ~8 us for 16 workgroups).
    // Stress-tested under uneven load with 400 workgroups x 2000 alternating calls, every word checked
    // (tests/test_gpu_parity.py::test_pose_rays_bwd_ticket_reduction_under_load).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        *is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!*is_last) return;
    // All threads read: thread (g, q) = (tid / W, tid % W) adds entry q of the rows g, g + G, g + 2G, ... with eight loads
    // in flight, then the G partial sums of an entry are added in group order (a fixed order: the result does not depend
    // on which workgroup is last).  (One thread per entry walking all rows is one dependent ~1 us load after the other:
    // 256 rows of a 16-ray-per-workgroup grid took 12 us that way.)
    __shared__ float red[T];
    {
        const int W = P * 12;
        const int G = W <= T ? T / W : 1;             // row groups (1: more entries than threads, every thread walks all rows)
        auto rows_sum = [&](int q, uint32_t first, uint32_t step) {
            float s = 0.f;
            uint32_t b = first;
            for (; b + 7u * step < gridDim.x; b += 8u * step) {
                float r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    r[u] = __hip_atomic_load(&part[(size_t)(b + (uint32_t)u * step) * W + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += r[u];
            }
            for (; b < gridDim.x; b += step) s += __hip_atomic_load(&part[(size_t)b * W + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return s;
        };
        if (G == 1) {
            for (int q = threadIdx.x; q < W; q += T) sacc[q] = rows_sum(q, 0u, 1u);
        } else {
            const int g = (int)threadIdx.x / W, q = (int)threadIdx.x % W;
            red[threadIdx.x] = g < G ? rows_sum(q, (uint32_t)g, (uint32_t)G) : 0.f;
            __syncthreads();
            if ((int)threadIdx.x < W) {
                float t = 0.f;
                for (int k = 0; k < G; ++k) t += red[k * W + (int)threadIdx.x];
                sacc[threadIdx.x] = t;
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += T) pose_chain(rot, sacc + 12 * (F + k), k, d_rot, d_trans, accumulate);
    if (threadIdx.x == 0) *ticket = 0u;
}

}  // namespace mipsf
