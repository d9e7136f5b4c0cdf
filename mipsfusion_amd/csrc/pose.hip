// Ray building from optimisable poses (reference: mipsfusion.py:320-322, 531-532 with
// helper_functions/geometry_helper.py:11-17 qt_to_transform_matrix and pytorch3d quaternion_to_matrix).
//
//   poses_all = [fixed poses (F) | quaternion+translation Parameters (K)]
//   rays_d[n] = R[owner[n]] * d_cam[n]        rays_o[n] = t[owner[n]]
//
// In the reference this is ~30 eager ops forward and ~120 backward per iteration (index_put with a sort in the
// gather's backward alone is 0.25 ms on MI355X); here it is one kernel each way plus a K-thread chain kernel, so
// the per-iteration pose path stops being launch-bound.  Gradients: dR[k] = sum_n g_d[n] (x) d_cam[n],
// dt[k] = sum_n g_o[n] over the rays owned by pose k, then the chain through R(q) = I + (2/|q|^2) A(q).
#include "pose_dev.h"

namespace mipsf {

MIPSF_SINGLE_FP32 __global__ __launch_bounds__(PR_BLOCK) void pose_rays_fwd_kernel(const float* __restrict__ fixed,
                                                                 const float* __restrict__ rot,
                                                                 const float* __restrict__ trans, int F, int K,
                                                                 const int64_t* __restrict__ owner,
                                                                 const float* __restrict__ d_cam,
                                                                 float* __restrict__ rays_o,
                                                                 float* __restrict__ rays_d, uint32_t N) {
    const uint32_t n = blockIdx.x * PR_BLOCK + threadIdx.x;
    if (n >= N) return;
    int64_t p = owner[n];
    if (p < 0) p += F + K;                       // python-style negative index (local_BA uses -1 = current frame)
    const bool in_range = p >= 0 && p < F + K;   // a bad owner gives NaN rays instead of reading a foreign pose
    const Mat34 m = load_pose(fixed, rot, trans, F, in_range ? (int)p : 0);
    const float dx = in_range ? d_cam[3 * n] : __builtin_nanf(""), dy = d_cam[3 * n + 1], dz = d_cam[3 * n + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        rays_d[3 * n + j] = (dx * m.r[3 * j] + dy * m.r[3 * j + 1]) + dz * m.r[3 * j + 2];
        rays_o[3 * n + j] = m.t[j];
    }
}

// The row gather of the ray table (mipsf_gather_rays, ro.hip) and the ray construction in ONE launch: both are one
// thread per ray and each was a 5 us launch of every iteration.  Writes d_cam / rgb / depth (the gather's outputs: d_cam
// is what the backward needs) and rays_o / rays_d; same arithmetic and the same NaN conventions as the two kernels.
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(PR_BLOCK) void gather_pose_rays_fwd_kernel(
    const float* __restrict__ db, uint64_t n_rows, const int64_t* __restrict__ idx, const float* __restrict__ fixed,
    const float* __restrict__ rot, const float* __restrict__ trans, int F, int K, const int64_t* __restrict__ owner,
    float* __restrict__ d_cam, float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ rays_o,
    float* __restrict__ rays_d, uint32_t N) {
    const uint32_t n = blockIdx.x * PR_BLOCK + threadIdx.x;
    if (n >= N) return;
    int64_t r = idx[n];
    if (r < 0) r += (int64_t)n_rows;
    const bool row_ok = r >= 0 && (uint64_t)r < n_rows;
    const float* s = db + 7 * (size_t)(row_ok ? r : 0);
    float v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = row_ok ? s[k] : __builtin_nanf("");
    d_cam[3 * (size_t)n] = v[0], d_cam[3 * (size_t)n + 1] = v[1], d_cam[3 * (size_t)n + 2] = v[2];
    rgb[3 * (size_t)n] = v[3], rgb[3 * (size_t)n + 1] = v[4], rgb[3 * (size_t)n + 2] = v[5];
    depth[n] = v[6];
    int64_t p = owner[n];
    if (p < 0) p += F + K;
    const bool in_range = p >= 0 && p < F + K;
    const Mat34 m = load_pose(fixed, rot, trans, F, in_range ? (int)p : 0);
    const float dx = in_range ? v[0] : __builtin_nanf(""), dy = v[1], dz = v[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        rays_d[3 * n + j] = (dx * m.r[3 * j] + dy * m.r[3 * j + 1]) + dz * m.r[3 * j + 2];
        rays_o[3 * n + j] = m.t[j];
    }
}

// ONE launch (it used to be zero-fill + accumulate-with-atomics + chain = three, ~5 us each in a captured iteration):
// every workgroup writes its per-pose partial {dR, dt} to its own row of `part`, the LAST one to finish (ticket
// counter) sums the rows in workgroup order -- no float atomics on global memory --, runs the
// quaternion chain and puts the ticket back to zero for the next call.
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(PR_BLOCK) void pose_rays_bwd_kernel(const float* __restrict__ g_o,
                                                                 const float* __restrict__ g_d,
                                                                 const float* __restrict__ d_cam,
                                                                 const int64_t* __restrict__ owner,
                                                                 const float* __restrict__ rot, int F, int K,
                                                                 float* __restrict__ part, uint32_t* __restrict__ ticket,
                                                                 float* __restrict__ d_rot, float* __restrict__ d_trans,
                                                                 uint32_t N, int accumulate) {
    __shared__ float sacc[PR_MAX_POSES * 12];
    __shared__ bool is_last;
    const int P = F + K;
    for (int q = threadIdx.x; q < P * 12; q += PR_BLOCK) sacc[q] = 0.f;
    __syncthreads();
    const uint32_t n = blockIdx.x * PR_BLOCK + threadIdx.x;
    const bool valid = n < N;
    const uint32_t nn = valid ? n : N - 1;
    int64_t p = owner[nn];
    if (p < 0) p += P;
    if (p < 0 || p >= P) p = 0;                  // forward already produced NaN rays for this owner
    float v[12];
    const float dx = d_cam[3 * nn], dy = d_cam[3 * nn + 1], dz = d_cam[3 * nn + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g = (valid && g_d) ? g_d[3 * nn + j] : 0.f;
        v[3 * j] = g * dx, v[3 * j + 1] = g * dy, v[3 * j + 2] = g * dz;
        v[9 + j] = (valid && g_o) ? g_o[3 * nn + j] : 0.f;
    }
    // Rays of one pose are contiguous in practice.  Every wave reduces its rays pose by pose (one pass per distinct owner: one
    // in practice, two at a pose boundary) with the fixed tree of wave_sum and leaves the sums in ITS OWN slots; one thread
    // per entry then adds the waves' slots in wave order.  (LDS float atomics from four waves arrive in any order: the pose
    // gradients of two identical launches used to differ in the last bit -- tools/dbg_tracking_determinism.py.)  A wave
    // with more than PR_MAXSEG distinct owners (scattered owners: not the reference's batches) falls back to atomics.
    constexpr int PR_WAVES = PR_BLOCK / 64, PR_MAXSEG = 4;
    __shared__ float wv[PR_WAVES][PR_MAXSEG][12];
    __shared__ int wp[PR_WAVES][PR_MAXSEG], wn[PR_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        unsigned long long remaining = __ballot(1);
        int nseg = 0;
        while (remaining) {
            const int src = __ffsll((long long)remaining) - 1;
            const int64_t pv = __shfl(p, src, 64);
            const bool mine = p == pv;
            const unsigned long long mask = __ballot(mine) & remaining;
            remaining &= ~mask;
            if (nseg < PR_MAXSEG) {
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const float sq = wave_sum(mine ? v[q] : 0.0f);
                    if (lane == 0) wv[w][nseg][q] = sq;
                }
                if (lane == 0) wp[w][nseg] = (int)pv;
                ++nseg;
            } else if (mine) {
#pragma unroll
                for (int q = 0; q < 12; ++q) atomicAdd(&sacc[(int)p * 12 + q], v[q]);
            }
        }
        if (lane == 0) wn[w] = nseg < PR_MAXSEG ? nseg : PR_MAXSEG;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        for (int ww = 0; ww < PR_WAVES; ++ww)
            for (int sg = 0; sg < wn[ww]; ++sg) sacc[wp[ww][sg] * 12 + (int)threadIdx.x] += wv[ww][sg][threadIdx.x];
    }
    pose_block_finish<PR_BLOCK>(sacc, &is_last, P, F, K, part, ticket, rot, d_rot, d_trans, accumulate != 0);
}

// One pose handed from one stage of a frame to the next ON THE DEVICE.  The reference passes a 4x4 between its stages
// (mipsfusion.py:479-501: RandomOptimizer -> tracking iterations; :556-575 -> local BA), i.e. every stage starts from
// matrix_to_quaternion of the previous stage's matrix -- a UNIT quaternion, also when the previous stage's Adam steps left
// its own quaternion slightly off the sphere.  Same here, without the host round trip:
//   src_kind 0: src = [3x3 rotation row-major | translation] (words 0..11 of the RandomOptimizer's device state)
//   src_kind 1: src = [w x y z | tx ty tz] (a stage's pose Parameters): qt_to_transform_matrix first (geometry_helper.py:11-17)
// then geometry_helper.matrix_to_quaternion (:20-33), operation by operation in IEEE fp32 (no contraction: the file is built
// with -ffp-contract=off; square roots and divisions through fp64, whose rounded result is the correctly rounded fp32 one:
// 53 >= 2 * 24 + 2 bits), so the result equals the frame loop's host helpers (sequence._qt_to_matrix_np,
// _matrix_to_quaternion_np) bit for bit; torch's own CPU kernels differ from IEEE in the last bit on some hosts.
MIPSF_SINGLE_FP32 __global__ void pose_handover_kernel(const float* __restrict__ src, int src_kind, float* __restrict__ rot, float* __restrict__ trans) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float a, b, c, d, e, f, g, h, i, tx, ty, tz;
    if (src_kind == 0) {
        a = src[0], b = src[1], c = src[2], d = src[3], e = src[4], f = src[5], g = src[6], h = src[7], i = src[8];
        tx = src[9], ty = src[10], tz = src[11];
    } else {
        const float w = src[0], x = src[1], y = src[2], z = src[3];
        const float k = (float)(2.0 / (double)(((w * w + x * x) + y * y) + z * z));
        a = 1.0f - k * (y * y + z * z), b = k * (x * y - z * w), c = k * (x * z + y * w);
        d = k * (x * y + z * w), e = 1.0f - k * (x * x + z * z), f = k * (y * z - x * w);
        g = k * (x * z - y * w), h = k * (y * z + x * w), i = 1.0f - k * (x * x + y * y);
        tx = src[4], ty = src[5], tz = src[6];
    }
    const float m2[4] = {((1.0f + a) + e) + i, ((1.0f + a) - e) - i, ((1.0f - a) + e) - i, ((1.0f - a) - e) + i};
    float mag[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mag[k] = m2[k] > 0.0f ? (float)sqrt((double)m2[k]) : 0.0f;
    int best = 0;                                    // (argmax: the first of equal maxima, as torch / numpy)
#pragma unroll
    for (int k = 1; k < 4; ++k) best = mag[k] > mag[best] ? k : best;
    float t[4];
    if (best == 0) t[0] = mag[0] * mag[0], t[1] = h - f, t[2] = c - g, t[3] = d - b;
    else if (best == 1) t[0] = h - f, t[1] = mag[1] * mag[1], t[2] = d + b, t[3] = c + g;
    else if (best == 2) t[0] = c - g, t[1] = d + b, t[2] = mag[2] * mag[2], t[3] = f + h;
    else t[0] = d - b, t[1] = g + c, t[2] = h + f, t[3] = mag[3] * mag[3];
    const float den = 2.0f * fmaxf(mag[best], 0.1f);
    float q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (float)((double)t[k] / (double)den);
    const bool neg = q[0] < 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) rot[k] = neg ? -q[k] : q[k];
    trans[0] = tx, trans[1] = ty, trans[2] = tz;
}

__global__ void pose_zero_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

}  // namespace mipsf

using namespace mipsf;

extern "C" {

static int pose_rays_fwd_plain(const float* fixed_poses, const float* rot, const float* trans, uint32_t F, uint32_t K,
                               const int64_t* owner, const float* d_cam, float* rays_o, float* rays_d, uint32_t N,
                               void* stream) {
    if (N == 0) return 0;
    MIPSF_REQUIRE(owner && d_cam && rays_o && rays_d, "null pointer");
    MIPSF_REQUIRE((F == 0 || fixed_poses) && (K == 0 || (rot && trans)), "null pose pointer");
    MIPSF_REQUIRE(F + K >= 1 && F + K <= PR_MAX_POSES, "number of poses %u outside [1,%d]", F + K, PR_MAX_POSES);
    hipLaunchKernelGGL(pose_rays_fwd_kernel, dim3((N + PR_BLOCK - 1) / PR_BLOCK), dim3(PR_BLOCK), 0,
                       (hipStream_t)stream, fixed_poses, rot, trans, (int)F, (int)K, owner, d_cam, rays_o, rays_d, N);
    return check_launch("pose_rays_fwd");
}

// db != NULL: the rows idx[N] of the ray table are gathered first (d_cam, rgb, depth are written); db NULL: d_cam is an input
int mipsf_pose_rays_fwd(const float* db, uint64_t n_rows, const int64_t* idx, const float* fixed_poses,
                        const float* rot, const float* trans, uint32_t F, uint32_t K, const int64_t* owner,
                        float* d_cam, float* rgb, float* depth, float* rays_o, float* rays_d, uint32_t N,
                        void* stream) {
    if (db == nullptr) return pose_rays_fwd_plain(fixed_poses, rot, trans, F, K, owner, d_cam, rays_o, rays_d, N, stream);
    if (N == 0) return 0;
    MIPSF_REQUIRE(db && idx && owner && d_cam && rgb && depth && rays_o && rays_d, "null pointer");
    MIPSF_REQUIRE((F == 0 || fixed_poses) && (K == 0 || (rot && trans)), "null pose pointer");
    MIPSF_REQUIRE(F + K >= 1 && F + K <= PR_MAX_POSES, "number of poses %u outside [1,%d]", F + K, PR_MAX_POSES);
    hipLaunchKernelGGL(gather_pose_rays_fwd_kernel, dim3((N + PR_BLOCK - 1) / PR_BLOCK), dim3(PR_BLOCK), 0,
                       (hipStream_t)stream, db, n_rows, idx, fixed_poses, rot, trans, (int)F, (int)K, owner, d_cam, rgb,
                       depth, rays_o, rays_d, N);
    return check_launch("gather_pose_rays_fwd");
}

}  // extern "C"
namespace mipsf {
uint64_t pose_rays_scratch_floats(uint32_t F, uint32_t K, uint32_t N) {
    return 1ull + 12ull * (F + K) * ((N + PR_BLOCK - 1) / PR_BLOCK);
}
}  // namespace mipsf
extern "C" {

int mipsf_pose_handover(const float* src, int src_kind, float* rot, float* trans, void* stream) {
    MIPSF_REQUIRE(src && rot && trans, "null pointer");
    MIPSF_REQUIRE(src_kind == MIPSF_POSE_MATRIX || src_kind == MIPSF_POSE_QUATERNION, "src_kind %d", src_kind);
    MIPSF_REQUIRE(src_kind == MIPSF_POSE_MATRIX || (src != rot && src + 4 != trans), "a quaternion source must not be the destination");
    hipLaunchKernelGGL(pose_handover_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, src_kind, rot, trans);
    return check_launch("pose_handover");
}

int mipsf_pose_rays_bwd(const float* g_rays_o, const float* g_rays_d, const float* rot, uint32_t F, uint32_t K,
                        const int64_t* owner, const float* d_cam, float* d_rot, float* d_trans, float* scratch,
                        uint32_t N, int accumulate, void* stream) {
    MIPSF_REQUIRE(K >= 1, "no optimisable pose");
    MIPSF_REQUIRE(rot && owner && d_cam && d_rot && d_trans && scratch, "null pointer");
    MIPSF_REQUIRE(F + K <= PR_MAX_POSES, "number of poses %u above %d", F + K, PR_MAX_POSES);
    hipStream_t s = (hipStream_t)stream;
    if (N == 0 && accumulate) return 0;       // nothing to add
    if (N == 0) {       // no rays: zero gradients
        hipLaunchKernelGGL(pose_zero_kernel, dim3(1), dim3(256), 0, s, d_rot, (int)(4 * K));
        hipLaunchKernelGGL(pose_zero_kernel, dim3(1), dim3(256), 0, s, d_trans, (int)(3 * K));
        return check_launch("pose_zero");
    }
    // scratch[0] = ticket (zero on entry, zero again on return), then one row of partials per workgroup
    hipLaunchKernelGGL(pose_rays_bwd_kernel, dim3((N + PR_BLOCK - 1) / PR_BLOCK), dim3(PR_BLOCK), 0, s, g_rays_o,
                       g_rays_d, d_cam, owner, rot, (int)F, (int)K, scratch + 1, reinterpret_cast<uint32_t*>(scratch),
                       d_rot, d_trans, N, accumulate);
    return check_launch("pose_rays_bwd");
}

}  // extern "C"
