// RandomOptimizer particle step (reference RandomOptimizer.py:54-88, 113-131, 137-224).
//
// One tracking round of the reference is ~25 eager torch ops, a [P,3,3] @ [3,n] batched matmul, a device->host
// synchronisation (`if success_flag:`) and two small host->device copies.  Here the state of the search (rotation,
// translation, search size) lives in a 32-float device buffer and a round is five launches with no host round
// trip, so that all rounds of a frame can be captured in one hipGraph:
//   ro_particles   PST rescale -> 7-D pose -> R(q) -> absolute pose -> lattice points to world -> fp64 normalisation
//   hashgrid_fwd, decoder_fwd (forward only), ro_fitness   (existing kernels)
//   ro_update      advanced-particle weights, weighted mean transform, pose and search-size update (one workgroup)
#include "common.h"

namespace mipsf {

// state layout (floats): see MIPSF_RO_* in include/mipsf.h
constexpr int RO_ROT = 0, RO_TRANS = 9, RO_SEARCH = 12, RO_SUCCESS = 18, RO_MEAN_SDF = 19, RO_FIT0 = 20,
              RO_MEAN_T = 21, RO_NBETTER = 28;

// pytorch3d.transforms.quaternion_to_matrix for a real-first, not necessarily unit quaternion
__device__ __forceinline__ void quat_to_mat(float w, float x, float y, float z, float (&m)[9]) {
    const float two_s = 2.0f / (((w * w + x * x) + y * y) + z * z);
    m[0] = 1 - two_s * (y * y + z * z), m[1] = two_s * (x * y - z * w), m[2] = two_s * (x * z + y * w);
    m[3] = two_s * (x * y + z * w), m[4] = 1 - two_s * (x * x + z * z), m[5] = two_s * (y * z - x * w);
    m[6] = two_s * (x * z - y * w), m[7] = two_s * (y * z + x * w), m[8] = 1 - two_s * (x * x + y * y);
}

__device__ __forceinline__ void mat_mul3(const float (&a)[9], const float (&b)[9], float (&c)[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

#ifdef MIPSF_RO_LANE_CHECK      // diagnosis build (tools/dbg_ro_lanes.py): which intermediate of the pose section differs between lanes
__device__ unsigned ro_chk_count;
__device__ float ro_chk_dump[32][64][24];
#endif

// one wave per particle; every lane derives the particle's pose (60 flops) and then walks the lattice points
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(256) void ro_particles_kernel(const float* __restrict__ pst,
                                                           const float* __restrict__ state,
                                                           const float* __restrict__ rays_d_cam,
                                                           const float* __restrict__ target_d, NormCfg nc,
                                                           float* __restrict__ xn, float* __restrict__ pst7,
                                                           uint32_t P, uint32_t n, int point_major) {
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) / MIPSF_WAVE;
    const uint32_t lane = threadIdx.x & (MIPSF_WAVE - 1);
    if (p >= P) return;
    float r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = pst[6 * (size_t)p + k] * state[RO_SEARCH + k];
    const float s = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
    const float qw = s <= 1.0f ? sqrtf(1.0f - s) : 0.0f;             // pose_6D_to_7D
    if (lane == 0) {
        float* o = pst7 + 7 * (size_t)p;
        o[0] = qw;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[1 + k] = r[k];
    }
    float rot[9], dR[9], aR[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) rot[k] = state[RO_ROT + k];
    quat_to_mat(qw, r[0], r[1], r[2], dR);
    mat_mul3(rot, dR, aR);                                            // get_abs_pose
    float t0 = state[RO_TRANS] + r[3], t1 = state[RO_TRANS + 1] + r[4], t2 = state[RO_TRANS + 2] + r[5];
    // The particle's pose is the same in all 64 lanes by construction; it is taken from lane 0 explicitly.  The section above
    // holds packed multiplies (v_pk_mul_f32 of r[] by the search size), and with a second process's wavefronts on the same CUs
    // their high halves can come back wrong in lanes 48..63 -- never in the first lanes (the note in the loop below; round 3 saw
    // ~55 of 800 frames leave this section with another pose in those lanes).
#ifdef MIPSF_RO_LANE_CHECK
    {
        unsigned long long differ = 0ull;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            differ |= __ballot(__float_as_uint(aR[k]) != (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(aR[k])));
        differ |= __ballot(__float_as_uint(t0) != (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(t0)));
        differ |= __ballot(__float_as_uint(t1) != (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(t1)));
        differ |= __ballot(__float_as_uint(t2) != (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(t2)));
        if (differ) {
            unsigned slot = 0;
            if (lane == 0) slot = atomicAdd(&ro_chk_count, 1u);
            slot = (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
            if (slot < 32u) {
                float* o = ro_chk_dump[slot][lane];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = r[k];
                o[6] = s, o[7] = qw;
#pragma unroll
                for (int k = 0; k < 9; ++k) o[8 + k] = aR[k];
                o[17] = t0, o[18] = t1, o[19] = t2;
                o[20] = pst[6 * (size_t)p], o[21] = state[RO_SEARCH], o[22] = dR[0], o[23] = (float)p;
            }
        }
    }
#endif
#pragma unroll
    for (int k = 0; k < 9; ++k) aR[k] = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(aR[k])));
    t0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(t0)));
    t1 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(t1)));
    t2 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(t2)));
#if MIPSF_RO_PACKED == 2   // reproducer build: the rotation back in VGPRs (hipcc then packs the multiplies from VGPR pairs)
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("v_mov_b32 %0, %0" : "+v"(aR[k]));
    asm volatile("v_mov_b32 %0, %0" : "+v"(t0));
    asm volatile("v_mov_b32 %0, %0" : "+v"(t1));
    asm volatile("v_mov_b32 %0, %0" : "+v"(t2));
#endif
    for (uint32_t i = lane; i < n; i += MIPSF_WAVE) {
        const float d = target_d[i];
        const float c0 = rays_d_cam[3 * i] * d, c1 = rays_d_cam[3 * i + 1] * d, c2 = rays_d_cam[3 * i + 2] * d;
#if MIPSF_RO_PACKED >= 3   // diagnosis builds: hipcc's instruction sequence for the transform verbatim (3), and with ONE change each:
                           // 4 the packed add with the crossed op_sel as two single adds; 5 the packed multiply that feeds it as two
                           // single multiplies; 6 verbatim after 32 idle cycles; 7 the packed multiply's result copied before use
        float w0, w1, w2;
        {
            const float r0 = rays_d_cam[3 * i], r1 = rays_d_cam[3 * i + 1], r2 = rays_d_cam[3 * i + 2];
#define RO_S(k) (int)__float_as_uint(aR[k])
            asm volatile(
                "s_mov_b32 s68, %[a1]\n s_mov_b32 s69, %[a3]\n s_mov_b32 s62, %[a4]\n s_mov_b32 s63, %[a0]\n"
                "s_mov_b32 s70, %[a2]\n s_mov_b32 s71, %[a5]\n s_mov_b32 s73, %[a6]\n s_mov_b32 s82, %[a7]\n"
                "s_mov_b32 s83, %[a8]\n s_mov_b32 s74, %[t0]\n s_mov_b32 s75, %[t1]\n s_mov_b32 s84, %[t2]\n"
                "v_mov_b32 v112, %[d]\n v_mov_b32 v113, %[r2]\n v_mov_b32 v114, %[r1]\n v_mov_b32 v115, %[r0]\n v_mov_b32 v117, 0\n"
#if MIPSF_RO_PACKED == 6
                "s_nop 15\n s_nop 15\n"
#endif
                "v_mul_f32_e32 v116, v112, v113\n"
                "v_pk_mul_f32 v[112:113], v[112:113], v[114:115] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[114:115], s[68:69], v[112:113]\n"
#if MIPSF_RO_PACKED == 5
                "v_mul_f32_e32 v118, s62, v112\n v_mul_f32_e32 v119, s63, v113\n"
#else
                "v_pk_mul_f32 v[118:119], s[62:63], v[112:113]\n"
#endif
                "v_mul_f32_e32 v113, s73, v113\n"
#if MIPSF_RO_PACKED == 4
                "v_add_f32_e32 v114, v114, v119\n v_add_f32_e32 v115, v115, v118\n"
#elif MIPSF_RO_PACKED == 7
                "v_mov_b32 v120, v119\n v_mov_b32 v121, v118\n"
                "v_pk_add_f32 v[114:115], v[114:115], v[120:121]\n"
#else
                "v_pk_add_f32 v[114:115], v[114:115], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n"
#endif
                "v_pk_mul_f32 v[118:119], s[70:71], v[116:117] op_sel_hi:[1,0]\n"
                "v_mul_f32_e32 v112, s82, v112\n"
                "v_pk_add_f32 v[114:115], v[114:115], v[118:119]\n"
                "v_add_f32_e32 v112, v113, v112\n"
                "v_mul_f32_e32 v113, s83, v116\n"
                "v_pk_add_f32 v[114:115], s[74:75], v[114:115]\n"
                "v_add_f32_e32 v112, v112, v113\n"
                "v_add_f32_e32 v130, s84, v112\n"
                "v_mov_b32 %[o0], v114\n v_mov_b32 %[o1], v115\n v_mov_b32 %[o2], v130\n"
                : [o0] "=v"(w0), [o1] "=v"(w1), [o2] "=v"(w2)
                : [d] "v"(d), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [a0] "s"(RO_S(0)), [a1] "s"(RO_S(1)), [a2] "s"(RO_S(2)),
                  [a3] "s"(RO_S(3)), [a4] "s"(RO_S(4)), [a5] "s"(RO_S(5)), [a6] "s"(RO_S(6)), [a7] "s"(RO_S(7)), [a8] "s"(RO_S(8)),
                  [t0] "s"((int)__float_as_uint(t0)), [t1] "s"((int)__float_as_uint(t1)), [t2] "s"((int)__float_as_uint(t2))
                : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v130", "s62", "s63", "s68", "s69",
                  "s70", "s71", "s73", "s74", "s75", "s82", "s83", "s84");
#undef RO_S
        }
#elif MIPSF_RO_PACKED      // reproducer builds (tools/micro/ro_diag.sh; build with -DMIPSF_KEEP_PACKED_FP32 as well): the products as hipcc packs them
        const float w0 = ((aR[0] * c0 + aR[1] * c1) + aR[2] * c2) + t0;     // batch_points_trans
        const float w1 = ((aR[3] * c0 + aR[4] * c1) + aR[5] * c2) + t1;
        const float w2 = ((aR[6] * c0 + aR[7] * c1) + aR[8] * c2) + t2;
#else
        // The nine products are issued as single v_mul_f32.  Left to hipcc they become v_pk_mul_f32 pairs, and with a SECOND
        // PROCESS's wavefronts on the same CUs the high half of such a product came back as 0 in lanes 48..63 of about one
        // wavefront-iteration in 10^6 (182 of 12000 launches with the packed form -- SGPR-pair or VGPR-pair sources
        // alike --, 0 of 12000 with this one, same box, alternating builds; never with one process on the GPU).
        float m[9];
#define RO_MUL(k, c) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m[k]) : "s"(aR[k]), "v"(c))
        RO_MUL(0, c0); RO_MUL(1, c1); RO_MUL(2, c2); RO_MUL(3, c0); RO_MUL(4, c1); RO_MUL(5, c2); RO_MUL(6, c0); RO_MUL(7, c1); RO_MUL(8, c2);
#undef RO_MUL
        const float w0 = ((m[0] + m[1]) + m[2]) + t0, w1 = ((m[3] + m[4]) + m[5]) + t1, w2 = ((m[6] + m[7]) + m[8]) + t2;   // batch_points_trans
#endif
        // point_major: sample index = point * P + particle -- the 64 samples of a hash-grid wavefront are then 64
        // particles' copies of ONE lattice point (a few cm apart: same or neighbouring cells on every level)
        // instead of 64 lattice points scattered over the depth image
        float* o = xn + 3 * (point_major ? (size_t)i * P + p : (size_t)p * n + i);
        o[0] = normalise1(w0, nc.sub[0], nc.div[0], nc.norm_factor);          // run_network's fp64 normalisation
        o[1] = normalise1(w1, nc.sub[1], nc.div[1], nc.norm_factor);
        o[2] = normalise1(w2, nc.sub[2], nc.div[2], nc.norm_factor);
    }
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
    return t;
}

// RandomOptimizer.py:196-224, one workgroup.  Sums are accumulated in fp64 (any fp32 summation order of the
// reference is within one rounding of them).
MIPSF_SINGLE_FP32 __global__ __launch_bounds__(256) void ro_update_kernel(const float* __restrict__ mean_masked,
                                                        const float* __restrict__ pst7, float* __restrict__ state,
                                                        float sdf_weight, float rescale, uint32_t P) {
    __shared__ double sh[4];
    const float f0 = mean_masked[0] * sdf_weight;
    double s_w = 0.0, s_wm = 0.0, s_t[7] = {0, 0, 0, 0, 0, 0, 0}, s_n = 0.0;
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
        const float mm = mean_masked[p];
        const float f = mm * sdf_weight;
        const float w = f < f0 ? f0 - f : 0.0f;
        s_n += f < f0 ? 1.0 : 0.0;
        s_w += (double)w;
        s_wm += (double)(w * mm);
#pragma unroll
        for (int k = 0; k < 7; ++k) s_t[k] += (double)(pst7[7 * (size_t)p + k] * w);
    }
    s_n = block_sum_d(s_n, sh);
    s_w = block_sum_d(s_w, sh);
    s_wm = block_sum_d(s_wm, sh);
#pragma unroll
    for (int k = 0; k < 7; ++k) s_t[k] = block_sum_d(s_t[k], sh);
    if (threadIdx.x != 0) return;
    const float wsum = (float)s_w + 0.00001f;
    const bool ok = s_n > 0.0;
    float mt[7] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mean_sdf = mean_masked[0];
    if (ok) {
        mean_sdf = (float)s_wm / wsum;
#pragma unroll
        for (int k = 0; k < 7; ++k) mt[k] = (float)s_t[k] / wsum;
        const float nq = sqrtf(((mt[0] * mt[0] + mt[1] * mt[1]) + mt[2] * mt[2]) + mt[3] * mt[3]) + 1e-5f;
#pragma unroll
        for (int k = 0; k < 4; ++k) mt[k] = mt[k] / nq;
        float rot[9], dR[9], nr[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) rot[k] = state[RO_ROT + k];
        quat_to_mat(mt[0], mt[1], mt[2], mt[3], dR);
        mat_mul3(rot, dR, nr);                                        // update_cur_pose
#pragma unroll
        for (int k = 0; k < 9; ++k) state[RO_ROT + k] = nr[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) state[RO_TRANS + k] = state[RO_TRANS + k] + mt[4 + k];
    }
    float sz[6], n2 = 0.f;                                            // update_search_size
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        sz[k] = fabsf(mt[1 + k]) + 0.0001f;
        n2 = n2 + sz[k] * sz[k];
    }
    const float nrm = sqrtf(n2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float v = rescale * mean_sdf * sz[k] / nrm + 0.0001f;
        state[RO_SEARCH + k] = ok ? v : v * 2.0f;
    }
    state[RO_SUCCESS] = ok ? 1.0f : 0.0f;
    state[RO_MEAN_SDF] = mean_sdf;
    state[RO_FIT0] = f0;
#pragma unroll
    for (int k = 0; k < 7; ++k) state[RO_MEAN_T + k] = mt[k];
    state[RO_NBETTER] = (float)s_n;
}

}  // namespace mipsf

using namespace mipsf;

extern "C" {

int mipsf_ro_particles(const float* pst, const float* state, const float* rays_d_cam, const float* target_d,
                       const mipsf_render_cfg* cfg, float* xn, float* pst7, uint32_t P, uint32_t n, int point_major,
                       void* stream) {
    if (P == 0 || n == 0) return 0;
    MIPSF_REQUIRE(pst && state && rays_d_cam && target_d && cfg && xn && pst7, "null pointer");
    const uint32_t threads = P * MIPSF_WAVE;
    hipLaunchKernelGGL(ro_particles_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, pst, state,
                       rays_d_cam, target_d, make_norm(*cfg), xn, pst7, P, n, point_major ? 1 : 0);
    return check_launch("ro_particles");
}

#ifdef MIPSF_RO_LANE_CHECK
int mipsf_ro_chk_read(unsigned* count, float* dump) {
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(ro_chk_count), sizeof(unsigned)) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(dump, HIP_SYMBOL(ro_chk_dump), sizeof(float) * 32 * 64 * 24) != hipSuccess) return 1;
    return 0;
}
#endif

int mipsf_ro_update(const float* mean_masked, const float* pst7, float* state, float sdf_weight, float rescale,
                    uint32_t P, void* stream) {
    MIPSF_REQUIRE(P >= 1, "empty particle swarm");
    MIPSF_REQUIRE(mean_masked && pst7 && state, "null pointer");
    hipLaunchKernelGGL(ro_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean_masked, pst7, state,
                       sdf_weight, rescale, P);
    return check_launch("ro_update");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Keyframe ray database gather (reference model/keyframeSet.py:264-290, 386-455 and
// mipsfusion.py:296-317): rows [direction(3) | rgb(3) | depth(1)] of the device-resident database are gathered by
// host-generated flat indices (the index stream stays on the host so that python's `random.sample` draws are the
// reference's) and split straight into the three tensors the iteration consumes.
namespace mipsf {
__global__ __launch_bounds__(256) void gather_rays_kernel(const float* __restrict__ db, const int64_t* __restrict__ idx,
                                                          uint64_t n_rows, uint32_t N, float* __restrict__ rays7,
                                                          float* __restrict__ d_cam, float* __restrict__ rgb,
                                                          float* __restrict__ depth) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    int64_t r = idx[t];
    if (r < 0) r += (int64_t)n_rows;
    // the reference's torch indexing raises IndexError; a device-resident index cannot raise, so a row outside the
    // database reads nothing and yields NaN rays (every loss of the iteration turns NaN: loud, never a foreign read)
    const bool in_range = r >= 0 && (uint64_t)r < n_rows;
    const float* s = db + 7 * (size_t)(in_range ? r : 0);
    float v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = in_range ? s[k] : __builtin_nanf("");
    if (rays7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) rays7[7 * (size_t)t + k] = v[k];
    }
    if (d_cam) d_cam[3 * (size_t)t] = v[0], d_cam[3 * (size_t)t + 1] = v[1], d_cam[3 * (size_t)t + 2] = v[2];
    if (rgb) rgb[3 * (size_t)t] = v[3], rgb[3 * (size_t)t + 1] = v[4], rgb[3 * (size_t)t + 2] = v[5];
    if (depth) depth[t] = v[6];
}
}  // namespace mipsf

extern "C" int mipsf_gather_rays(const float* db, uint64_t n_rows, const int64_t* idx, uint32_t N, float* rays7,
                                 float* d_cam, float* rgb, float* depth, void* stream) {
    if (N == 0) return 0;
    MIPSF_REQUIRE(db && idx, "null pointer");
    MIPSF_REQUIRE(rays7 || d_cam || rgb || depth, "no output requested");
    hipLaunchKernelGGL(mipsf::gather_rays_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, db, idx,
                       n_rows, N, rays7, d_cam, rgb, depth);
    return mipsf::check_launch("gather_rays");
}
