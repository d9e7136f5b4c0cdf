// Small streaming kernels: frequency encoding (module API), point normalisation, dense Adam,
// RandomOptimizer fitness.  All HBM-bound: coalesced, vectorised where the layout allows.
#include "common.h"

namespace mipsf {

constexpr float PI_F = 3.14159265358979323846f;
constexpr float HALF_PI_F = 1.57079632679489661923f;   // fp32(pi)/2 == fp32(pi/2)

// ---------------------------------------------------------------- frequency (tcnn frequency.h)
// out[i, d*2F + 2k + s] = sin(fma(ldexp(x_d, k), pi, s*pi/2)); one thread per (sample, dim).
__global__ __launch_bounds__(256) void freq_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                       uint32_t n, uint32_t n_dims, uint32_t n_freq) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t i = t / n_dims, d = t - i * n_dims;
    const float v = x[t];
    float* o = out + ((size_t)i * n_dims + d) * 2 * n_freq;
    for (uint32_t k = 0; k < n_freq; ++k) {
        const float s = ldexpf(v, (int)k);
        o[2 * k] = sinf(fmaf(s, PI_F, 0.0f));
        o[2 * k + 1] = sinf(fmaf(s, PI_F, HALF_PI_F));
    }
}

__global__ __launch_bounds__(256) void freq_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                       float* __restrict__ dx, uint32_t n, uint32_t n_dims,
                                                       uint32_t n_freq) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t i = t / n_dims, d = t - i * n_dims;
    const float v = x[t];
    const float* g = dout + ((size_t)i * n_dims + d) * 2 * n_freq;
    float acc = 0.f;
    for (uint32_t k = 0; k < n_freq; ++k) {
        const float s = ldexpf(v, (int)k);
        const float fpi = ldexpf(1.0f, (int)k) * PI_F;
        acc = acc + g[2 * k] * (fpi * cosf(fmaf(s, PI_F, 0.0f)));
        acc = acc + g[2 * k + 1] * (fpi * cosf(fmaf(s, PI_F, HALF_PI_F)));
    }
    dx[t] = acc;
}

// ------------------------------------------------------------- normalisation (scene_rep.py:134-142)
__global__ __launch_bounds__(256) void normalise_kernel(const float* __restrict__ pts, float* __restrict__ xn,
                                                        uint64_t n3, NormCfg nc) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    const int d = (int)(t % 3);
    xn[t] = normalise1(pts[t], nc.sub[d], nc.div[d], nc.norm_factor);
}

__global__ __launch_bounds__(256) void normalise_bwd_kernel(const float* __restrict__ dxn, float* __restrict__ dpts,
                                                            uint64_t n3, NormCfg nc) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    const int d = (int)(t % 3);
    dpts[t] = (float)(((double)dxn[t] / nc.norm_factor) / nc.div[d]);
}

// ----------------------------------------------------------------- dense Adam (torch.optim.Adam)
// p, g, m, v streamed once: 16 B read + 12 B write per parameter (28 B with the fused zero-grad).
struct AdamK {
    float lr_over_bc1, beta1, beta2, one_minus_b1, one_minus_b2, inv_sqrt_bc2, eps, wd;
    const float* dev_hyper;   // when non-null: {lr/bc1, 1/sqrt(bc2)} are read from device memory (graph capture)
};

__device__ __forceinline__ AdamK resolve(AdamK k) {
    if (k.dev_hyper) {
        k.lr_over_bc1 = k.dev_hyper[0];
        k.inv_sqrt_bc2 = k.dev_hyper[1];
    }
    return k;
}

// advances a device-resident step counter and refreshes the two step-dependent scalars (captured into hipGraphs)
__global__ void adam_advance_kernel(int* __restrict__ step, float* __restrict__ hyper, float lr, float beta1,
                                    float beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int t = step[0] + 1;
        step[0] = t;
        const double bc1 = 1.0 - pow((double)beta1, (double)t);
        const double bc2 = 1.0 - pow((double)beta2, (double)t);
        hyper[0] = (float)((double)lr / bc1);
        hyper[1] = (float)(1.0 / sqrt(bc2));
    }
}

struct AdvanceN {
    int* step[MIPSF_ADAM_MAX_GROUPS];
    float* hyper[MIPSF_ADAM_MAX_GROUPS];
    float lr[MIPSF_ADAM_MAX_GROUPS], beta1[MIPSF_ADAM_MAX_GROUPS], beta2[MIPSF_ADAM_MAX_GROUPS];
    uint32_t n;
};
// the same for all parameter groups of an optimiser in one launch (one thread per group)
__global__ void adam_advance_n_kernel(AdvanceN a) {
    const uint32_t i = threadIdx.x;
    if (i >= a.n) return;
    const int t = a.step[i][0] + 1;
    a.step[i][0] = t;
    const double bc1 = 1.0 - pow((double)a.beta1[i], (double)t);
    const double bc2 = 1.0 - pow((double)a.beta2[i], (double)t);
    a.hyper[i][0] = (float)((double)a.lr[i] / bc1);
    a.hyper[i][1] = (float)(1.0 / sqrt(bc2));
}

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, const AdamK& k) {
    float gg = g;
    if (k.wd != 0.f) gg = gg + k.wd * p;
    m = m + (gg - m) * k.one_minus_b1;            // exp_avg.lerp_(grad, 1 - beta1)
    v = v * k.beta2 + (k.one_minus_b2 * gg) * gg;   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.lr_over_bc1 * (m / denom);
}

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, uint64_t n,
                                                   AdamK k_in) {
    const AdamK k = resolve(k_in);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, k);
        adam1(pp.y, gg.y, mm.y, vv.y, k);
        adam1(pp.z, gg.z, mm.z, vv.z, k);
        adam1(pp.w, gg.w, mm.w, vv.w, k);
        p4[i] = pp;
        m4[i] = mm;
        v4[i] = vv;
        if (ZERO) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adam1(pp, gg, mm, vv, k);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
        if (ZERO) g[i] = 0.f;
    }
}

// one launch for a group of small tensors (the decoder's 10 nn.Linear tensors): blockIdx.y = tensor
struct AdamMulti {
    float* p[MIPSF_ADAM_MAX_TENSORS];
    float* g[MIPSF_ADAM_MAX_TENSORS];
    float* m[MIPSF_ADAM_MAX_TENSORS];
    float* v[MIPSF_ADAM_MAX_TENSORS];
    uint64_t n[MIPSF_ADAM_MAX_TENSORS];
};

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMulti t, AdamK k_in) {
    const AdamK k = resolve(k_in);
    const int ti = blockIdx.y;
    float *p = t.p[ti], *g = t.g[ti], *m = t.m[ti], *v = t.v[ti];
    const uint64_t n = t.n[ti];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adam1(pp, gg, mm, vv, k);
        p[i] = pp, m[i] = mm, v[i] = vv;
        if (ZERO) g[i] = 0.f;
    }
}

// whole step of an optimiser with tiny tensors only: one workgroup = advance of every group + all element updates
template <bool ZERO>
__global__ __launch_bounds__(256) void adam_small_kernel(mipsf_adam_small d) {
    __shared__ float hyp[MIPSF_ADAM_MAX_GROUPS][2];
    if (threadIdx.x < d.n_groups) {
        const uint32_t i = threadIdx.x;
        const int t = d.step_dev[i][0] + 1;
        d.step_dev[i][0] = t;
        const double bc1 = 1.0 - pow((double)d.beta1[i], (double)t);
        const double bc2 = 1.0 - pow((double)d.beta2[i], (double)t);
        const float h0 = (float)((double)d.lr[i] / bc1), h1 = (float)(1.0 / sqrt(bc2));
        hyp[i][0] = h0, hyp[i][1] = h1;
        d.hyper_dev[i][0] = h0, d.hyper_dev[i][1] = h1;
    }
    __syncthreads();
    for (uint32_t ti = 0; ti < d.n_tensors; ++ti) {
        const uint32_t gi = d.group_of[ti];
        AdamK k;
        k.lr_over_bc1 = hyp[gi][0], k.inv_sqrt_bc2 = hyp[gi][1];
        k.beta1 = d.beta1[gi], k.beta2 = d.beta2[gi];
        k.one_minus_b1 = (float)(1.0 - (double)d.beta1[gi]), k.one_minus_b2 = (float)(1.0 - (double)d.beta2[gi]);
        k.eps = d.eps[gi], k.wd = d.weight_decay[gi], k.dev_hyper = nullptr;
        float *p = d.param[ti], *g = d.grad[ti], *m = d.exp_avg[ti], *v = d.exp_avg_sq[ti];
        for (uint32_t i = threadIdx.x; i < d.numel[ti]; i += blockDim.x) {
            float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
            adam1(pp, gg, mm, vv, k);
            p[i] = pp, m[i] = mm, v[i] = vv;
            if (ZERO) g[i] = 0.f;
        }
    }
}

// The whole step of ANY capturable optimiser in one launch (map optimiser: the 36 MB table + the decoder's ten tensors; it
// used to be adam_advance_n + adam_kernel + adam_multi_kernel = three launches, two of them 5 us of latency each).
// Workgroup b serves tensor `tensor_of(b)`; every workgroup derives its group's step-dependent scalars from the device step
// counter itself (old value + 1, the arithmetic of adam_advance_n_kernel), nobody writes the counter while others may still
// read it: the LAST workgroup to take a ticket (each takes one after its read) advances the counters, refreshes hyper_dev
// and puts the ticket back to zero.  ticket: MIPSF_ADAM_TICKET_WORDS words that belong to ONE optimiser (they also carry
// the next step's scalars of its groups).
// beta^t for an integer step count by repeated squaring in double precision (within a few double-precision ulps of libm's
// pow -- 1e-16 relative, far below the fp32 rounding of the scalars derived from it -- at a tenth of its registers: with
// pow in the kernel its 105 SGPRs admit 6 workgroups per CU instead of 8, and the bandwidth-bound update ran 48 us
// instead of 42)
__device__ __forceinline__ double powi(double b, int t) {
    double r = 1.0;
    while (t > 0) {
        if (t & 1) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

// what the scalars left for the next step were computed from (a learning-rate schedule changes lr between two steps)
__device__ __forceinline__ uint32_t adam_key(float lr, float b1, float b2) {
    return __float_as_uint(lr) ^ (__float_as_uint(b1) * 3u) ^ (__float_as_uint(b2) * 7u);
}

struct AdamAllMap {
    uint32_t blk0[MIPSF_ADAM_MAX_TENSORS + 1];      // first workgroup of every tensor
};

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_all_kernel(mipsf_adam_small d, AdamAllMap map, uint32_t* __restrict__ ticket) {
    __shared__ float hyp[2];
    uint32_t ti = 0;
    while (ti + 1 < d.n_tensors && blockIdx.x >= map.blk0[ti + 1]) ++ti;
    const uint32_t gi = d.group_of[ti];
    float *p = d.param[ti], *g = d.grad[ti], *m = d.exp_avg[ti], *v = d.exp_avg_sq[ti];
    const uint64_t n = d.numel[ti];
    const uint32_t nb = map.blk0[ti + 1] - map.blk0[ti], b = blockIdx.x - map.blk0[ti];
    const uint64_t stride = (uint64_t)nb * 256;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const uint64_t n4 = vec ? n / 4 : 0;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    // the first elements are requested BEFORE the step-dependent scalars are worked out (two double-precision pow by one
    // thread and a barrier: ~3 us in front of every workgroup's first load otherwise)
    uint64_t i = (uint64_t)b * 256 + threadIdx.x;
    float4 pp, gg, mm, vv;
    if (i < n4) pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    // The step-dependent scalars of THIS step were left by the last workgroup of the previous step (tagged with the step
    // they belong to, behind the tickets); only a first call -- or one after a reset / a loaded state -- works them out here
    // (two double-precision pow by one thread: ~1 us in front of the workgroup's first store).
    uint32_t* nexth = ticket + 16u * 33u + 4u * gi;            // {step tag, lr / bc1, 1 / sqrt(bc2), -}
    if (threadIdx.x == 0) {
        const int t = d.step_dev[gi][0] + 1;
        const uint32_t tag = nexth[0];
        const float h0 = __uint_as_float(nexth[1]), h1 = __uint_as_float(nexth[2]);
        if (tag == (uint32_t)t && nexth[3] == adam_key(d.lr[gi], d.beta1[gi], d.beta2[gi])) {
            hyp[0] = h0, hyp[1] = h1;
        } else {
            const double bc1 = 1.0 - powi((double)d.beta1[gi], t);
            const double bc2 = 1.0 - powi((double)d.beta2[gi], t);
            hyp[0] = (float)((double)d.lr[gi] / bc1), hyp[1] = (float)(1.0 / sqrt(bc2));
        }
    }
    __syncthreads();
    AdamK k;
    k.lr_over_bc1 = hyp[0], k.inv_sqrt_bc2 = hyp[1];
    k.beta1 = d.beta1[gi], k.beta2 = d.beta2[gi];
    k.one_minus_b1 = (float)(1.0 - (double)d.beta1[gi]), k.one_minus_b2 = (float)(1.0 - (double)d.beta2[gi]);
    k.eps = d.eps[gi], k.wd = d.weight_decay[gi], k.dev_hyper = nullptr;
    // Two-level ticket, taken by ONE thread right after this workgroup has read the counter and the scalars (the other
    // three waves go on; nobody waits at a barrier for the atomic's round trip, and nothing is left to do when the
    // kernel ends): same-address device atomics retire at ~90 per microsecond and the grid has ~2000 workgroups (one
    // ticket word: 23 us of a 42 us kernel).  32 sub-tickets on their own cache lines take ~64 increments each; the last
    // workgroup of a sub-ticket takes the master ticket, the last of those knows that EVERY workgroup has read: it
    // advances the counters and leaves the next step's scalars while the others are still streaming.
    if (threadIdx.x == 0) {
        const uint32_t sub = blockIdx.x & 31u;
        const uint32_t n_sub = (gridDim.x - sub + 31u) / 32u;                 // workgroups with this sub-ticket
        bool last = false;
        if (__hip_atomic_fetch_add(ticket + 16u * (1u + sub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_sub - 1) {
            __hip_atomic_store(ticket + 16u * (1u + sub), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t n_master = gridDim.x < 32u ? gridDim.x : 32u;
            last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_master - 1;
        }
        if (last) {
            for (uint32_t q = 0; q < d.n_groups; ++q) {
                const int t = d.step_dev[q][0] + 1;
                d.step_dev[q][0] = t;
                const double b1 = (double)d.beta1[q], b2 = (double)d.beta2[q];
                d.hyper_dev[q][0] = (float)((double)d.lr[q] / (1.0 - powi(b1, t)));
                d.hyper_dev[q][1] = (float)(1.0 / sqrt(1.0 - powi(b2, t)));
                uint32_t* nx = ticket + 16u * 33u + 4u * q;          // the scalars of step t + 1
                nx[1] = __float_as_uint((float)((double)d.lr[q] / (1.0 - powi(b1, t + 1))));
                nx[2] = __float_as_uint((float)(1.0 / sqrt(1.0 - powi(b2, t + 1))));
                nx[3] = adam_key(d.lr[q], d.beta1[q], d.beta2[q]);
                nx[0] = (uint32_t)(t + 1);
            }
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (i < n4) {           // the element requested before the barrier
        adam1(pp.x, gg.x, mm.x, vv.x, k);
        adam1(pp.y, gg.y, mm.y, vv.y, k);
        adam1(pp.z, gg.z, mm.z, vv.z, k);
        adam1(pp.w, gg.w, mm.w, vv.w, k);
        p4[i] = pp, m4[i] = mm, v4[i] = vv;
        if (ZERO) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (fresh registers per round: a loop that carries the loaded values from one round into the next makes the compiler
    // drain every store before the next load may land in the registers the store reads -- s_waitcnt vmcnt(0) per round,
    // one load in flight at a time: 48 instead of 42 us)
    for (uint64_t r = i + stride; r < n4; r += stride) {
        float4 p1 = p4[r], g1 = g4[r], m1 = m4[r], v1 = v4[r];
        // all four loads in flight before any arithmetic: the compiler otherwise issues the fourth tensor's load after the
        // first three have ARRIVED and been used (two memory round trips per round instead of one; the plain table kernel
        // does not do that: 42 vs 48 us).  The empty asm "uses" the four values at one point.
        asm volatile("" : "+v"(p1.x), "+v"(g1.x), "+v"(m1.x), "+v"(v1.x));
        adam1(p1.x, g1.x, m1.x, v1.x, k);
        adam1(p1.y, g1.y, m1.y, v1.y, k);
        adam1(p1.z, g1.z, m1.z, v1.z, k);
        adam1(p1.w, g1.w, m1.w, v1.w, k);
        p4[r] = p1, m4[r] = m1, v4[r] = v1;
        if (ZERO) g4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (uint64_t q = n4 * 4 + (uint64_t)b * 256 + threadIdx.x; q < n; q += stride) {
        float p1 = p[q], g1 = g[q], m1 = m[q], v1 = v[q];
        adam1(p1, g1, m1, v1, k);
        p[q] = p1, m[q] = m1, v[q] = v1;
        if (ZERO) g[q] = 0.f;
    }
}

// ------------------------------------------------------ RandomOptimizer.get_fitness (RandomOptimizer.py:125-129)
// one wave per particle: mean_j( (d_j > 0) * |sdf_pj * trunc| )
// row of (particle, point) = particle * row_p + point * row_j, `stride` floats per row, SDF in column 3
__global__ __launch_bounds__(256) void ro_fitness_kernel(const float* __restrict__ raw, uint32_t stride,
                                                         const float* __restrict__ target_d, float trunc,
                                                         float* __restrict__ out, uint32_t P, uint32_t n,
                                                         uint32_t row_p, uint32_t row_j) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / MIPSF_WAVE;
    const uint32_t lane = threadIdx.x & (MIPSF_WAVE - 1);
    if (wave >= P) return;
    float acc = 0.f;
    for (uint32_t j = lane; j < n; j += MIPSF_WAVE) {
        const float s = raw[((size_t)wave * row_p + (size_t)j * row_j) * stride + 3] * trunc;
        acc += (target_d[j] > 0.f) ? fabsf(s) : 0.f;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[wave] = acc / (float)n;
}

}  // namespace mipsf

using namespace mipsf;

extern "C" {

int mipsf_freq_fwd(const float* x, float* out, uint32_t M, uint32_t n_dims, uint32_t n_freq, void* stream) {
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && out, "null pointer");
    const uint32_t n = M * n_dims;
    hipLaunchKernelGGL(freq_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, n, n_dims,
                       n_freq);
    return check_launch("freq_fwd");
}

int mipsf_freq_bwd(const float* x, const float* dout, float* dx, uint32_t M, uint32_t n_dims, uint32_t n_freq,
                   void* stream) {
    if (M == 0) return 0;
    MIPSF_REQUIRE(x && dout && dx, "null pointer");
    const uint32_t n = M * n_dims;
    hipLaunchKernelGGL(freq_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, dout, dx, n,
                       n_dims, n_freq);
    return check_launch("freq_bwd");
}

int mipsf_normalise_points(const float* pts, const mipsf_render_cfg* cfg, float* xn, uint32_t M, void* stream) {
    if (M == 0) return 0;
    MIPSF_REQUIRE(pts && cfg && xn, "null pointer");
    const uint64_t n3 = (uint64_t)M * 3;
    hipLaunchKernelGGL(normalise_kernel, dim3((uint32_t)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts,
                       xn, n3, make_norm(*cfg));
    return check_launch("normalise_points");
}

int mipsf_normalise_bwd(const float* dxn, const mipsf_render_cfg* cfg, float* dpts, uint32_t M, void* stream) {
    if (M == 0) return 0;
    MIPSF_REQUIRE(dxn && cfg && dpts, "null pointer");
    const uint64_t n3 = (uint64_t)M * 3;
    hipLaunchKernelGGL(normalise_bwd_kernel, dim3((uint32_t)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dxn, dpts, n3, make_norm(*cfg));
    return check_launch("normalise_bwd");
}

static AdamK make_adam(float lr, float beta1, float beta2, float eps, float weight_decay, uint32_t step,
                       const float* dev_hyper) {
    AdamK k;
    const double bc1 = 1.0 - pow((double)beta1, (double)(step ? step : 1));
    const double bc2 = 1.0 - pow((double)beta2, (double)(step ? step : 1));
    k.lr_over_bc1 = (float)((double)lr / bc1);
    k.beta1 = beta1;
    k.beta2 = beta2;
    k.one_minus_b1 = (float)(1.0 - (double)beta1);
    k.one_minus_b2 = (float)(1.0 - (double)beta2);
    k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    k.eps = eps;
    k.wd = weight_decay;
    k.dev_hyper = dev_hyper;
    return k;
}

int mipsf_adam_advance_n(int32_t* const* step_dev, float* const* hyper_dev, const float* lr, const float* beta1,
                         const float* beta2, uint32_t n_groups, void* stream) {
    if (n_groups == 0) return 0;
    MIPSF_REQUIRE(step_dev && hyper_dev && lr && beta1 && beta2, "null pointer");
    MIPSF_REQUIRE(n_groups <= MIPSF_ADAM_MAX_GROUPS, "at most %d groups per call", MIPSF_ADAM_MAX_GROUPS);
    AdvanceN a;
    a.n = n_groups;
    for (uint32_t i = 0; i < n_groups; ++i) {
        MIPSF_REQUIRE(step_dev[i] && hyper_dev[i], "null pointer in group %u", i);
        a.step[i] = step_dev[i], a.hyper[i] = hyper_dev[i], a.lr[i] = lr[i], a.beta1[i] = beta1[i], a.beta2[i] = beta2[i];
    }
    hipLaunchKernelGGL(adam_advance_n_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch("adam_advance_n");
}

int mipsf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, uint32_t step, const float* hyper_dev,
                    int zero_grad, void* stream) {
    if (n == 0) return 0;
    MIPSF_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
    MIPSF_REQUIRE(step >= 1 || hyper_dev, "step must be >= 1 (or hyper_dev given)");
    MIPSF_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                  "adam buffers must be 16-byte aligned");
    const AdamK k = make_adam(lr, beta1, beta2, eps, weight_decay, step, hyper_dev);
    uint64_t want = (n / 4 + 255) / 256;
    uint32_t blocks = (uint32_t)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    if (zero_grad)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                           exp_avg_sq, n, k);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                           exp_avg_sq, n, k);
    return check_launch("adam_step");
}

int mipsf_adam_step_multi(const mipsf_adam_tensors* t, float lr, float beta1, float beta2, float eps,
                          float weight_decay, uint32_t step, const float* hyper_dev, int zero_grad, void* stream) {
    MIPSF_REQUIRE(t != nullptr, "null tensor table");
    if (t->count == 0) return 0;
    MIPSF_REQUIRE(t->count <= MIPSF_ADAM_MAX_TENSORS, "too many tensors (%u)", t->count);
    MIPSF_REQUIRE(step >= 1 || hyper_dev, "step must be >= 1 (or hyper_dev given)");
    AdamMulti a;
    uint64_t nmax = 0;
    for (uint32_t i = 0; i < MIPSF_ADAM_MAX_TENSORS; ++i) {
        const bool live = i < t->count;
        a.p[i] = live ? t->param[i] : nullptr, a.g[i] = live ? t->grad[i] : nullptr;
        a.m[i] = live ? t->exp_avg[i] : nullptr, a.v[i] = live ? t->exp_avg_sq[i] : nullptr;
        a.n[i] = live ? t->numel[i] : 0;
        if (live) {
            MIPSF_REQUIRE(a.p[i] && a.g[i] && a.m[i] && a.v[i], "null pointer in tensor %u", i);
            nmax = a.n[i] > nmax ? a.n[i] : nmax;
        }
    }
    const AdamK k = make_adam(lr, beta1, beta2, eps, weight_decay, step, hyper_dev);
    uint64_t bx = (nmax + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    const dim3 grid((uint32_t)bx, t->count);
    if (zero_grad) hipLaunchKernelGGL(adam_multi_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, k);
    else hipLaunchKernelGGL(adam_multi_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, k);
    return check_launch("adam_step_multi");
}

int mipsf_adam_step_small(const mipsf_adam_small* d, int zero_grad, void* stream) {
    MIPSF_REQUIRE(d, "null descriptor");
    MIPSF_REQUIRE(d->n_groups >= 1 && d->n_groups <= MIPSF_ADAM_MAX_GROUPS, "bad group count %u", d->n_groups);
    MIPSF_REQUIRE(d->n_tensors >= 1 && d->n_tensors <= MIPSF_ADAM_MAX_TENSORS, "bad tensor count %u", d->n_tensors);
    for (uint32_t i = 0; i < d->n_groups; ++i) MIPSF_REQUIRE(d->step_dev[i] && d->hyper_dev[i], "null pointer in group %u", i);
    for (uint32_t i = 0; i < d->n_tensors; ++i) {
        MIPSF_REQUIRE(d->param[i] && d->grad[i] && d->exp_avg[i] && d->exp_avg_sq[i], "null pointer in tensor %u", i);
        MIPSF_REQUIRE(d->numel[i] <= MIPSF_ADAM_SMALL_MAX_NUMEL && d->group_of[i] < d->n_groups, "tensor %u out of range", i);
    }
    if (zero_grad) hipLaunchKernelGGL(adam_small_kernel<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(adam_small_kernel<false>, dim3(1), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch("adam_step_small");
}

int mipsf_adam_step_all(const mipsf_adam_small* d, int zero_grad, uint32_t* ticket, void* stream) {
    MIPSF_REQUIRE(d && ticket, "null pointer");
    MIPSF_REQUIRE(d->n_groups >= 1 && d->n_groups <= MIPSF_ADAM_MAX_GROUPS, "bad group count %u", d->n_groups);
    MIPSF_REQUIRE(d->n_tensors >= 1 && d->n_tensors <= MIPSF_ADAM_MAX_TENSORS, "bad tensor count %u", d->n_tensors);
    for (uint32_t i = 0; i < d->n_groups; ++i) MIPSF_REQUIRE(d->step_dev[i] && d->hyper_dev[i], "null pointer in group %u", i);
    AdamAllMap map;
    // All workgroups must be resident at once (8 of 256 threads per CU): a workgroup that has to wait for a slot starts
    // when the first grid-stride loop ends, i.e. a whole kernel duration late (measured: the decoder's 40 small workgroups
    // behind 2048 table workgroups made the launch 48 us instead of 42).  The budget is shared in proportion to the sizes.
    const int cus = device_cus();
    if (cus <= 0) return 3;
    const uint64_t budget = (uint64_t)cus * 8;
    uint64_t want_all = 0;
    for (uint32_t i = 0; i < d->n_tensors; ++i) want_all += ((uint64_t)d->numel[i] / 4 + 255) / 256 + 1;
    uint32_t blocks = 0;
    for (uint32_t i = 0; i < d->n_tensors; ++i) {
        MIPSF_REQUIRE(d->param[i] && d->grad[i] && d->exp_avg[i] && d->exp_avg_sq[i], "null pointer in tensor %u", i);
        MIPSF_REQUIRE(d->group_of[i] < d->n_groups, "tensor %u: group out of range", i);
        map.blk0[i] = blocks;
        uint64_t want = ((uint64_t)d->numel[i] / 4 + 255) / 256 + 1;
        if (want_all > budget) want = want * budget / want_all;        // (rounded down: the sum stays within the budget)
        blocks += (uint32_t)(want < 1 ? 1 : want);
    }
    for (uint32_t i = d->n_tensors; i <= MIPSF_ADAM_MAX_TENSORS; ++i) map.blk0[i] = blocks;
    if (zero_grad) hipLaunchKernelGGL(adam_all_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d, map, ticket);
    else hipLaunchKernelGGL(adam_all_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d, map, ticket);
    return check_launch("adam_step_all");
}

int mipsf_ro_fitness(const float* raw, uint32_t raw_stride, const float* target_d, float trunc, float* mean_masked,
                     uint32_t P, uint32_t n, void* stream) {
    if (P == 0) return 0;
    MIPSF_REQUIRE(raw && target_d && mean_masked, "null pointer");
    MIPSF_REQUIRE(raw_stride >= 4, "raw_stride must cover the sdf column");
    const uint32_t threads = P * MIPSF_WAVE;
    hipLaunchKernelGGL(ro_fitness_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw,
                       raw_stride, target_d, trunc, mean_masked, P, n, n, 1u);
    return check_launch("ro_fitness");
}

int mipsf_ro_fitness_sdf(const float* sdf, const float* target_d, float trunc, float* mean_masked, uint32_t P,
                         uint32_t n, int point_major, void* stream) {
    if (P == 0) return 0;
    MIPSF_REQUIRE(sdf && target_d && mean_masked, "null pointer");
    const uint32_t threads = P * MIPSF_WAVE;
    // the kernel reads element 3 of rows `stride` floats apart: a dense [P*n] SDF array is that table shifted by 3
    hipLaunchKernelGGL(ro_fitness_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, sdf - 3, 1u,
                       target_d, trunc, mean_masked, P, n, point_major ? 1u : n, point_major ? P : 1u);
    return check_launch("ro_fitness");
}

}  // extern "C"
