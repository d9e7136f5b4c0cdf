/* Host-side replica of torch's DEFAULT CPU GENERATOR draws used by the reference's samplers -- the serial stage that bounds
 * the frame time when the index stream has to be the reference's own (mipsfusion_amd/sequence.py):
 *   torch.rand(n, dtype=float32)             at::uniform_real_distribution<float>(0, 1) over at::mt19937
 *   torch.randn(n >= 16, dtype=float32)      ATen/native/cpu/DistributionTemplates.h normal_fill + normal_fill_16_AVX2
 *                                            (Box-Muller on 8 + 8 uniforms per block of 16, cephes log / sincos of
 *                                            ATen/native/cpu/avx_mathfun.h), tail block recomputed as torch does
 * Same generator stream, same values to the last bit: the Mersenne twister runs here on the state taken from (and given
 * back to) torch.get_rng_state(), and the transcendental kernels repeat the vector code's operations one by one --
 * including WHICH multiply-adds the compiler of the torch wheels fused, found by search against torch.randn
 * (tools/micro/randn_match.py): fused = fmaf() below, unfused = separate operations (this file is compiled with
 * -ffp-contract=off).  mipsfusion_amd/hostrng.py verifies the replica against torch at start-up and falls back to torch's
 * own functions if a build of torch should ever differ.  The serial part is the twister alone (~1 ns per value); the
 * Box-Muller transform runs on `threads` OpenMP threads.                                                             */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MT_N 624
#define MT_M 397

#include "mipsf_host.h"          /* mipsf_mt and the entry points (include/mipsf_host.h) */

static inline uint32_t mt_twist(uint32_t u, uint32_t v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

static void mt_next_state(mipsf_mt* g) {        /* at::mt19937::next_state */
    uint32_t* p = g->state;
    g->left = MT_N;
    g->next = 0;
    for (int j = MT_N - MT_M + 1; --j; p++) *p = p[MT_M] ^ mt_twist(p[0], p[1]);
    for (int j = MT_M; --j; p++) *p = p[MT_M - MT_N] ^ mt_twist(p[0], p[1]);
    *p = p[MT_M - MT_N] ^ mt_twist(p[0], g->state[0]);
}

/* at::mt19937::operator() is `if (--left == 0) next_state(); y = temper(state[next++])`, i.e. with `left` = L there are
 * L - 1 words to take before the state is regenerated (and regeneration leaves left = 624 AFTER its first word is taken).
 * at::uniform_real_distribution<float>(0, 1): (random() & (2^24 - 1)) * 2^-24.  In bulk: temper + convert a run of words
 * in one vectorisable loop. */
void mipsf_mt_uniform_f32(mipsf_mt* g, float* out, int64_t n) {
    while (n > 0) {
        if (g->left == 1) {
            mt_next_state(g);
            g->left = MT_N + 1;
        }
        int64_t k = g->left - 1;
        if (k > n) k = n;
        const uint32_t* src = g->state + g->next;
#pragma omp simd
        for (int64_t i = 0; i < k; ++i) {
            uint32_t y = src[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[i] = (float)(int32_t)(y & 0xffffffu) * 5.9604644775390625e-8f;
        }
        g->left -= (int32_t)k;
        g->next += (uint32_t)k;
        out += k;
        n -= k;
    }
}

/* ---- 8 lanes at a time with explicit AVX2 / FMA instructions: `fma` below = fused in torch's build, mul / add = not ---- */
#include <immintrin.h>
typedef __m256 v8f;
typedef __m256i v8i;
static inline __attribute__((always_inline)) v8f set1(float x) { return _mm256_set1_ps(x); }
static inline __attribute__((always_inline)) v8f fma8(v8f a, v8f b, v8f c) { return _mm256_fmadd_ps(a, b, c); }

/* avx_mathfun.h log256_ps */
static inline __attribute__((always_inline)) v8f log8(v8f x) {
    const v8f one = set1(1.0f);
    x = _mm256_max_ps(x, _mm256_castsi256_ps(_mm256_set1_epi32(0x00800000)));
    v8i imm0 = _mm256_srli_epi32(_mm256_castps_si256(x), 23);
    x = _mm256_and_ps(x, _mm256_castsi256_ps(_mm256_set1_epi32(~0x7f800000)));
    x = _mm256_or_ps(x, set1(0.5f));
    imm0 = _mm256_sub_epi32(imm0, _mm256_set1_epi32(0x7f));
    v8f e = _mm256_add_ps(_mm256_cvtepi32_ps(imm0), one);
    const v8f mask = _mm256_cmp_ps(x, set1(0.707106781186547524f), _CMP_LT_OS);
    const v8f tmp0 = _mm256_and_ps(x, mask);
    x = _mm256_sub_ps(x, one);
    e = _mm256_sub_ps(e, _mm256_and_ps(one, mask));
    x = _mm256_add_ps(x, tmp0);
    const v8f z = _mm256_mul_ps(x, x);
    v8f y = set1(7.0376836292E-2f);
    y = fma8(y, x, set1(-1.1514610310E-1f));
    y = fma8(y, x, set1(1.1676998740E-1f));
    y = fma8(y, x, set1(-1.2420140846E-1f));
    y = fma8(y, x, set1(1.4249322787E-1f));
    y = fma8(y, x, set1(-1.6668057665E-1f));
    y = fma8(y, x, set1(2.0000714765E-1f));
    y = fma8(y, x, set1(-2.4999993993E-1f));
    y = fma8(y, x, set1(3.3333331174E-1f));
    y = _mm256_mul_ps(y, x);
    y = fma8(y, z, _mm256_mul_ps(e, set1(-2.12194440e-4f)));     /* (y x) z + e q1: e q1 rounded, the rest one fma */
    y = _mm256_sub_ps(y, _mm256_mul_ps(z, set1(0.5f)));
    x = _mm256_add_ps(x, y);
    x = _mm256_add_ps(x, _mm256_mul_ps(e, set1(0.693359375f)));
    return x;
}

/* avx_mathfun.h sincos256_ps (AVX2 integer path) */
static inline __attribute__((always_inline)) void sincos8(v8f x, v8f* s, v8f* c) {
    const v8f sign_mask = _mm256_castsi256_ps(_mm256_set1_epi32((int)0x80000000));
    v8f sign_bit_sin = _mm256_and_ps(x, sign_mask);
    x = _mm256_andnot_ps(sign_mask, x);
    v8f y = _mm256_mul_ps(x, set1(1.27323954473516f));
    v8i imm2 = _mm256_cvttps_epi32(y);
    imm2 = _mm256_add_epi32(imm2, _mm256_set1_epi32(1));
    imm2 = _mm256_and_si256(imm2, _mm256_set1_epi32(~1));
    y = _mm256_cvtepi32_ps(imm2);
    v8i imm4 = imm2;
    v8i imm0 = _mm256_slli_epi32(_mm256_and_si256(imm2, _mm256_set1_epi32(4)), 29);
    imm2 = _mm256_cmpeq_epi32(_mm256_and_si256(imm2, _mm256_set1_epi32(2)), _mm256_setzero_si256());
    const v8f swap_sign_bit_sin = _mm256_castsi256_ps(imm0);
    const v8f poly_mask = _mm256_castsi256_ps(imm2);
    x = _mm256_add_ps(x, _mm256_mul_ps(y, set1(-0.78515625f)));
    x = _mm256_add_ps(x, _mm256_mul_ps(y, set1(-2.4187564849853515625e-4f)));
    x = _mm256_add_ps(x, _mm256_mul_ps(y, set1(-3.77489497744594108e-8f)));
    imm4 = _mm256_sub_epi32(imm4, _mm256_set1_epi32(2));
    imm4 = _mm256_slli_epi32(_mm256_andnot_si256(imm4, _mm256_set1_epi32(4)), 29);
    const v8f sign_bit_cos = _mm256_castsi256_ps(imm4);
    sign_bit_sin = _mm256_xor_ps(sign_bit_sin, swap_sign_bit_sin);
    const v8f z = _mm256_mul_ps(x, x);
    v8f yc = set1(2.443315711809948E-005f);
    yc = fma8(yc, z, set1(-1.388731625493765E-003f));
    yc = fma8(yc, z, set1(4.166664568298827E-002f));
    yc = _mm256_mul_ps(yc, z);
    yc = _mm256_fmsub_ps(yc, z, _mm256_mul_ps(z, set1(0.5f)));   /* (y z) z - z/2 as one fmsub */
    yc = _mm256_add_ps(yc, set1(1.0f));
    v8f ys = set1(-1.9515295891E-4f);
    ys = fma8(ys, z, set1(8.3321608736E-3f));
    ys = fma8(ys, z, set1(-1.6666654611E-1f));
    ys = _mm256_mul_ps(ys, z);
    ys = fma8(ys, x, x);
    const v8f ysin2 = _mm256_and_ps(poly_mask, ys);
    const v8f ysin1 = _mm256_andnot_ps(poly_mask, yc);
    ys = _mm256_sub_ps(ys, ysin2);
    yc = _mm256_sub_ps(yc, ysin1);
    const v8f xs = _mm256_add_ps(ysin1, ysin2), xc = _mm256_add_ps(yc, ys);
    *s = _mm256_xor_ps(xs, sign_bit_sin);
    *c = _mm256_xor_ps(xc, sign_bit_cos);
}

/* normal_fill_16_AVX2 with mean 0, std 1: data[0..7], data[8..15] uniforms -> normals */
static inline __attribute__((always_inline)) void normal_block16(float* data) {
    const v8f u1 = _mm256_sub_ps(set1(1.0f), _mm256_loadu_ps(data));
    const v8f u2 = _mm256_loadu_ps(data + 8);
    const v8f radius = _mm256_sqrt_ps(_mm256_mul_ps(set1(-2.0f), log8(u1)));
    const v8f theta = _mm256_mul_ps(set1(6.28318548202514648f), u2);          /* float(2 pi) */
    v8f sn, cs;
    sincos8(theta, &sn, &cs);
    _mm256_storeu_ps(data, _mm256_mul_ps(radius, cs));
    _mm256_storeu_ps(data + 8, _mm256_mul_ps(radius, sn));
}

/* torch.randn semantics for a contiguous float32 tensor of n >= 16 elements (normal_fill) */
int mipsf_mt_normal_f32(mipsf_mt* g, float* out, int64_t n, int threads) {
    if (n < 16) return 1;
    mipsf_mt_uniform_f32(g, out, n);
    (void)threads;                         /* vectorised, single thread: ~2 ns per value, the twister is the rest */
    const int64_t blocks = n / 16;
    for (int64_t b = 0; b < blocks; ++b) normal_block16(out + 16 * b);
    if (n % 16 != 0) {                     /* torch recomputes the last 16 values from 16 fresh uniforms */
        float* tail = out + n - 16;
        mipsf_mt_uniform_f32(g, tail, 16);
        normal_block16(tail);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * python's `random.sample(range(n), k)` (CPython 3.10 Lib/random.py: Random.sample, _randbelow_with_getrandbits;
 * Modules/_randommodule.c: getrandbits(b <= 32) = genrand_uint32() >> (32 - b)) -- the keyframe-ray draws of every
 * mapping iteration (keyframeSet.py:386-436).  Here g->next is python's `index` into the state (state[624] of
 * random.getstate()); `left` is unused.  scratch: n int64 (pool branch) or n bytes (set branch) -- n * 8 bytes cover both.
 * Returns 0, -1 for arguments python would refuse or that exceed this replica (n >= 2^32).                           */
static inline uint32_t py_genrand(mipsf_mt* g) {
    if (g->next >= MT_N) mt_next_state(g);
    uint32_t y = g->state[g->next++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static inline int64_t py_randbelow(mipsf_mt* g, int64_t n) {       /* n >= 1 */
    int bits = 64 - __builtin_clzll((unsigned long long)n);          /* n.bit_length() */
    uint32_t r = py_genrand(g) >> (32 - bits);
    while ((int64_t)r >= n) r = py_genrand(g) >> (32 - bits);
    return (int64_t)r;
}

int mipsf_py_sample_range(mipsf_mt* g, int64_t n, int64_t k, int64_t* out, void* scratch) {
    if (k < 0 || k > n || n >= ((int64_t)1 << 32)) return -1;
    double setsize = 21.0;
    if (k > 5) setsize += pow(4.0, ceil(log((double)(k * 3)) / log(4.0)));      /* 4 ** _ceil(_log(k * 3, 4)) */
    if ((double)n <= setsize) {
        int64_t* pool = (int64_t*)scratch;
        for (int64_t i = 0; i < n; ++i) pool[i] = i;
        for (int64_t i = 0; i < k; ++i) {
            const int64_t j = py_randbelow(g, n - i);
            out[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
    } else {
        uint8_t* selected = (uint8_t*)scratch;
        memset(selected, 0, (size_t)n);
        for (int64_t i = 0; i < k; ++i) {
            int64_t j = py_randbelow(g, n);
            while (selected[j]) j = py_randbelow(g, n);
            selected[j] = 1;
            out[i] = j;
        }
    }
    return 0;
}

int mipsf_hostrng_abi(void) { return 2; }
