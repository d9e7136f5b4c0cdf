// Hardware facts the round-6 weight-gradient kernel is built on, checked on the device (tools/micro/tr_probe):
//   1. ds_read_b64_tr_b16: which 16-bit element lands in which lane / register (hypothesis of the programming guide: inside a
//      16-lane group, result lane i, element k = element (i & 3) of the 8-byte chunk addressed by source lane 4 k + (i >> 2)).
//   2. v_mfma_f32_16x16x32_bf16 operand layout: A lane l = row l & 15, k = 8 (l >> 4) + u;  B lane l = column l & 15, same k;
//      D lane l = column l & 15, rows 4 (l >> 4) + r.
//   3. a 32 x 32 block in the decoder records' load layout -> LDS -> transpose reads = the A / B operand of
//      v_mfma_f32_32x32x16_bf16 (checked through the product with a second, directly built operand).
//   4. cycles of the LDS round trip of 3 beside the matrix-core transpose it replaces.
// build: hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define LDS_S4(p) ((__attribute__((address_space(3))) s4*)(p))

__global__ void k_map(short* out) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    // every lane addresses its own chunk, chunks in a scrambled order so that the mapping cannot be read off contiguity
    const int chunk = (threadIdx.x * 37 + 11) & 255;
    const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S4(lds + chunk * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

__global__ void k_mfma16(const float* A, const float* B, float* D) {      // A [16][32], B [32][16] -> D [16][16]
    const int l = threadIdx.x, i = l & 15, kg = l >> 4;
    bf8 a, b;
    for (int u = 0; u < 8; ++u) a[u] = (__bf16)A[i * 32 + 8 * kg + u], b[u] = (__bf16)B[(8 * kg + u) * 16 + i];
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * kg + r) * 16 + i] = c[r];
}

// the records' load layout: lane (j = l & 31 = sample, h = l >> 5) holds the 16 features 16 q + 8 a + 4 h + e (q, a in {0, 1},
// e = 0..3) of a 32 x 32 block V[sample][feature] as four 4-element chunks; chunk id fg = 4 q + 2 a + h
__device__ __forceinline__ int chunk_addr(int s, int fg) { return (s * 8 + (fg ^ ((s >> 1) & 7))) * 4; }      // in 16-bit elements
__global__ void k_block(const float* V, const float* W, float* D, unsigned long long* cyc) {
    // D[f][n] = sum_s V[s][f] W[s][n]: A operand = V transposed through LDS, B operand built directly
    __shared__ __attribute__((aligned(16))) short lds[32 * 32];
    const int l = threadIdx.x, j = l & 31, h = l >> 5;
    unsigned long long t0 = 0, t1 = 0;
    bf8 A[2];
    for (int rep = 0; rep < 2; ++rep) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int q = 0; q < 2; ++q)
            for (int a = 0; a < 2; ++a) {
                const int fg = 4 * q + 2 * a + h;
                bf4 w;
                for (int e = 0; e < 4; ++e) w[e] = (__bf16)V[j * 32 + 4 * fg + e];
                *reinterpret_cast<bf4*>(lds + chunk_addr(j, fg)) = w;
            }
        // A operand of k-step m: lane (i = l & 31 = feature, kg = l >> 5) holds samples 16 m + 8 kg + u.  Two reads of four
        // samples; the 16-lane group g = l >> 4 covers features 16 (g & 1) .., source lane i' = l & 15 addresses the chunk of
        // sample base + (i' >> 2), feature chunk 4 (g & 1) + (i' & 3)
        const int g = l >> 4, ii = l & 15;
        // (whole-vector bit casts: element-wise __builtin_bit_cast(__bf16, v[e]) of the result compiled to element 0 four times)
        for (int m = 0; m < 2; ++m) {
            s4 v[2];
            for (int r = 0; r < 2; ++r) {
                const int s = 16 * m + 8 * (g >> 1) + 4 * r + (ii >> 2), fg = 4 * (g & 1) + (ii & 3);
                v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S4(lds + chunk_addr(s, fg)));
            }
            A[m] = __builtin_bit_cast(bf8, __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int m = 0; m < 2; ++m) {
        bf8 b;
        for (int u = 0; u < 8; ++u) b[u] = (__bf16)W[(16 * m + 8 * h + u) * 32 + j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], b, c, 0, 0, 0);
    }
    // D layout of 32x32: lane l: column l & 31, rows 8 (r >> 2) + 4 (l >> 5) + (r & 3)
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + j] = c[r];
    if (l == 0) cyc[0] = t1 - t0;
}

int main() {
    short* d_out;
    hipMalloc(&d_out, 256 * sizeof(short));
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<short> out(256);
    hipMemcpy(out.data(), d_out, 256 * sizeof(short), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            const int src = 16 * (l >> 4) + 4 * k + ((l & 15) >> 2);
            const int expect = ((src * 37 + 11) & 255) * 4 + (l & 3);
            if (out[l * 4 + k] != expect) {
                if (bad < 8) printf("  tr map: lane %d elem %d = %d, hypothesis %d\n", l, k, out[l * 4 + k], expect);
                ++bad;
            }
        }
    printf("1. ds_read_b64_tr_b16 mapping: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as assumed", bad);
    if (bad) {
        printf("   raw: lane -> (chunk, element) of each result element\n");
        for (int l = 0; l < 64; ++l) {
            printf("   lane %2d (chunk %3d):", l, (l * 37 + 11) & 255);
            for (int k = 0; k < 4; ++k) {
                const int v = out[l * 4 + k], c = v >> 2;
                int src = -1;
                for (int s = 0; s < 64; ++s) if (((s * 37 + 11) & 255) == c) src = s;
                printf("  [lane %2d].%d", src, v & 3);
            }
            printf("\n");
        }
    }

    std::vector<float> A(16 * 32), B(32 * 16), D(16 * 16), Dr(16 * 16, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)(((i * 7 + k * 3) % 5) - 2);
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B[k * 16 + n] = (float)(((k * 5 + n * 11) % 7) - 3);
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) Dr[i * 16 + n] += A[i * 32 + k] * B[k * 16 + n];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (int q = 0; q < 256; ++q) if (D[q] != Dr[q]) ++bad;
    printf("2. v_mfma_f32_16x16x32_bf16 layout: %s (%d of 256 differ)\n", bad ? "DIFFERENT" : "as assumed", bad);

    std::vector<float> V(32 * 32), W(32 * 32), E(32 * 32), Er(32 * 32, 0.f);
    for (int s = 0; s < 32; ++s) for (int f = 0; f < 32; ++f) V[s * 32 + f] = (float)(((s * 13 + f * 5) % 9) - 4), W[s * 32 + f] = (float)(((s * 3 + f * 17) % 7) - 3);
    for (int f = 0; f < 32; ++f) for (int n = 0; n < 32; ++n) for (int s = 0; s < 32; ++s) Er[f * 32 + n] += V[s * 32 + f] * W[s * 32 + n];
    float *dV, *dW, *dE;
    unsigned long long* dC;
    hipMalloc(&dV, 4096), hipMalloc(&dW, 4096), hipMalloc(&dE, 4096), hipMalloc(&dC, 8);
    hipMemcpy(dV, V.data(), 4096, hipMemcpyHostToDevice), hipMemcpy(dW, W.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, 0, dV, dW, dE, dC);
    hipMemcpy(E.data(), dE, 4096, hipMemcpyDeviceToHost);
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost);
    bad = 0;
    for (int q = 0; q < 1024; ++q) if (E[q] != Er[q]) ++bad;
    printf("3. load layout -> LDS -> transpose reads = MFMA operand: %s (%d of 1024 differ); write + 4 reads: %llu ticks of s_memtime\n",
           bad ? "WRONG" : "right", bad, cyc);
    return 0;
}
