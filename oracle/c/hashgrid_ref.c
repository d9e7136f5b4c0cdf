/* Plain-C restatement of the tiny-cuda-nn 1.7 hash-grid forward (grid.h kernel_grid, common_device.h
 * grid_scale / grid_resolution / pos_fract / grid_index / coherent_prime_hash).
 *
 * TEST INFRASTRUCTURE (oracle): an implementation with real uint32 wrap-around and a real fmaf, independent of
 * the int64-masked / fp64-emulated arithmetic of oracle/tcnn_cpu.py; tests require the two to agree bit for bit
 * on the corner indices.  *** parity unpinned *** like the rest of the tcnn restatement (tinycudann is absent
 * from /root/reference and the reference holds no vector at this boundary).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC hashgrid_ref.c -o libhashgrid_ref.so -lm   (done by build())
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct {
    uint32_t n_levels, n_features;
    const uint32_t* offsets;      /* n_levels + 1, in entries */
    const uint32_t* resolutions;  /* n_levels */
    const float* scales;          /* n_levels */
} grid_desc;

static uint32_t grid_index(const uint32_t p[3], uint32_t size, uint32_t res) {
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < 3 && stride <= size; ++d) {
        index += p[d] * stride;
        stride *= res;
    }
    if (size < stride) index = (p[0] * 1u) ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u);
    return index % size;
}

/* x [M,3] -> idx [M, L, 8] (entry inside the level), y [M, L*2] (may be NULL) */
void hashgrid_ref(const float* x, const float* params, const grid_desc* g, uint32_t M, uint32_t* idx, float* y) {
    for (uint32_t i = 0; i < M; ++i) {
        for (uint32_t l = 0; l < g->n_levels; ++l) {
            const uint32_t off = g->offsets[l], size = g->offsets[l + 1] - off, res = g->resolutions[l];
            const float scale = g->scales[l];
            uint32_t cell[3];
            float frac[3];
            for (int d = 0; d < 3; ++d) {
                float pos = fmaf(scale, x[3 * i + d], 0.5f);
                const float fl = floorf(pos);
                cell[d] = (uint32_t)(int)fl;
                frac[d] = pos - fl;
            }
            float acc[2] = {0.f, 0.f};
            for (int c = 0; c < 8; ++c) {
                float w = 1.f;
                uint32_t p[3];
                for (int d = 0; d < 3; ++d) {
                    if ((c >> d) & 1) { w *= frac[d]; p[d] = cell[d] + 1u; }
                    else { w *= 1.f - frac[d]; p[d] = cell[d]; }
                }
                const uint32_t e = grid_index(p, size, res);
                idx[((size_t)i * g->n_levels + l) * 8 + c] = e;
                if (y) {
                    acc[0] = fmaf(w, params[2 * ((size_t)off + e)], acc[0]);
                    acc[1] = fmaf(w, params[2 * ((size_t)off + e) + 1], acc[1]);
                }
            }
            if (y) { y[(size_t)i * g->n_levels * 2 + 2 * l] = acc[0]; y[(size_t)i * g->n_levels * 2 + 2 * l + 1] = acc[1]; }
        }
    }
}
