// Error state, ABI info and host-only helpers of libmipsf_hip.so.
#include "common.h"

#include <math.h>
#include <string.h>

namespace mipsf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

int device_cus() {
    static int cus[MAX_DEVICES] = {0};
    const int d = device_slot();
    if (cus[d] <= 0) cus[d] = mipsf_device_cu_count();
    return cus[d];
}

}  // namespace mipsf

extern "C" {

const char* mipsf_last_error(void) { return mipsf::g_err; }

int mipsf_abi_version(void) { return MIPSF_ABI_VERSION; }

uint64_t mipsf_buffer_size(int which, uint32_t n, uint32_t a, uint32_t b, const mipsf_grid_meta* meta) {
    switch (which) {
        case MIPSF_SIZE_HASHGRID_BWD_SCRATCH:
        case MIPSF_SIZE_HASHGRID_COUNTER_WORDS: {
            if (!meta) { mipsf::set_error("mipsf_buffer_size: buffer %d needs the grid's level table", which); return ~0ull; }
            const uint64_t v = which == MIPSF_SIZE_HASHGRID_BWD_SCRATCH ? mipsf::hashgrid_bwd_scratch_floats(meta, n, (int)a)
                                                                        : mipsf::hashgrid_counter_words(meta);
            if (v == 0) { mipsf::set_error("mipsf_buffer_size: bad level table"); return ~0ull; }
            return v;
        }
        case MIPSF_SIZE_DECODER_PACKED: return mipsf::decoder_packed_floats();
        case MIPSF_SIZE_DECODER_SAVED: return mipsf::decoder_saved_floats(n);
        case MIPSF_SIZE_DECODER_DACT: return mipsf::decoder_dact_floats(n);
        case MIPSF_SIZE_DECODER_WGRAD_PARTIAL: return mipsf::decoder_wgrad_partial_floats();
        case MIPSF_SIZE_DECODER_PACKED16: return mipsf::decoder_packed16_floats((int)a);
        case MIPSF_SIZE_DECODER_TILE_WORDS: return mipsf::decoder_tile_words(n);
        case MIPSF_SIZE_RENDER_PARTIAL: return mipsf::render_partial_floats(n);
        case MIPSF_SIZE_PLACE_POSE_SCRATCH: return mipsf::place_pose_scratch_floats(a, b, n);
        case MIPSF_SIZE_POSE_RAYS_SCRATCH: return mipsf::pose_rays_scratch_floats(a, b, n);
        default: mipsf::set_error("mipsf_buffer_size: unknown buffer %d", which); return ~0ull;
    }
}

int mipsf_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        mipsf::set_error("hipGetDevice failed (no GPU?)");
        return -1;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        mipsf::set_error("hipDeviceGetAttribute failed");
        return -1;
    }
    return n;
}

// tiny-cuda-nn GridEncodingTemplated constructor + common_device.h grid_scale / grid_resolution.
// exp2f / log2f are evaluated through double so that the result is the correctly rounded fp32 value
// on every libm (the oracle does the same).
int mipsf_hashgrid_meta_init(mipsf_grid_meta* m, uint32_t n_levels, uint32_t n_features,
                             uint32_t log2_hashmap_size, uint32_t base_resolution, double per_level_scale) {
    MIPSF_REQUIRE(m != nullptr, "meta is null");
    MIPSF_REQUIRE(n_levels >= 1 && n_levels <= MIPSF_MAX_LEVELS, "n_levels %u out of range", n_levels);
    MIPSF_REQUIRE(n_features == 2, "only n_features_per_level == 2 is built (got %u)", n_features);
    MIPSF_REQUIRE(log2_hashmap_size >= 4 && log2_hashmap_size <= 28, "log2_hashmap_size %u out of range",
                  log2_hashmap_size);
    memset(m, 0, sizeof(*m));
    m->n_levels = n_levels;
    m->n_features = n_features;
    m->log2_hashmap_size = log2_hashmap_size;
    m->base_resolution = base_resolution;
    const float pls = (float)per_level_scale;
    const float l2 = (float)log2((double)pls);
    m->per_level_scale = pls;
    m->log2_per_level_scale = l2;
    uint64_t off = 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        volatile float arg = (float)l * l2;
        volatile float e = (float)exp2((double)arg);
        volatile float prod = e * (float)base_resolution;
        const float scale = prod - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2;
        uint64_t n = ((double)res * res * res > (double)max_params) ? max_params : (uint64_t)res * res * res;
        n = (n + 7) / 8 * 8;
        const uint64_t cap = 1ull << log2_hashmap_size;
        if (n > cap) n = cap;
        m->scales[l] = scale;
        m->resolutions[l] = res;
        m->offsets[l] = (uint32_t)off;
        off += n;
        MIPSF_REQUIRE(off * n_features < 0xFFFFFFFFull, "grid too large");
    }
    m->offsets[n_levels] = (uint32_t)off;
    m->n_params = (uint32_t)(off * n_features);
    return 0;
}

}  // extern "C"
