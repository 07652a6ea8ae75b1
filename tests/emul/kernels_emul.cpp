// TEST INFRASTRUCTURE: harness of the CPU build (tests/emul/build_emul.py appends one #include per transformed kernel file).  Entry points that live
// in kernel files NOT part of this build but are called by ones that are get plain C++ stand-ins with the SAME contract (include/marius_hip.h):
// they are dependencies of the code under test, not the code under test.
#include <algorithm>
#include <numeric>

#include "common.h"

extern "C" const char* marius_hip_last_error(void) { return marius::g_last_error; }
extern "C" int marius_hip_abi_version(void) { return MARIUS_HIP_ABI_VERSION; }
extern "C" int marius_config_reload(void) {
    marius::g_env = marius::read_env();
    return MARIUS_OK;
}

extern "C" size_t marius_sort_unique_workspace_bytes(int64_t) { return 256; }
extern "C" int marius_sort_unique(const int64_t* ids, int64_t n, int32_t, int64_t* uniq, int64_t* inverse, int32_t* perm, int32_t* seg_offsets,
                                  int64_t* num_unique_dev, void*, size_t, marius_stream_t) {
    std::vector<int32_t> p((size_t)n);
    std::iota(p.begin(), p.end(), 0);
    std::stable_sort(p.begin(), p.end(), [&](int32_t a, int32_t b) { return ids[a] < ids[b]; });
    int64_t U = 0;
    for (int64_t k = 0; k < n; ++k) {
        if (k == 0 || ids[p[k]] != ids[p[k - 1]]) {
            uniq[U] = ids[p[k]];
            seg_offsets[U] = (int32_t)k;
            ++U;
        }
        inverse[p[k]] = U - 1;
        perm[k] = p[k];
    }
    for (int64_t k = U; k < n; ++k) uniq[k] = 0;
    seg_offsets[U] = (int32_t)n;
    *num_unique_dev = U;
    return MARIUS_OK;
}

