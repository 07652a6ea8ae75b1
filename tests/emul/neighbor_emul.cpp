// TEST INFRASTRUCTURE: the CPU build of marius_amd/csrc/kernels/neighbor.hip (tests/emul/common.h emulates the HIP execution model on host
// threads).  marius_nbr_delta_ids calls marius_sort_unique, which lives in another kernel file: a plain C++ stand-in with the SAME contract
// (include/marius_hip.h) is defined here — it is a dependency of the code under test, not the code under test.
#include <algorithm>
#include <numeric>

#include "common.h"

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

extern "C" size_t marius_sort_unique_workspace_bytes(int64_t) { return 256; }

#include "neighbor.hip.inc"
