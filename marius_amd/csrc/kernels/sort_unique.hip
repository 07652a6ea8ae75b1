// map_tensors on the device: stable LSD radix sort of (id, position) + run-head scan.
//
// n is ~2B + 2CN (200k for a Freebase86m batch) and ids need only ceil(log2(num_nodes)) key bits (27), so the sort is
// a handful of short passes; rocPRIM (AMD's native primitives, header-only) provides the radix passes and the scan,
// the run-head / inverse / segment-offset kernels are ours.  Outputs:
//   uniq[u]        ascending unique ids                       (Batch::unique_node_indices_)
//   inverse[p]     index into uniq of input position p        (the mapped tensors of map_tensors)
//   perm[k]        input position of the k-th sorted id       (stable: equal ids keep input order => deterministic sums)
//   seg_offsets[u] first sorted position of run u, seg_offsets[U] = n
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace marius {

__global__ __launch_bounds__(256) void head_flags_kernel(const int64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ flags) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void emit_unique_kernel(const int64_t* __restrict__ keys, const int32_t* __restrict__ scan,
                                                          const int32_t* __restrict__ perm, int64_t n, int64_t* __restrict__ uniq,
                                                          int64_t* __restrict__ inverse, int32_t* __restrict__ seg_offsets,
                                                          int64_t* __restrict__ num_unique) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t u = scan[k] - 1;
        const bool head = (k == 0) || (scan[k - 1] != scan[k]);
        inverse[perm[k]] = u;
        if (head) {
            uniq[u] = keys[k];
            seg_offsets[u] = (int32_t)k;
        }
        if (k == n - 1) {
            seg_offsets[u + 1] = (int32_t)n;
            *num_unique = (int64_t)u + 1;
        }
    }
}

__global__ void zero_count_kernel(int64_t* num_unique, int32_t* seg_offsets) {
    *num_unique = 0;
    seg_offsets[0] = 0;
}

// first position in uniq[0..U) whose id >= q * shard_rows, for q = 0..P  (owner bucket boundaries of a sorted id list)
__global__ void owner_offsets_kernel(const int64_t* __restrict__ uniq, const int64_t* __restrict__ num_unique, int64_t shard_rows,
                                     int P, int64_t* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > P) return;
    const int64_t U = *num_unique;
    const int64_t key = (int64_t)q * shard_rows;
    int64_t lo = 0, hi = U;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (uniq[mid] < key) lo = mid + 1; else hi = mid;
    }
    out[q] = (q == P) ? U : lo;
}

// ids = concatenation of `nruns` ascending runs (each run strictly increasing: one sender's unique ids).  Sorted position of element i of
// run r with value v = (i - start_r) + sum over runs q < r of #{x in q : x <= v} + sum over runs q > r of #{x in q : x < v}: exactly the
// position a stable sort by (value, input position) gives it, found with nruns - 1 binary searches instead of radix passes.
constexpr int MERGE_MAX_RUNS = 64;
struct RunOffsets {
    int64_t off[MERGE_MAX_RUNS + 1];
};
__global__ __launch_bounds__(256) void merge_rank_kernel(const int64_t* __restrict__ ids, int64_t n, RunOffsets ro, int nruns, uint64_t* __restrict__ keys,
                                                         int32_t* __restrict__ perm) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int r = 0;
        while (r + 1 < nruns && i >= ro.off[r + 1]) ++r;
        const int64_t v = ids[i];
        int64_t rank = i - ro.off[r];
        for (int q = 0; q < nruns; ++q) {
            if (q == r) continue;
            int64_t lo = ro.off[q], hi = ro.off[q + 1];
            const int64_t base = lo;
            if (q < r) {  // upper bound: elements <= v
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (ids[mid] <= v) lo = mid + 1; else hi = mid;
                }
            } else {      // lower bound: elements < v
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (ids[mid] < v) lo = mid + 1; else hi = mid;
                }
            }
            rank += lo - base;
        }
        keys[rank] = (uint64_t)v;
        perm[rank] = (int32_t)i;
    }
}

struct SortPlan {
    size_t keys_off, scan_off, temp_off, temp_bytes, total;
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int make_plan(int64_t n, SortPlan& p) {
    size_t sort_tmp = 0, scan_tmp = 0;
    int64_t nn = n > 0 ? n : 1;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                             rocprim::counting_iterator<int32_t>(0), (int32_t*)nullptr, (size_t)nn, 0u, 64u,
                                             (hipStream_t)0);
    if (e != hipSuccess) return MARIUS_ERR_HIP;
    e = rocprim::inclusive_scan(nullptr, scan_tmp, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)nn, rocprim::plus<int32_t>(),
                                (hipStream_t)0);
    if (e != hipSuccess) return MARIUS_ERR_HIP;
    p.keys_off = 0;
    p.scan_off = align_up((size_t)nn * 8, 256);
    p.temp_off = p.scan_off + align_up((size_t)nn * 4 * 2, 256);  // flags + scan
    p.temp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    p.total = p.temp_off + align_up(p.temp_bytes, 256) + 256;
    return MARIUS_OK;
}

}  // namespace marius

using namespace marius;

extern "C" size_t marius_sort_unique_workspace_bytes(int64_t n) {
    SortPlan p;
    if (make_plan(n, p) != MARIUS_OK) return 0;
    return p.total;
}

extern "C" int marius_sort_unique(const int64_t* ids, int64_t n, int32_t key_bits, int64_t* uniq, int64_t* inverse,
                                  int32_t* perm, int32_t* seg_offsets, int64_t* num_unique_dev, void* workspace,
                                  size_t workspace_bytes, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && n < (1ll << 31) && key_bits > 0 && key_bits <= 64, "sort_unique: bad n/key_bits");
    MARIUS_REQUIRE(seg_offsets && num_unique_dev, "sort_unique: null outputs");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        zero_count_kernel<<<1, 1, 0, st>>>(num_unique_dev, seg_offsets);
        return check_launch("sort_unique(empty)");
    }
    MARIUS_REQUIRE(ids && uniq && inverse && perm && workspace, "sort_unique: null pointer");
    SortPlan p;
    if (make_plan(n, p) != MARIUS_OK) {
        set_last_error("sort_unique: rocprim size query failed");
        return MARIUS_ERR_HIP;
    }
    MARIUS_REQUIRE(workspace_bytes >= p.total, "sort_unique: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    ProfScope ps(PROF_SORT_UNIQUE, st);
    // uniq[U..n) reads as id 0 so that capacity-sized gathers downstream stay in bounds without a host sync on U
    if (hipMemsetAsync(uniq, 0, (size_t)n * sizeof(int64_t), st) != hipSuccess) {
        set_last_error("sort_unique: memset failed");
        return MARIUS_ERR_HIP;
    }
    char* ws = (char*)workspace;
    uint64_t* keys = (uint64_t*)(ws + p.keys_off);
    int32_t* flags = (int32_t*)(ws + p.scan_off);
    int32_t* scan = flags + n;
    void* tmp = ws + p.temp_off;
    size_t tmp_bytes = p.temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, (const uint64_t*)ids, keys, rocprim::counting_iterator<int32_t>(0),
                                             perm, (size_t)n, 0u, (unsigned)key_bits, st);
    if (e != hipSuccess) {
        set_last_error("sort_unique: radix_sort_pairs: %s", hipGetErrorString(e));
        return MARIUS_ERR_HIP;
    }
    int64_t blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    head_flags_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const int64_t*)keys, n, flags);
    tmp_bytes = p.temp_bytes;
    e = rocprim::inclusive_scan(tmp, tmp_bytes, flags, scan, (size_t)n, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) {
        set_last_error("sort_unique: inclusive_scan: %s", hipGetErrorString(e));
        return MARIUS_ERR_HIP;
    }
    emit_unique_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const int64_t*)keys, scan, perm, n, uniq, inverse,
                                                                    seg_offsets, num_unique_dev);
    return check_launch("sort_unique");
}

extern "C" int marius_merge_unique_runs(const int64_t* ids, int64_t n, const int64_t* run_offsets_host, int32_t num_runs, int64_t* uniq, int64_t* inverse,
                                        int32_t* perm, int32_t* seg_offsets, int64_t* num_unique_dev, void* workspace, size_t workspace_bytes,
                                        marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && n < (1ll << 31) && num_runs >= 1 && num_runs <= MERGE_MAX_RUNS, "merge_unique_runs: bad n / num_runs");
    MARIUS_REQUIRE(seg_offsets && num_unique_dev && run_offsets_host, "merge_unique_runs: null outputs");
    MARIUS_REQUIRE(run_offsets_host[0] == 0 && run_offsets_host[num_runs] == n, "merge_unique_runs: run offsets must span [0, n]");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        zero_count_kernel<<<1, 1, 0, st>>>(num_unique_dev, seg_offsets);
        return check_launch("merge_unique_runs(empty)");
    }
    MARIUS_REQUIRE(ids && uniq && inverse && perm && workspace, "merge_unique_runs: null pointer");
    SortPlan p;
    if (make_plan(n, p) != MARIUS_OK) {
        set_last_error("merge_unique_runs: rocprim size query failed");
        return MARIUS_ERR_HIP;
    }
    MARIUS_REQUIRE(workspace_bytes >= p.total, "merge_unique_runs: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    RunOffsets ro;
    for (int q = 0; q <= num_runs; ++q) {
        ro.off[q] = run_offsets_host[q];
        MARIUS_REQUIRE(q == 0 || ro.off[q] >= ro.off[q - 1], "merge_unique_runs: run offsets must ascend");
    }
    ProfScope ps(PROF_SORT_UNIQUE, st);
    if (hipMemsetAsync(uniq, 0, (size_t)n * sizeof(int64_t), st) != hipSuccess) {
        set_last_error("merge_unique_runs: memset failed");
        return MARIUS_ERR_HIP;
    }
    char* ws = (char*)workspace;
    uint64_t* keys = (uint64_t*)(ws + p.keys_off);
    int32_t* flags = (int32_t*)(ws + p.scan_off);
    int32_t* scan = flags + n;
    int64_t blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    merge_rank_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(ids, n, ro, num_runs, keys, perm);
    head_flags_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const int64_t*)keys, n, flags);
    size_t tmp_bytes = p.temp_bytes;
    hipError_t e = rocprim::inclusive_scan(ws + p.temp_off, tmp_bytes, flags, scan, (size_t)n, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) {
        set_last_error("merge_unique_runs: inclusive_scan: %s", hipGetErrorString(e));
        return MARIUS_ERR_HIP;
    }
    emit_unique_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const int64_t*)keys, scan, perm, n, uniq, inverse, seg_offsets, num_unique_dev);
    return check_launch("merge_unique_runs");
}

extern "C" int marius_owner_offsets(const int64_t* uniq, const int64_t* num_unique_dev, int64_t shard_rows, int32_t num_shards,
                                    int64_t* out, marius_stream_t stream) {
    MARIUS_REQUIRE(uniq && num_unique_dev && out && shard_rows > 0 && num_shards > 0, "owner_offsets: bad arguments");
    owner_offsets_kernel<<<dim3((unsigned)cdiv(num_shards + 1, 64)), dim3(64), 0, as_stream(stream)>>>(uniq, num_unique_dev, shard_rows,
                                                                                                    num_shards, out);
    return check_launch("owner_offsets");
}
