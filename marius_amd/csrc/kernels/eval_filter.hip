// Filtered evaluation: the (edge, node) pairs whose scores must be masked because the corrupted triple is a known true edge.
//
// Replaces the global branch of compute_filter_corruption (src/cpp/src/data/samplers/negative.cpp:50-205 on the CPU, :212-293 as a
// chain of libtorch ops on the GPU: searchsorted, repeat_interleave, index_select, masked_select, cat).  Here: one wave per batch edge.
//   pass 1 (count):  binary-search the run of known edges that share the batch edge's uncorrupted endpoint in the list sorted by that
//                    endpoint, count the ones with the same relation                                   -> counts[B]
//   scan:            exclusive prefix sum over the B counts (one workgroup; B is an evaluation batch)   -> offsets[B + 1]
//   pass 2 (emit):   same walk, matching entries written in sorted-list order at offsets[e] + rank     -> filter[F, 2] = (e, corrupted node)
// The output order (edge id, then position in the sorted list) is the reference's.  No atomics.
#include "common.h"

namespace marius {

struct FilterArgs {
    const int64_t* sorted;   // [n_sorted, cols] every known edge, sorted by column key_col
    int64_t n_sorted;
    int cols, key_col, corrupt_col;
    const int64_t* edges;    // [B, cols] batch edges (global ids)
    int64_t B;
};

__device__ __forceinline__ int64_t lower_bound_col(const int64_t* sorted, int64_t n, int cols, int col, int64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted[mid * cols + col] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void true_edge_filter_kernel(FilterArgs a, int64_t* counts, const int64_t* offsets, int64_t* filter) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= a.B) return;
    const int64_t key = a.edges[e * a.cols + a.key_col];
    const int64_t rel = a.cols == 3 ? a.edges[e * a.cols + 1] : 0;
    const int64_t s0 = lower_bound_col(a.sorted, a.n_sorted, a.cols, a.key_col, key);
    const int64_t s1 = lower_bound_col(a.sorted, a.n_sorted, a.cols, a.key_col, key + 1);
    int64_t base = EMIT ? offsets[e] : 0;
    int64_t total = 0;
    for (int64_t p0 = s0; p0 < s1; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool hit = p < s1 && (a.cols != 3 || a.sorted[p * a.cols + 1] == rel);
        const unsigned long long m = __ballot(hit);
        if (EMIT && hit) {
            const int64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
            filter[2 * pos] = e;
            filter[2 * pos + 1] = a.sorted[p * a.cols + a.corrupt_col];
        }
        const int c = __popcll(m);
        base += c;
        total += c;
    }
    if (!EMIT && lane == 0) counts[e] = total;
}

// offsets[0] = 0, offsets[i + 1] = sum_{j <= i} counts[j]; single workgroup, sequential over 1024-element slabs
__global__ __launch_bounds__(1024) void exclusive_scan_small_kernel(const int64_t* counts, int64_t n, int64_t* offsets) {
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) {
        carry = 0;
        offsets[0] = 0;
    }
    __syncthreads();
    for (int64_t b0 = 0; b0 < n; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        part[threadIdx.x] = i < n ? counts[i] : 0;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int64_t v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (i < n) offsets[i + 1] = carry + part[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

}  // namespace marius

using namespace marius;

static int fill(FilterArgs& a, const int64_t* sorted_edges, int64_t n_sorted, int32_t cols, int32_t inverse, const int64_t* edges, int64_t B) {
    MARIUS_REQUIRE(cols == 2 || cols == 3, "true_edge_filter: edge lists must have 3 or 2 columns");
    MARIUS_REQUIRE(n_sorted >= 0 && B >= 0, "true_edge_filter: bad sizes");
    MARIUS_REQUIRE((n_sorted == 0 || sorted_edges) && (B == 0 || edges), "true_edge_filter: null pointer");
    a.sorted = sorted_edges;
    a.n_sorted = n_sorted;
    a.cols = cols;
    a.key_col = inverse ? cols - 1 : 0;      // the endpoint that stays: dst when the source is corrupted, src otherwise
    a.corrupt_col = inverse ? 0 : cols - 1;
    a.edges = edges;
    a.B = B;
    return MARIUS_OK;
}

extern "C" int marius_true_edge_filter_offsets(const int64_t* sorted_edges, int64_t n_sorted, int32_t cols, int32_t inverse, const int64_t* edges,
                                               int64_t B, int64_t* counts, int64_t* offsets, marius_stream_t stream) {
    FilterArgs a;
    int rc = fill(a, sorted_edges, n_sorted, cols, inverse, edges, B);
    if (rc) return rc;
    MARIUS_REQUIRE(offsets && (B == 0 || counts), "true_edge_filter_offsets: null output");
    hipStream_t st = as_stream(stream);
    if (B > 0) true_edge_filter_kernel<false><<<dim3((unsigned)cdiv(B, 4)), dim3(256), 0, st>>>(a, counts, nullptr, nullptr);
    exclusive_scan_small_kernel<<<dim3(1), dim3(1024), 0, st>>>(counts, B, offsets);
    return check_launch("true_edge_filter_offsets");
}

extern "C" int marius_true_edge_filter_emit(const int64_t* sorted_edges, int64_t n_sorted, int32_t cols, int32_t inverse, const int64_t* edges,
                                            int64_t B, const int64_t* offsets, int64_t* filter, marius_stream_t stream) {
    FilterArgs a;
    int rc = fill(a, sorted_edges, n_sorted, cols, inverse, edges, B);
    if (rc) return rc;
    if (B == 0) return MARIUS_OK;
    MARIUS_REQUIRE(offsets && filter, "true_edge_filter_emit: null pointer");
    true_edge_filter_kernel<true><<<dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream)>>>(a, nullptr, offsets, filter);
    return check_launch("true_edge_filter_emit");
}
