// DENSE neighbour sampling and GraphSage aggregation (cfg4: the tail of SURVEY.md §8 (f.4); first slice, round 6).
//
// The reference's device branch is a chain of ATen calls per hop — index_select x2, cumsum, repeat_interleave x3-5, arange, randint, fmod_,
// where, index_select (src/cpp/src/data/graph.cpp:128-236, src/cpp/src/data/samplers/neighbor.cpp:9-17, 81-105) — then a num_nodes-sized bitmap
// zero fill + index_fill + nonzero for the next hop's ids (neighbor.cpp:515-529) and a num_nodes-sized zeros + index_copy + gather for the
// batch-local mapping (graph.cpp:361-398).  Here:
//   marius_nbr_degrees     degrees / CSR offsets of the requested nodes, capped degrees, their exclusive scan and total      (3 launches)
//   marius_nbr_gather      every sampled edge in one launch: owner by binary search in the scan, ALL: k-th neighbour, UNIFORM: start + rand % degree
//   marius_nbr_dropout_*   NeighborSamplingLayer::DROPOUT (neighbor.cpp:236-253): keep flags from the caller's torch::rand draw, ONE scan of the flags gives both
//                          the kept neighbours' output positions (masked_select's order) and the nodes' new local offsets
//   marius_nbr_delta_ids   next hop's ids = ascending unique neighbour ids not yet in the batch: O(batch) — a persistent mark array tested per
//                          candidate, marked candidates mapped to a sentinel key, marius_sort_unique, sentinel dropped — instead of O(num_nodes)
//   marius_nbr_positions   batch-local position of every neighbour through a persistent position table (written for the batch's ids only)
//   marius_segment_gather_sum  GraphSage's a_i: rows gathered by index and summed per segment IN INDEX ORDER (the order of the reference's CPU
//                          index_add_, layer_helpers.cpp:19-30: bit-identical sums, no atomics), optional second list (outgoing + incoming),
//                          MEAN / GCN normalisation (graph_sage_layer.cpp:78-90); with a per-row pre-divisor it is also the backward of the gather
// All integer outputs are bit-exact restatements; the float sums are bit-exact against the CPU op sequence (tests/test_gpu_zz_unverified_cfg4.py).
#include "common.h"

namespace marius {

constexpr int NB_T = 256, NB_ITEMS = 4, NB_TILE = NB_T * NB_ITEMS;

// ---- degrees + per-tile sums
__global__ __launch_bounds__(NB_T) void nbr_degrees_kernel(const int64_t* __restrict__ node_ids, int64_t n, const int64_t* __restrict__ num_tbl,
                                                           const int64_t* __restrict__ off_tbl, int64_t max_neighbors, int64_t* __restrict__ num,
                                                           int64_t* __restrict__ global_offsets, int64_t* __restrict__ capped, int64_t* __restrict__ tile_sums) {
    __shared__ int64_t red[NB_T / 64];
    const int64_t base = (int64_t)blockIdx.x * NB_TILE;
    int64_t mine = 0;
#pragma unroll
    for (int r = 0; r < NB_ITEMS; ++r) {
        const int64_t i = base + r * NB_T + threadIdx.x;
        if (i < n) {
            const int64_t id = node_ids[i];
            const int64_t v = num_tbl[id];
            num[i] = v;
            global_offsets[i] = off_tbl[id];
            const int64_t c = (max_neighbors >= 0 && v > max_neighbors) ? max_neighbors : v;
            capped[i] = c;
            mine += c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t s = 0;
        for (int w = 0; w < NB_T / 64; ++w) s += red[w];
        tile_sums[blockIdx.x] = s;
    }
}

// exclusive scan of the tile sums, in place, by one workgroup; total -> *total
__global__ __launch_bounds__(NB_T) void nbr_scan_tiles_kernel(int64_t* __restrict__ tile_sums, int64_t ntiles, int64_t* __restrict__ total) {
    __shared__ int64_t wsum[NB_T / 64];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b = 0; b < ntiles; b += NB_T) {
        const int64_t i = b + threadIdx.x;
        const int64_t v = i < ntiles ? tile_sums[i] : 0;
        int64_t x = v;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int64_t before = carry + x - v;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < ntiles) tile_sums[i] = before;
        __syncthreads();
        if (threadIdx.x == NB_T - 1) carry = before + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// local_offsets = exclusive scan of capped (tile offset + scan inside the tile)
__global__ __launch_bounds__(NB_T) void nbr_local_offsets_kernel(const int64_t* __restrict__ capped, int64_t n, const int64_t* __restrict__ tile_offsets,
                                                                 int64_t* __restrict__ local_offsets) {
    __shared__ int64_t wsum[NB_T / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // thread t owns NB_ITEMS consecutive elements
    const int64_t i0 = (int64_t)blockIdx.x * NB_TILE + (int64_t)threadIdx.x * NB_ITEMS;
    int64_t v[NB_ITEMS], mine = 0;
#pragma unroll
    for (int r = 0; r < NB_ITEMS; ++r) {
        v[r] = (i0 + r < n) ? capped[i0 + r] : 0;
        mine += v[r];
    }
    int64_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int64_t before = tile_offsets[blockIdx.x] + x - mine;
    for (int w = 0; w < wave; ++w) before += wsum[w];
#pragma unroll
    for (int r = 0; r < NB_ITEMS; ++r) {
        if (i0 + r < n) local_offsets[i0 + r] = before;
        before += v[r];
    }
}

// ---- one thread per sampled edge
__global__ __launch_bounds__(256) void nbr_gather_kernel(const int64_t* __restrict__ sorted_edges, int cols, const int64_t* __restrict__ num,
                                                         const int64_t* __restrict__ global_offsets, const int64_t* __restrict__ local_offsets,
                                                         const int64_t* __restrict__ capped, int64_t n, const int64_t* __restrict__ rand_samples,
                                                         int64_t total, int64_t* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        // owner: the last i with local_offsets[i] <= p (nodes without neighbours share their successor's offset and are skipped by "last")
        int64_t lo = 0, hi = n;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (local_offsets[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t i = lo;
        const int64_t k = p - local_offsets[i];
        const int64_t deg = num[i];
        // neighbor.cpp:94-101: where(degree > max, start + rand % degree, start + k)
        const int64_t idx = global_offsets[i] + ((rand_samples && deg > capped[i]) ? (rand_samples[p] % deg) : k);
        const int64_t* e = sorted_edges + idx * cols;
        int64_t* o = out + p * cols;
        o[0] = e[0];
        o[1] = e[1];
        if (cols == 3) o[2] = e[2];
    }
}

// ---- dropout sampler: keep flags + their tile sums; after the scan of the flags S: output position of a kept neighbour p is S[p] (masked_select's
// order is the global order) and a node's new local offset is S at its first neighbour
__global__ __launch_bounds__(NB_T) void nbr_keep_flags_kernel(const float* __restrict__ keep_rand, int64_t total, float rate, int64_t* __restrict__ keep,
                                                              int64_t* __restrict__ tile_sums) {
    __shared__ int64_t red[NB_T / 64];
    const int64_t base = (int64_t)blockIdx.x * NB_TILE;
    int64_t mine = 0;
#pragma unroll
    for (int r = 0; r < NB_ITEMS; ++r) {
        const int64_t p = base + r * NB_T + threadIdx.x;
        if (p < total) {
            const int64_t k = keep_rand[p] >= rate ? 1 : 0;  // torch::ge(keep_mask, rate): neighbor.cpp:243-244
            keep[p] = k;
            mine += k;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t s = 0;
        for (int w = 0; w < NB_T / 64; ++w) s += red[w];
        tile_sums[blockIdx.x] = s;
    }
}
__global__ __launch_bounds__(256) void nbr_dropout_offsets_kernel(const int64_t* __restrict__ local_offsets, int64_t n, const int64_t* __restrict__ scan, int64_t total,
                                                                  const int64_t* __restrict__ total_kept, int64_t* __restrict__ new_local_offsets) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = local_offsets[i];
        new_local_offsets[i] = lo < total ? scan[lo] : *total_kept;
    }
}
__global__ __launch_bounds__(256) void nbr_dropout_emit_kernel(const int64_t* __restrict__ sorted_edges, int cols, const int64_t* __restrict__ global_offsets,
                                                               const int64_t* __restrict__ local_offsets, int64_t n, const int64_t* __restrict__ keep,
                                                               const int64_t* __restrict__ scan, int64_t total, int64_t* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        if (!keep[p]) continue;
        int64_t lo = 0, hi = n;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (local_offsets[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t* e = sorted_edges + (global_offsets[lo] + (p - local_offsets[lo])) * cols;
        int64_t* o = out + scan[p] * cols;
        o[0] = e[0];
        o[1] = e[1];
        if (cols == 3) o[2] = e[2];
    }
}

// ---- delta ids
__global__ __launch_bounds__(256) void nbr_mark_kernel(const int64_t* __restrict__ ids, int64_t n, uint8_t* __restrict__ marks, uint8_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) marks[ids[i]] = v;
}
// candidate p: column 0 of the incoming edges, then the last column of the outgoing edges; already in the batch -> the sentinel key
__global__ __launch_bounds__(256) void nbr_keys_kernel(const int64_t* __restrict__ in_edges, int64_t n_in, const int64_t* __restrict__ out_edges, int64_t n_out,
                                                       int cols, const uint8_t* __restrict__ marks, int64_t sentinel, int64_t* __restrict__ keys) {
    const int64_t n = n_in + n_out;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = p < n_in ? in_edges[p * cols] : out_edges[(p - n_in) * cols + cols - 1];
        keys[p] = marks[id] ? sentinel : id;
    }
}
__global__ void nbr_drop_sentinel_kernel(int64_t* __restrict__ uniq, int64_t* __restrict__ count, int64_t sentinel) {
    const int64_t U = *count;
    if (U > 0 && uniq[U - 1] == sentinel) {
        uniq[U - 1] = 0;
        *count = U - 1;
    }
}

// ---- positions
__global__ __launch_bounds__(256) void nbr_scatter_positions_kernel(const int64_t* __restrict__ node_ids, int64_t n, int64_t* __restrict__ table) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) table[node_ids[i]] = i;
}
__global__ __launch_bounds__(256) void nbr_gather_positions_kernel(const int64_t* __restrict__ edges, int cols, int col, int64_t T, const int64_t* __restrict__ table,
                                                                   int64_t* __restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) out[t] = table[edges[t * cols + col]];
}

// ---- segmented gather-sum: one wave per output row, lanes stride over the columns; neighbours in index order, four row loads in flight
constexpr int SG_E = 8;  // columns per lane: d <= 64 * SG_E
struct SegList {
    const int64_t* index;    // [T] row of `rows` per entry
    const int64_t* offsets;  // [n] first entry of every segment (ascending; segment s ends where s + 1 starts, the last one at T)
    int64_t T;
};
struct SegGatherArgs {
    const float* rows;
    int64_t rows_ld;
    int d;
    SegList a, b;            // b.index == nullptr: one list
    int64_t n;
    const int64_t* pre_div;  // optional [rows]: every gathered row is divided by (float)pre_div[row index] before it is added
    const int64_t* deg_a;    // optional [n]: normalisation (with deg_b if given): mode 1 MEAN: / where(deg != 0, deg, 1); mode 2 GCN: (sum + self) / (deg + 1)
    const int64_t* deg_b;
    int mode;
    const float* self_rows;  // GCN: [n, self_ld]
    int64_t self_ld;
    float* out;
    int64_t out_ld;
};
__device__ __forceinline__ void sg_accumulate(const SegGatherArgs& A, const SegList& L, int64_t s, int lane, float (&acc)[SG_E]) {
#pragma clang fp contract(off)
    const int64_t t0 = L.offsets[s], t1 = (s + 1 < A.n) ? L.offsets[s + 1] : L.T;
    for (int64_t t = t0; t < t1; t += 4) {
        int64_t r[4];
        float v[4][SG_E];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = (t + u < t1) ? L.index[t + u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < SG_E; ++e) {
                const int c = lane + 64 * e;
                v[u][e] = (r[u] >= 0 && c < A.d) ? A.rows[r[u] * A.rows_ld + c] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r[u] < 0) continue;
            const float div = A.pre_div ? (float)A.pre_div[r[u]] : 1.f;
#pragma unroll
            for (int e = 0; e < SG_E; ++e) acc[e] += A.pre_div ? v[u][e] / div : v[u][e];
        }
    }
}
__global__ __launch_bounds__(256) void segment_gather_sum_kernel(SegGatherArgs A) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= A.n) return;
    float acc[SG_E];
#pragma unroll
    for (int e = 0; e < SG_E; ++e) acc[e] = 0.f;
    sg_accumulate(A, A.a, s, lane, acc);
    if (A.b.index) {  // (sum over the first list) + (sum over the second): graph_sage_layer.cpp:67-71
        float acc2[SG_E];
#pragma unroll
        for (int e = 0; e < SG_E; ++e) acc2[e] = 0.f;
        sg_accumulate(A, A.b, s, lane, acc2);
#pragma unroll
        for (int e = 0; e < SG_E; ++e) acc[e] = acc[e] + acc2[e];
    }
    int64_t deg = 0;
    if (A.deg_a) deg = A.deg_a[s] + (A.deg_b ? A.deg_b[s] : 0);
#pragma unroll
    for (int e = 0; e < SG_E; ++e) {
        const int c = lane + 64 * e;
        if (c >= A.d) continue;
        float x = acc[e];
        if (A.mode == 2) {
            x = x + A.self_rows[s * A.self_ld + c];
            x = x / (float)(deg + 1);
        } else if (A.mode == 1) {
            x = x / (float)(deg != 0 ? deg : 1);
        }
        A.out[s * A.out_ld + c] = x;
    }
}

static inline unsigned nb_blocks(int64_t n, int per) {
    int64_t b = cdiv(n > 0 ? n : 1, per);
    return (unsigned)(b > 65535 * 16 ? 65535 * 16 : b);
}

}  // namespace marius

using namespace marius;

extern "C" size_t marius_nbr_workspace_bytes(int64_t n) { return (size_t)(cdiv(n > 0 ? n : 1, NB_TILE) + 1) * sizeof(int64_t); }

extern "C" int marius_nbr_degrees(const int64_t* node_ids, int64_t n, const int64_t* num_neighbors_tbl, const int64_t* offsets_tbl, int64_t max_neighbors,
                                  int64_t* num, int64_t* global_offsets, int64_t* capped, int64_t* local_offsets, int64_t* total_dev, void* workspace,
                                  size_t workspace_bytes, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && total_dev && workspace, "nbr_degrees: bad arguments");
    MARIUS_REQUIRE(workspace_bytes >= marius_nbr_workspace_bytes(n), "nbr_degrees: workspace too small");
    hipStream_t st = as_stream(stream);
    int64_t* tiles = (int64_t*)workspace;
    const int64_t ntiles = n > 0 ? cdiv(n, NB_TILE) : 0;
    if (n > 0) {
        MARIUS_REQUIRE(node_ids && num_neighbors_tbl && offsets_tbl && num && global_offsets && capped && local_offsets, "nbr_degrees: null pointer");
        nbr_degrees_kernel<<<dim3((unsigned)ntiles), dim3(NB_T), 0, st>>>(node_ids, n, num_neighbors_tbl, offsets_tbl, max_neighbors, num, global_offsets, capped, tiles);
    }
    nbr_scan_tiles_kernel<<<dim3(1), dim3(NB_T), 0, st>>>(tiles, ntiles, total_dev);
    if (n > 0) nbr_local_offsets_kernel<<<dim3((unsigned)ntiles), dim3(NB_T), 0, st>>>(capped, n, tiles, local_offsets);
    return check_launch("nbr_degrees");
}

extern "C" int marius_nbr_gather(const int64_t* sorted_edges, int32_t cols, const int64_t* num, const int64_t* global_offsets, const int64_t* local_offsets,
                                 const int64_t* capped, int64_t n, const int64_t* rand_samples, int64_t total, int64_t* out_edges, marius_stream_t stream) {
    MARIUS_REQUIRE((cols == 2 || cols == 3) && n >= 0 && total >= 0, "nbr_gather: bad arguments");
    if (total == 0) return MARIUS_OK;
    MARIUS_REQUIRE(sorted_edges && num && global_offsets && local_offsets && capped && out_edges && n > 0, "nbr_gather: null pointer");
    nbr_gather_kernel<<<dim3(nb_blocks(total, 256)), dim3(256), 0, as_stream(stream)>>>(sorted_edges, cols, num, global_offsets, local_offsets, capped, n, rand_samples, total, out_edges);
    return check_launch("nbr_gather");
}

extern "C" int marius_nbr_dropout_offsets(const int64_t* local_offsets, int64_t n, int64_t total, const float* keep_rand, float rate, int64_t* keep, int64_t* scan,
                                          int64_t* new_local_offsets, int64_t* total_kept_dev, void* workspace, size_t workspace_bytes, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && total >= 0 && total_kept_dev && workspace, "nbr_dropout_offsets: bad arguments");
    MARIUS_REQUIRE(workspace_bytes >= marius_nbr_workspace_bytes(total), "nbr_dropout_offsets: workspace too small (marius_nbr_workspace_bytes(total))");
    hipStream_t st = as_stream(stream);
    int64_t* tiles = (int64_t*)workspace;
    const int64_t ntiles = total > 0 ? cdiv(total, NB_TILE) : 0;
    if (total > 0) {
        MARIUS_REQUIRE(keep_rand && keep && scan, "nbr_dropout_offsets: null pointer");
        nbr_keep_flags_kernel<<<dim3((unsigned)ntiles), dim3(NB_T), 0, st>>>(keep_rand, total, rate, keep, tiles);
    }
    nbr_scan_tiles_kernel<<<dim3(1), dim3(NB_T), 0, st>>>(tiles, ntiles, total_kept_dev);
    if (total > 0) nbr_local_offsets_kernel<<<dim3((unsigned)ntiles), dim3(NB_T), 0, st>>>(keep, total, tiles, scan);
    if (n > 0) {
        MARIUS_REQUIRE(local_offsets && new_local_offsets, "nbr_dropout_offsets: null pointer");
        nbr_dropout_offsets_kernel<<<dim3(nb_blocks(n, 256)), dim3(256), 0, st>>>(local_offsets, n, scan, total, total_kept_dev, new_local_offsets);
    }
    return check_launch("nbr_dropout_offsets");
}

extern "C" int marius_nbr_dropout_emit(const int64_t* sorted_edges, int32_t cols, const int64_t* global_offsets, const int64_t* local_offsets, int64_t n,
                                       const int64_t* keep, const int64_t* scan, int64_t total, int64_t* out_edges, marius_stream_t stream) {
    MARIUS_REQUIRE((cols == 2 || cols == 3) && n >= 0 && total >= 0, "nbr_dropout_emit: bad arguments");
    if (total == 0) return MARIUS_OK;
    MARIUS_REQUIRE(sorted_edges && global_offsets && local_offsets && keep && scan && out_edges && n > 0, "nbr_dropout_emit: null pointer");
    nbr_dropout_emit_kernel<<<dim3(nb_blocks(total, 256)), dim3(256), 0, as_stream(stream)>>>(sorted_edges, cols, global_offsets, local_offsets, n, keep, scan, total, out_edges);
    return check_launch("nbr_dropout_emit");
}

extern "C" int marius_nbr_delta_ids(const int64_t* in_edges, int64_t n_in, const int64_t* out_edges, int64_t n_out, int32_t cols, const int64_t* node_ids,
                                    int64_t n_node_ids, int64_t num_nodes, uint8_t* marks, int64_t* keys, int64_t* uniq, int64_t* inverse, int32_t* perm,
                                    int32_t* seg_offsets, int64_t* num_unique_dev, void* sort_workspace, size_t sort_workspace_bytes, marius_stream_t stream) {
    MARIUS_REQUIRE((cols == 2 || cols == 3) && n_in >= 0 && n_out >= 0 && n_node_ids >= 0 && num_nodes > 0 && marks, "nbr_delta_ids: bad arguments");
    hipStream_t st = as_stream(stream);
    const int64_t n = n_in + n_out;
    MARIUS_REQUIRE(n == 0 || keys, "nbr_delta_ids: null key buffer");
    if (n_node_ids > 0) nbr_mark_kernel<<<dim3(nb_blocks(n_node_ids, 256)), dim3(256), 0, st>>>(node_ids, n_node_ids, marks, (uint8_t)1);
    if (n > 0) nbr_keys_kernel<<<dim3(nb_blocks(n, 256)), dim3(256), 0, st>>>(in_edges, n_in, out_edges, n_out, cols, marks, num_nodes, keys);
    int bits = 1;
    while (bits < 63 && (1ll << bits) <= num_nodes) ++bits;  // the sentinel key num_nodes must fit
    int rc = marius_sort_unique(keys, n, bits, uniq, inverse, perm, seg_offsets, num_unique_dev, sort_workspace, sort_workspace_bytes, stream);
    if (rc) return rc;
    if (n > 0) nbr_drop_sentinel_kernel<<<dim3(1), dim3(1), 0, st>>>(uniq, num_unique_dev, num_nodes);
    if (n_node_ids > 0) nbr_mark_kernel<<<dim3(nb_blocks(n_node_ids, 256)), dim3(256), 0, st>>>(node_ids, n_node_ids, marks, (uint8_t)0);  // the mark array is all zero again
    return check_launch("nbr_delta_ids");
}

extern "C" int marius_nbr_positions(const int64_t* node_ids, int64_t n, const int64_t* edges, int32_t cols, int32_t col, int64_t T, int64_t* table, int64_t* out,
                                    marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && T >= 0 && (cols == 2 || cols == 3) && col >= 0 && col < cols && table, "nbr_positions: bad arguments");
    hipStream_t st = as_stream(stream);
    if (n > 0) nbr_scatter_positions_kernel<<<dim3(nb_blocks(n, 256)), dim3(256), 0, st>>>(node_ids, n, table);
    if (T > 0) {
        MARIUS_REQUIRE(edges && out, "nbr_positions: null pointer");
        nbr_gather_positions_kernel<<<dim3(nb_blocks(T, 256)), dim3(256), 0, st>>>(edges, cols, col, T, table, out);
    }
    return check_launch("nbr_positions");
}

extern "C" int marius_segment_gather_sum(const float* rows, int64_t rows_ld, int32_t d, const int64_t* index_a, const int64_t* offsets_a, int64_t T_a,
                                         const int64_t* index_b, const int64_t* offsets_b, int64_t T_b, int64_t n, const int64_t* pre_div, const int64_t* deg_a,
                                         const int64_t* deg_b, int32_t mode, const float* self_rows, int64_t self_ld, float* out, int64_t out_ld,
                                         marius_stream_t stream) {
    MARIUS_REQUIRE(d > 0 && d <= 64 * SG_E && n >= 0 && T_a >= 0 && T_b >= 0 && mode >= 0 && mode <= 2, "segment_gather_sum: bad arguments (d <= %d)", 64 * SG_E);
    if (n == 0) return MARIUS_OK;
    MARIUS_REQUIRE(rows && offsets_a && out && (T_a == 0 || index_a) && (!index_b || offsets_b), "segment_gather_sum: null pointer");
    MARIUS_REQUIRE(mode == 0 || deg_a, "segment_gather_sum: MEAN / GCN need the neighbour counts");
    MARIUS_REQUIRE(mode != 2 || self_rows, "segment_gather_sum: GCN needs the self rows");
    SegGatherArgs A;
    A.rows = rows;
    A.rows_ld = rows_ld;
    A.d = d;
    A.a = SegList{index_a, offsets_a, T_a};
    A.b = SegList{index_b, offsets_b, T_b};
    A.n = n;
    A.pre_div = pre_div;
    A.deg_a = deg_a;
    A.deg_b = deg_b;
    A.mode = mode;
    A.self_rows = self_rows;
    A.self_ld = self_ld;
    A.out = out;
    A.out_ld = out_ld;
    segment_gather_sum_kernel<<<dim3((unsigned)cdiv(n, 4)), dim3(256), 0, as_stream(stream)>>>(A);
    return check_launch("segment_gather_sum");
}
