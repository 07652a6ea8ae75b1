// Fixed-capacity row exchange of the sharded node table: the device-side halves of the all-to-all row fetch / gradient return
// (a2a_rows_{post,wait}; 8(e): node table sharded along Marius's partition axis, src/storage/storage.cpp:75,
// cross-partition rows by an RCCL all-to-all over xGMI; replaces the per-device model replicas + host-memory embeddings of
// src/cpp/src/pipeline/pipeline_gpu.cpp:23-80).
//
// Why fixed capacity.  An all-to-all(v) needs its split sizes on the HOST, i.e. a device -> host read-back per batch in the training loop
// (round 4: 0.35-0.45 ms of every 0.83 ms step blocked in hipEventSynchronize for them).  Here every (requester, owner) pair owns a block of
// `cap` slots in each payload — cap = the planned maximum, marius_a2a_capacity — the transport is ONE equal-split all-to-all per payload
// (ncclAllToAll / c10d alltoall_base without split vectors: no size ever leaves the device), and the counts ride in the payload itself: the
// unused slots of a block carry the id -1, which every consumer behind this C-ABI already treats as "no row" (gathers skip negative ids; the
// segment plan marks them dead, so the owner's reduction neither reads their gradient rows nor writes anything for them).
//
//   requester                                   owner
//   marius_a2a_rows_post   ids  -> req_send     (all-to-all int64 [world x cap])
//                                               marius_gather_rows(shard, req_recv) -> rows_send            (negative ids skipped)
//                                               marius_merge_unique_runs(req_recv: world runs of cap) + marius_segment_plan   (a step ahead of the gradients)
//   (all-to-all float [world x cap x d])
//   marius_a2a_rows_wait   magnitude bound of the received rows (and, on request, a copy in batch order)
//   ... forward / backward on rows_recv IN PLACE: the batch's local indices were rewritten to payload slots by the post call ...
//   marius_segment_sum_rows_planned(out_rows = place) -> grad_send [world x cap x d]
//   (all-to-all float [world x cap x d])
//                                               marius_segment_adagrad_scatter_group(rows = grad_recv, the plan above)
//
// A block is filled from its END: slots [0, cap - cnt) hold -1, then the cnt local row ids in ascending order — so a block is a
// non-decreasing run under signed comparison and the owner's merge of the `world` runs (marius_merge_unique_runs) needs no counts either.
#include "common.h"

namespace marius {

// one thread per slot of the request payload
__global__ __launch_bounds__(256) void a2a_post_kernel(const int64_t* __restrict__ uniq, const int64_t* __restrict__ offs, int64_t shard_rows, int world,
                                                       int64_t cap, int64_t* __restrict__ req_send, int64_t* __restrict__ place, int32_t* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)world * cap) return;
    const int q = (int)(i / cap);
    const int64_t j = i - (int64_t)q * cap;
    const int64_t o0 = offs[q], cnt = offs[q + 1] - o0;
    if (cnt > cap) {  // more rows of one owner than the planned maximum: flagged (the host refuses to go on), the first `cap` are served
        if (j == 0) atomicOr(reinterpret_cast<unsigned int*>(overflow), 1u);
        // the rows beyond them get slot 0: a defined index inside the payload, so that nothing downstream (slot_of_occ, the in-place scoring of
        // the payload) indexes with an unwritten value in the steps before the host reads the flag
        for (int64_t r = cap + j; r < cnt; r += cap) place[o0 + r] = 0;
    }
    const int64_t c = cnt < cap ? cnt : cap;
    const int64_t k = j - (cap - c);  // position inside the owner's run of the batch's ascending unique ids
    if (k < 0) {
        req_send[i] = -1;
    } else {
        req_send[i] = uniq[o0 + k] - (int64_t)q * shard_rows;
        place[o0 + k] = i;
    }
}

// out[i] = place[inverse[i]]: the payload slot of every occurrence (the batch's local indices — edges_, the negatives' mappings — in slot terms, so
// that the decoder reads the received payload in place instead of a compacted copy of it)
__global__ __launch_bounds__(256) void a2a_slot_of_occ_kernel(const int64_t* __restrict__ place, const int64_t* __restrict__ inverse, int64_t n, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = place[inverse[i]];
}

// emb[u] = rows_recv[place[u]] for u < *num_unique; max |x| of the moved rows max'ed into *absmax.  A row's 16-byte pieces map one to one onto
// lanes (d = 100: 25 lanes per row), four rows in flight per thread row.
template <int VEC>
__global__ __launch_bounds__(256) void a2a_wait_kernel(const float* __restrict__ rows_recv, int64_t recv_ld, const int64_t* __restrict__ place,
                                                       const int64_t* __restrict__ num_unique, int64_t capacity, int vpr, int TX, float* __restrict__ emb,
                                                       int64_t emb_ld, float* __restrict__ absmax) {
    const int TY = 256 / TX, ty = threadIdx.x / TX, tx = threadIdx.x - ty * TX;  // TX need not divide 256: the threads left over move nothing
    int64_t U = *num_unique;
    U = U < capacity ? U : capacity;
    constexpr int UNR = 4;
    float mx = 0.f;
    int64_t row[UNR], src[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
        row[k] = ((int64_t)blockIdx.x * UNR + k) * TY + ty;
        src[k] = (ty < TY && row[k] < U) ? place[row[k]] : -1;
    }
    for (int c = tx; c < vpr; c += TX) {
        float v[UNR][VEC];
#pragma unroll
        for (int k = 0; k < UNR; ++k)
            if (src[k] >= 0) __builtin_memcpy(v[k], rows_recv + src[k] * recv_ld + (int64_t)c * VEC, sizeof(float) * VEC);
#pragma unroll
        for (int k = 0; k < UNR; ++k)
            if (src[k] >= 0) {
                __builtin_memcpy(emb + row[k] * emb_ld + (int64_t)c * VEC, v[k], sizeof(float) * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) mx = fmaxf(mx, fabsf(v[k][e]));
            }
    }
    if (absmax) {
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0 && mx > *reinterpret_cast<volatile float*>(absmax)) atomicMax(reinterpret_cast<unsigned int*>(absmax), __float_as_uint(mx));
    }
}

// ---- the exchange header of a batch, handed to the host in ONE ordered record ------------------------------------------------------------
// An all-to-all(v) needs its split sizes on the host.  Rounds 2-5 moved them with three independent non-blocking device -> host copies (split
// points, receive counts, a stamp) and let the host poll the stamp: correct only as long as the copies become host-visible in stream order,
// which nothing in the loop ever checked — and a stale split vector gives an all-to-all(v) whose two sides disagree: a collective that never
// completes.  Now ONE single-wave kernel writes the whole header straight into the (device-mapped, fine-grained) pinned record:
//   word 0                    stamp      written LAST, system-scope release, after a system fence behind the payload stores
//   words 1 .. W+1            owner split points offs[0 .. W]
//   words W+2 .. 2W+1         rows every requester asks of this rank (recv_counts; = the send counts when there is no count exchange)
//   word 2W+2                 overflow flag of the fixed-capacity form (0 otherwise)
//   word 2W+3                 checksum over stamp and payload (a2a_record_checksum, same function on the host)
// The host acquires the stamp, copies the record out, recomputes the checksum and re-polls on a mismatch: a torn or stale read is then
// something the loop DETECTS and counts (ShardedTrainer::torn_reads_), never something it acts on.
__host__ __device__ inline uint64_t a2a_mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h * 0xff51afd7ed558ccdull;
}

__global__ __launch_bounds__(64) void a2a_publish_kernel(const int64_t* __restrict__ offs, const int64_t* __restrict__ recv_counts,
                                                         const int32_t* __restrict__ overflow, int world, int64_t stamp, int64_t* __restrict__ rec) {
    const int W = world, lane = threadIdx.x;
    uint64_t h = a2a_mix(0x6d617269757361ull, (uint64_t)stamp);  // every lane walks the whole payload for the checksum (W is the rank count: tiny)
    for (int w = 1; w <= 2 * W + 2; ++w) {
        int64_t v;
        if (w <= W + 1) v = offs[w - 1];
        else if (w <= 2 * W + 1) v = recv_counts ? recv_counts[w - W - 2] : offs[w - W - 1] - offs[w - W - 2];
        else v = overflow ? (int64_t)*overflow : 0;
        h = a2a_mix(h, (uint64_t)v);
        if ((w & 63) == lane) __hip_atomic_store(rec + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (((2 * W + 3) & 63) == lane) __hip_atomic_store(rec + 2 * W + 3, (int64_t)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();  // every lane: its payload stores are performed at system scope ...
    __syncthreads();         // ... before lane 0 passes this barrier (release fences are cumulative across it)
    if (lane == 0) __hip_atomic_store(rec, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// offs as owner_offsets_kernel; counts[q] = offs[q + 1] - offs[q] (the all-to-all(v) send counts) in the same launch
__global__ void owner_offsets_counts_kernel(const int64_t* __restrict__ uniq, const int64_t* __restrict__ num_unique, int64_t shard_rows, int P,
                                            int64_t* __restrict__ out, int64_t* __restrict__ counts) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > P) return;
    const int64_t U = *num_unique;
    int64_t bound[2];
    for (int k = 0; k < 2; ++k) {
        const int qq = q + k;
        const int64_t key = (int64_t)qq * shard_rows;
        int64_t lo = 0, hi = U;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (uniq[mid] < key) lo = mid + 1; else hi = mid;
        }
        bound[k] = (qq >= P) ? U : lo;
    }
    out[q] = bound[0];
    if (q < P && counts) counts[q] = bound[1] - bound[0];
}

}  // namespace marius

using namespace marius;

extern "C" int64_t marius_a2a_capacity(int64_t max_rows, int32_t world, double slack) {
    if (max_rows <= 0 || world <= 0) return 0;
    if (world == 1) return max_rows;
    if (!(slack >= 1.0)) slack = 1.0;
    int64_t cap = (int64_t)((double)max_rows / world * slack) + 1;
    cap = (cap + 255) / 256 * 256;  // whole 2 KB runs of ids per pair
    return cap < max_rows ? cap : max_rows;
}

extern "C" int marius_a2a_rows_post(const int64_t* uniq, const int64_t* owner_offsets, int64_t shard_rows, int32_t world, int64_t cap, int64_t* req_send,
                                    int64_t* place, int32_t* overflow_flag, const int64_t* inverse, int64_t n_occ, int64_t* slot_of_occ, marius_stream_t stream) {
    MARIUS_REQUIRE(world >= 1 && cap >= 1 && shard_rows >= 1, "a2a_rows_post: bad sizes world=%d cap=%ld shard_rows=%ld", world, (long)cap, (long)shard_rows);
    MARIUS_REQUIRE(uniq && owner_offsets && req_send && place && overflow_flag, "a2a_rows_post: null pointer");
    MARIUS_REQUIRE(n_occ == 0 || (inverse && slot_of_occ), "a2a_rows_post: null occurrence map");
    const int64_t n = (int64_t)world * cap;
    hipStream_t st = as_stream(stream);
    a2a_post_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st>>>(uniq, owner_offsets, shard_rows, world, cap, req_send, place, overflow_flag);
    if (n_occ > 0) a2a_slot_of_occ_kernel<<<dim3((unsigned)cdiv(n_occ, 256)), dim3(256), 0, st>>>(place, inverse, n_occ, slot_of_occ);
    return check_launch("a2a_rows_post");
}

extern "C" int marius_table_absmax(const float* table, int64_t rows, int64_t ld, int32_t d, float* absmax, marius_stream_t stream);

extern "C" int marius_a2a_rows_wait(const float* rows_recv, int64_t recv_ld, int64_t rows, int32_t d, float* absmax, const int64_t* place,
                                    const int64_t* num_unique_dev, int64_t capacity, float* emb, int64_t emb_ld, marius_stream_t stream) {
    MARIUS_REQUIRE(rows >= 0 && d > 0 && recv_ld >= d, "a2a_rows_wait: bad sizes");
    if (rows == 0) return MARIUS_OK;
    MARIUS_REQUIRE(rows_recv, "a2a_rows_wait: null pointer");
    if (!emb) {  // the payload is scored in place: only the bound (unused slots hold rows of earlier batches or the zeros of the first allocation: the
                 // bound stays an upper bound)
        if (!absmax) return MARIUS_OK;
        return marius_table_absmax(rows_recv, rows, recv_ld, d, absmax, stream);
    }
    MARIUS_REQUIRE(place && num_unique_dev && capacity >= 0 && emb_ld >= d, "a2a_rows_wait: a batch-order copy needs place, the unique count and its capacity");
    if (capacity == 0) return MARIUS_OK;
    int vec = row_vec_width(rows_recv, recv_ld, d);
    const int v2 = row_vec_width(emb, emb_ld, d);
    vec = vec < v2 ? vec : v2;
    const int vpr = d / vec;
    int tx = vpr < 64 ? vpr : 64;
    if (tx < 1) tx = 1;
    const int ty = 256 / tx;
    dim3 block(256), grid((unsigned)cdiv(capacity, (int64_t)ty * 4));
    hipStream_t st = as_stream(stream);
    if (vec == 4) a2a_wait_kernel<4><<<grid, block, 0, st>>>(rows_recv, recv_ld, place, num_unique_dev, capacity, vpr, tx, emb, emb_ld, absmax);
    else if (vec == 2) a2a_wait_kernel<2><<<grid, block, 0, st>>>(rows_recv, recv_ld, place, num_unique_dev, capacity, vpr, tx, emb, emb_ld, absmax);
    else a2a_wait_kernel<1><<<grid, block, 0, st>>>(rows_recv, recv_ld, place, num_unique_dev, capacity, vpr, tx, emb, emb_ld, absmax);
    return check_launch("a2a_rows_wait");
}

extern "C" int32_t marius_a2a_record_words(int32_t world) { return world > 0 ? 2 * world + 4 : 0; }

extern "C" uint64_t marius_a2a_record_checksum(const int64_t* record, int32_t world) {
    uint64_t h = a2a_mix(0x6d617269757361ull, (uint64_t)record[0]);
    for (int w = 1; w <= 2 * world + 2; ++w) h = a2a_mix(h, (uint64_t)record[w]);
    return h;
}

extern "C" int marius_a2a_publish(const int64_t* owner_offsets, const int64_t* recv_counts, const int32_t* overflow_flag, int32_t world, int64_t stamp,
                                  int64_t* record_mapped, marius_stream_t stream) {
    MARIUS_REQUIRE(owner_offsets && record_mapped && world >= 1 && stamp != 0, "a2a_publish: bad arguments (stamp 0 is the unpublished state)");
    a2a_publish_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(owner_offsets, recv_counts, overflow_flag, world, stamp, record_mapped);
    return check_launch("a2a_publish");
}

extern "C" int marius_owner_offsets_counts(const int64_t* uniq, const int64_t* num_unique_dev, int64_t shard_rows, int32_t num_shards, int64_t* out,
                                           int64_t* counts, marius_stream_t stream) {
    MARIUS_REQUIRE(uniq && num_unique_dev && out && shard_rows > 0 && num_shards > 0, "owner_offsets_counts: bad arguments");
    owner_offsets_counts_kernel<<<dim3((unsigned)cdiv(num_shards + 1, 64)), dim3(64), 0, as_stream(stream)>>>(uniq, num_unique_dev, shard_rows, num_shards, out, counts);
    return check_launch("owner_offsets_counts");
}
