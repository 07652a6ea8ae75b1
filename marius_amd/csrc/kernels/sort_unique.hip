// map_tensors on the device: stable LSD radix sort of (id, position) + run-head scan.
//
// n is ~2B + 2CN (200k for a Freebase86m batch) and ids need only ceil(log2(num_nodes)) key bits (27), so the sort is
// a handful of short passes; rocPRIM (AMD's native primitives, header-only) provides the radix passes and the scan,
// the run-head / inverse / segment-offset kernels are ours.  Outputs:
//   uniq[u]        ascending unique ids                       (Batch::unique_node_indices_)
//   inverse[p]     index into uniq of input position p        (the mapped tensors of map_tensors)
//   perm[k]        input position of the k-th sorted id       (stable: equal ids keep input order => deterministic sums)
//   seg_offsets[u] first sorted position of run u, seg_offsets[U] = n
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "seg_plan.h"

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
// zero_uniq / zero_state (optional): what the emit launch that follows expects to find zero (uniq[] tail, head-count granules)
__global__ __launch_bounds__(256) void merge_rank_kernel(const int64_t* __restrict__ ids, int64_t n, RunOffsets ro, int nruns, uint64_t* __restrict__ keys,
                                                         int32_t* __restrict__ perm, int64_t* __restrict__ zero_uniq, uint32_t* __restrict__ zero_state,
                                                         int64_t nstate, uint32_t* __restrict__ zero_ctr) {
    if (zero_ctr && blockIdx.x == 0 && threadIdx.x == 0) *zero_ctr = 0u;  // the tile counter the emit launch that follows draws from
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (zero_uniq) zero_uniq[i] = 0;
        if (zero_state && i < nstate) zero_state[i] = 0u;
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


// ------------------------------------------------------------------------------------------------ hand-written LSD radix sort
// n is small (2e5 keys, 27 key bits: 2.4 MB of keys + positions), so the sort is latency- and launch-bound, not bandwidth-bound:
// what matters is the number of dependent launches and how little each does.  "Onesweep" structure with 9-bit digits (3 passes for
// 27 bits, 2 for the 14-bit relation ids):
//   rs_ghist_kernel    ONE launch: the global digit histograms of all passes (key order does not matter for them); also zeroes the
//                      granules and the output uniq[] (capacity-sized consumers read ids past the unique count as 0)
//   rs_sweep_kernel    one launch per pass: a block ranks its tile stably (wave-level match on the digit + per-wave running counters),
//                      publishes its per-digit counts as flagged 4-byte granules, sums the granules of the tiles before it (they were
//                      dispatched earlier: a spin on a granule always ends) and scatters (key, position).  No per-tile histogram
//                      pass, no scan launch.
//   rs_emit_kernel     ONE launch: run heads -> unique index (same granule hand-off for the head counts of earlier tiles), uniq /
//                      inverse / seg_offsets / count.
// 1 + passes + 1 launches per call (5 for node ids), no memset, instead of rocPRIM's 14.
constexpr int RS_BITS = 9, RS_RADIX = 1 << RS_BITS, RS_THREADS = 256, RS_WAVES = RS_THREADS / 64, RS_ITEMS = 16, RS_TILE = RS_THREADS * RS_ITEMS;
constexpr int SW_THREADS = 512, SW_WAVES = SW_THREADS / 64, SW_ITEMS = RS_TILE / SW_THREADS;  // rs_sweep_kernel: two waves per SIMD share the ranking
constexpr int RS_MAX_PASSES = 4, RS_MAX_KEY_BITS = RS_MAX_PASSES * RS_BITS;  // beyond that the rocPRIM path stays (never here: ids < 2^36)
constexpr int RS_MAX_TILES = 512;  // every block reads the counts of the tiles before it: quadratic, the library sort takes over beyond 2M keys
constexpr int EM_ITEMS = 4, EM_TILE = RS_THREADS * EM_ITEMS;  // rs_emit_kernel: small tiles, the whole chip
constexpr uint32_t RS_FLAG = 0x80000000u;

// Control block at offset 0 of the workspace (zero when the workspace is first used; every call leaves `ready` and ctr[0] zero again and
// zeroes the other counters itself before its later launches draw from them):
//   ready   the ticket hand-shake of rs_ghist_kernel
//   ctr[k]  tile counter of the k-th launch of a call.  A workgroup's tile is the value it draws from the counter, not blockIdx: a tile
//           then only ever waits for tiles whose workgroups have already started (HIP promises no dispatch order between workgroups, and
//           these launches share the device with persistent kernels on other streams), which is what makes the spins below terminate.
struct SortCtl {
    unsigned long long ready;
    uint32_t ctr[8];
};
constexpr size_t SORT_CTL_BYTES = 256;

__device__ __forceinline__ int rs_draw_tile(uint32_t* ctr) {
    __shared__ int s_tile;
    if (threadIdx.x == 0) s_tile = (int)atomicAdd(ctr, 1u);
    __syncthreads();
    return s_tile;
}

__device__ __forceinline__ void rs_publish(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v | RS_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t rs_poll(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// COH = true (the fused launch): data one phase writes and a later phase reads inside the SAME launch moves with agent-scope accesses (sc1: written
// through to memory, read past the XCD-local L2) instead of being bracketed by agent-scope fences.  A fence there is a whole-cache operation
// (buffer_wbl2 / buffer_inv) executed per work item by every wave — measured: the launch took 226 us, twice the separate launches it replaces, and
// each invalidate also drops the lines the main stream's kernels keep L2-resident.  COH = false (the separate launches): plain accesses, the
// kernel boundary is the fence.
template <bool COH, typename T>
__device__ __forceinline__ T rs_ld(const T* p) {
    if constexpr (COH) return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH, typename T>
__device__ __forceinline__ void rs_st(T* p, T v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Every spin of the fused launch is BOUNDED by the constant 100 MHz clock (a wait that lasts RS_WAIT_TICKS = 50 ms is four orders of magnitude past
// a whole launch): the waiter then gives up, leaves a record {where, phase, tile, what it waited for, last value seen} in the control block and the
// launch reports the impossible unique count -1 instead of hanging the device.  err == nullptr (the separate launches): unbounded, as before.
constexpr unsigned long long RS_WAIT_TICKS = 5000000ull;
constexpr int RS_ERR_RECORDS = 6;
struct RsErr {
    uint32_t n;
    uint32_t rec[RS_ERR_RECORDS][4];
};
struct RsWatch {
    RsErr* err;
    int phase, tile;
    unsigned long long t0;
    uint32_t it;
    __device__ __forceinline__ RsWatch(RsErr* e, int ph, int tl) : err(e), phase(ph), tile(tl), t0(0), it(0) {}
    // true: give up
    __device__ __forceinline__ bool expired(int where, int what, uint32_t seen) {
        if (!err) return false;
        if ((++it & 255u) != 0) return false;
        const unsigned long long now = wall_clock64();
        if (t0 == 0) {
            t0 = now;
            return false;
        }
        if (now - t0 < RS_WAIT_TICKS) return false;
        const uint32_t k = __hip_atomic_fetch_add(&err->n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k < (uint32_t)RS_ERR_RECORDS) {
            err->rec[k][0] = (uint32_t)where | ((uint32_t)phase << 8);
            err->rec[k][1] = (uint32_t)tile;
            err->rec[k][2] = (uint32_t)what;
            err->rec[k][3] = seen;
        }
        return true;
    }
};

// ghist: [passes][RS_RADIX] global digit histograms.  Nothing is zeroed by a launch of its own: workgroup 0 zeroes ghist and then raises
// `ready` to this call's ticket (a process-wide counter: no earlier call, no stale workspace content and no graph replay — rs_emit_kernel
// puts 0 back — can hold the same value); the other workgroups build their LDS histograms meanwhile and wait for the ticket before their
// global atomics.  Also zeroes what the later kernels expect to find zero: the granules of this tile (state: [passes][ntiles][RS_RADIX];
// tile_state: EM tiles) and uniq[].
__global__ __launch_bounds__(SW_THREADS) void rs_ghist_kernel(const uint64_t* __restrict__ keys, int64_t n, int passes, int ntiles, uint32_t* __restrict__ ghist,
                                                              uint32_t* __restrict__ state, uint32_t* __restrict__ tile_state, int64_t* __restrict__ uniq,
                                                              SortCtl* __restrict__ ctl, unsigned long long ticket) {
    __shared__ uint32_t h[RS_MAX_PASSES][RS_RADIX];
    unsigned long long* ready = &ctl->ready;
    const int tile = rs_draw_tile(&ctl->ctr[0]);
    if (tile == 0) {
        if (threadIdx.x >= 1 && threadIdx.x < 8) ctl->ctr[threadIdx.x] = 0u;  // the later launches of this call start counting at 0
        for (int b = threadIdx.x; b < passes * RS_RADIX; b += SW_THREADS) __hip_atomic_store(ghist + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();  // every lane's stores are issued; the release below orders them before the ticket
        if (threadIdx.x == 0) __hip_atomic_store(ready, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int b = threadIdx.x; b < RS_MAX_PASSES * RS_RADIX; b += SW_THREADS) (&h[0][0])[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)tile * RS_TILE;
#pragma unroll
    for (int r = 0; r < SW_ITEMS; ++r) {
        const int64_t i = base + r * SW_THREADS + threadIdx.x;
        if (i < n) {
            const uint64_t k = keys[i];
            uniq[i] = 0;
            for (int ps = 0; ps < passes; ++ps) atomicAdd(&h[ps][(int)((k >> (ps * RS_BITS)) & (RS_RADIX - 1))], 1u);
        }
    }
    if (threadIdx.x < RS_TILE / EM_TILE) tile_state[tile * (RS_TILE / EM_TILE) + threadIdx.x] = 0;
    for (int ps = 0; ps < passes; ++ps) {
        const size_t row = ((size_t)ps * ntiles + tile) * RS_RADIX;
        for (int b = threadIdx.x; b < RS_RADIX; b += SW_THREADS) state[row + b] = 0;
    }
    if (tile != 0 && threadIdx.x == 0)
        while (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ticket) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    for (int b = threadIdx.x; b < passes * RS_RADIX; b += SW_THREADS) {
        const uint32_t v = (&h[0][0])[b];
        if (v) atomicAdd(&ghist[b], v);
    }
}

// pay_in == nullptr: the payload of key i is i (first pass).  ghist: this pass's global digit histogram; state: [ntiles][RS_RADIX]
// granules of this pass (zero on entry).
struct SweepSmem {
    int32_t off[RS_RADIX];            // global position of this tile's first key of each digit
    int32_t cw[SW_WAVES][RS_RADIX];   // per-wave running digit counts, then per-wave bases
    int32_t wsum[SW_WAVES];
};
// one tile of one pass (SW_THREADS threads).  ghist is read with an agent-scope load either way — in the fused launch it was accumulated by
// atomics of workgroups on other XCDs inside the SAME launch
template <bool COH = false>
__device__ __forceinline__ void rs_sweep_tile(SweepSmem& sm, const int tile, const uint64_t* __restrict__ keys_in, const int32_t* __restrict__ pay_in, int64_t n, int shift,
                                              const uint32_t* __restrict__ ghist, uint32_t* __restrict__ state, uint64_t* __restrict__ keys_out, int32_t* __restrict__ pay_out,
                                              RsErr* err = nullptr, int phase = 0) {
    int32_t (&off)[RS_RADIX] = sm.off;
    int32_t (&cw)[SW_WAVES][RS_RADIX] = sm.cw;
    int32_t (&wsum)[SW_WAVES] = sm.wsum;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t wbase_idx = (int64_t)tile * RS_TILE + (int64_t)wave * 64 * SW_ITEMS;
    uint64_t key[SW_ITEMS];
    int32_t pay[SW_ITEMS];
#pragma unroll
    for (int r = 0; r < SW_ITEMS; ++r) {
        const int64_t i = wbase_idx + r * 64 + lane;
        key[r] = i < n ? rs_ld<COH>(keys_in + i) : 0ull;
        pay[r] = (i < n && pay_in) ? rs_ld<COH>(pay_in + i) : (int32_t)i;
    }
    for (int b = tid; b < SW_WAVES * RS_RADIX; b += SW_THREADS) (&cw[0][0])[b] = 0;
    __syncthreads();
    // ---- stable ranking.  Wave w owns the contiguous keys [w * 64 * ITEMS, (w + 1) * 64 * ITEMS) of the tile, round r the 64 keys at
    // r * 64: positions grow with (wave, round, lane), and so do the ranks handed out below.
    int32_t rank[SW_ITEMS];
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < SW_ITEMS; ++r) {
        const int64_t i = wbase_idx + r * 64 + lane;
        const bool valid = i < n;
        const int d = (int)((key[r] >> shift) & (RS_RADIX - 1));
        uint64_t m = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int bit = 0; bit < RS_BITS; ++bit) {
            const uint64_t bb = __builtin_amdgcn_ballot_w64((d >> bit) & 1);
            m &= ((d >> bit) & 1) ? bb : ~bb;
        }
        const int before = __builtin_popcountll(m & lt), cnt = __builtin_popcountll(m);
        int32_t prev = 0;
        if (valid) prev = cw[wave][d];
        __builtin_amdgcn_wave_barrier();  // every lane has read the counter before its group's leader advances it
        if (valid && before == 0) cw[wave][d] = prev + cnt;
        __builtin_amdgcn_wave_barrier();
        rank[r] = prev + before;
    }
    __syncthreads();
    // ---- thread t owns digit t: per-wave bases, then the tile's count is published (one flagged 4-byte granule) before anything waits
    static_assert(RS_MAX_PASSES + 2 <= 7 && sizeof(SortCtl) <= SORT_CTL_BYTES, "control block");
    static_assert(RS_RADIX == SW_THREADS, "one digit per thread");
    {
        int32_t run = 0;
#pragma unroll
        for (int w = 0; w < SW_WAVES; ++w) {
            const int32_t c = cw[w][tid];
            cw[w][tid] = run;
            run += c;
        }
        rs_publish(state + (int64_t)tile * RS_RADIX + tid, (uint32_t)run);
    }
    // ---- keys of the digit in earlier tiles: their granules, RS_POLL in flight, re-polled until every flag is up
    constexpr int RS_POLL = 24;
    int32_t prefix = 0;
    RsWatch watch(err, phase, tile);
    for (int t0 = 0; t0 < tile; t0 += RS_POLL) {
        uint32_t v[RS_POLL];
        bool done;
        do {
            done = true;
#pragma unroll
            for (int u = 0; u < RS_POLL; ++u) v[u] = (t0 + u < tile) ? rs_poll(state + (int64_t)(t0 + u) * RS_RADIX + tid) : RS_FLAG;
#pragma unroll
            for (int u = 0; u < RS_POLL; ++u) done = done && (v[u] & RS_FLAG);
            if (!done && watch.expired(1, t0, v[0])) break;
        } while (!done);
#pragma unroll
        for (int u = 0; u < RS_POLL; ++u) prefix += (int32_t)(v[u] & ~RS_FLAG);
    }
    // ---- exclusive scan of the global digit totals
    {
        const int32_t total = (int32_t)rs_poll(ghist + tid);
        int32_t x = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int32_t before = x - total;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        off[tid] = before + prefix;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < SW_ITEMS; ++r) {
        const int64_t i = wbase_idx + r * 64 + lane;
        if (i < n) {
            const int d = (int)((key[r] >> shift) & (RS_RADIX - 1));
            const int64_t pos = (int64_t)off[d] + cw[wave][d] + rank[r];
            rs_st<COH>(keys_out + pos, key[r]);
            rs_st<COH>(pay_out + pos, pay[r]);
        }
    }
}

__global__ __launch_bounds__(SW_THREADS) void rs_sweep_kernel(const uint64_t* __restrict__ keys_in, const int32_t* __restrict__ pay_in, int64_t n, int shift,
                                                              const uint32_t* __restrict__ ghist, uint32_t* __restrict__ state, uint64_t* __restrict__ keys_out,
                                                              int32_t* __restrict__ pay_out, uint32_t* __restrict__ tile_ctr) {
    __shared__ SweepSmem sm;
    const int tile = rs_draw_tile(tile_ctr);
    rs_sweep_tile(sm, tile, keys_in, pay_in, n, shift, ghist, state, keys_out, pay_out);
}

// unique index of every sorted position = heads in earlier tiles (granule hand-off) + inclusive scan inside the tile; emits uniq / inverse /
// seg_offsets / count.  tile_state: [ntiles] granules, zero on entry.  THREADS threads, tiles of THREADS * EM_ITEMS positions.
template <int THREADS>
struct EmitSmem {
    int32_t red[THREADS / 64], wsum[THREADS / 64];
};
template <int THREADS, bool COH = false>
__device__ __forceinline__ void rs_emit_tile(EmitSmem<THREADS>& sm, const int tile, const uint64_t* __restrict__ keys, const int32_t* __restrict__ perm, int64_t n,
                                             uint32_t* __restrict__ tile_state, int64_t* __restrict__ uniq, int64_t* __restrict__ inverse, int32_t* __restrict__ seg_offsets,
                                             int64_t* __restrict__ num_unique, RsErr* err = nullptr, int phase = 0) {
    constexpr int WAVES = THREADS / 64, TILE = THREADS * EM_ITEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // thread t owns EM_ITEMS consecutive positions: local head count, block scan of the thread sums
    const int64_t k0 = (int64_t)tile * TILE + (int64_t)tid * EM_ITEMS;
    uint64_t kk[EM_ITEMS];
    int32_t pp[EM_ITEMS];
    bool head[EM_ITEMS];
    uint64_t prev = (k0 > 0 && k0 - 1 < n) ? rs_ld<COH>(keys + (k0 - 1)) : 0ull;
    int32_t mine = 0;
#pragma unroll
    for (int r = 0; r < EM_ITEMS; ++r) {
        const int64_t k = k0 + r;
        kk[r] = k < n ? rs_ld<COH>(keys + k) : 0ull;
        pp[r] = k < n ? rs_ld<COH>(perm + k) : 0;
        head[r] = k < n && (k == 0 || kk[r] != prev);
        prev = kk[r];
        mine += head[r] ? 1 : 0;
    }
    int32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) sm.wsum[wave] = x;
    __syncthreads();
    int32_t tile_total = 0;
    for (int w = 0; w < WAVES; ++w) tile_total += sm.wsum[w];
    if (tid == 0) rs_publish(tile_state + tile, (uint32_t)tile_total);
    // heads in earlier tiles
    int32_t b = 0;
    RsWatch watch(err, phase, tile);
    for (int t = tid; t < tile; t += THREADS) {
        uint32_t v;
        do {
            v = rs_poll(tile_state + t);
        } while (!(v & RS_FLAG) && !watch.expired(2, t, v));
        b += (int32_t)(v & ~RS_FLAG);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
    if (lane == 0) sm.red[wave] = b;
    __syncthreads();
    int32_t base = 0;
    for (int w = 0; w < WAVES; ++w) base += sm.red[w];
    int32_t before = base + x - mine;
    for (int w = 0; w < wave; ++w) before += sm.wsum[w];
    int32_t u = before - 1;  // unique index of the position before this thread's first one
#pragma unroll
    for (int r = 0; r < EM_ITEMS; ++r) {
        const int64_t k = k0 + r;
        if (k < n) {
            if (head[r]) {
                ++u;
                rs_st<COH>(uniq + u, (int64_t)kk[r]);
                rs_st<COH>(seg_offsets + u, (int32_t)k);
            }
            rs_st<COH>(inverse + pp[r], (int64_t)u);
            if (k == n - 1) {
                rs_st<COH>(seg_offsets + (u + 1), (int32_t)n);
                rs_st<COH>(num_unique, (int64_t)u + 1);
            }
        }
    }
}

__global__ __launch_bounds__(RS_THREADS) void rs_emit_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ perm, int64_t n,
                                                             uint32_t* __restrict__ tile_state, int64_t* __restrict__ uniq, int64_t* __restrict__ inverse,
                                                             int32_t* __restrict__ seg_offsets, int64_t* __restrict__ num_unique, SortCtl* __restrict__ ctl, int my_ctr) {
    __shared__ EmitSmem<RS_THREADS> sm;
    const int tile = rs_draw_tile(&ctl->ctr[my_ctr]);
    if (tile == 0 && threadIdx.x == 0) {  // last launch of a call: hand-shake word and the first launch's counter return to zero (a replay of a captured
        ctl->ready = 0ull;                // call waits and counts again).  This launch's own counter, like those of the sweeps, is zeroed by the FIRST launch
        ctl->ctr[0] = 0u;                 // of the next call that uses the workspace (rs_ghist_kernel / merge_rank_kernel), before anything draws from it.
    }
    rs_emit_tile<RS_THREADS>(sm, tile, keys, perm, n, tile_state, uniq, inverse, seg_offsets, num_unique);
}

// ------------------------------------------------------------------------------------------------ one launch for the whole map chain
// marius_prepare_maps (round 6; VERDICT r3-r5: "one persistent sort + plan launch"): assemble ids -> digit histograms -> radix passes -> run heads /
// unique map -> batch-local edges + segment plan, for up to two id lists at once (the batch's node ids and its relation ids), in ONE persistent
// launch instead of 1 + 5 + 1 + 1 (+ 1 + 4 + 1) dependent launches on the preparation stream.
//
// Work = a queue of items (phase, tile), phases in dependency order, the two jobs' phases interleaved.  Every workgroup draws the next item from
// one counter; before it starts an item it waits until the phase the item depends on (the previous phase of the same job) has COMPLETED — a
// per-phase completion counter — with an agent-scope acquire after the wait and an agent-scope release before every completion (the fence
// pair a grid barrier is made of: L2 write-back / invalidate across the XCDs).  No co-residency is assumed anywhere: items are drawn in
// order, so whatever an item waits for — the earlier phase, and inside a pass the granules of earlier tiles — has already been DRAWN by a
// workgroup that is running; a launch with a single workgroup completes (slowly).  The last workgroup to leave puts the control block back
// to zero (replay, reuse).
constexpr int PM_MAX_JOBS = 2, PM_MAX_PHASES = 16;
enum { PM_INIT = 0, PM_HIST = 1, PM_SWEEP = 2, PM_EMIT = 3, PM_PLAN = 4 };
constexpr int PM_EM_TILE = SW_THREADS * EM_ITEMS;  // emit tile of the fused launch (512 threads)
struct PmCtl {  // at byte 64 of job 0's workspace control block (SortCtl sits at byte 0: the two forms can alternate on one workspace)
    uint32_t next, exited;
    uint32_t done[PM_MAX_PHASES];
    RsErr err;  // waits that gave up (marius_prepare_maps_errors): zero after a healthy launch
};
static_assert(64 + sizeof(PmCtl) <= SORT_CTL_BYTES && sizeof(SortCtl) <= 64, "control block");
struct PmJob {
    // where the ids come from: ids_in (given), or assembled into `ids` by the first phase — col < 0: cat(src, dst, src_neg, dst_neg)
    // (marius_assemble_ids order); col >= 0: that column of edges
    const int64_t* ids_in;
    int64_t* ids;
    const int64_t* edges;
    const int64_t* src_neg;
    const int64_t* dst_neg;
    int64_t B, CN, n;
    int cols, col, passes, ntiles, etiles, ptiles, has_plan;
    uint64_t *keys, *keys2;
    int32_t *payA, *payB, *perm;
    uint32_t *ghist, *state, *tile_state;
    int64_t *uniq, *inverse, *count;
    int32_t* seg;
    SegPlanPtrs plan;
    int64_t* edges_out;
};
struct PmPhase {
    int job, kind, pass, first, nitems, dep, dep_items;  // dep: phase this one waits for (-1: none), dep_items: its item count
};
struct PmArgs {
    PmJob job[PM_MAX_JOBS];
    PmPhase ph[PM_MAX_PHASES];
    int nph, total;
    PmCtl* ctl;
};

__device__ __forceinline__ int64_t pm_id(const PmJob& J, int64_t i) {
    if (J.ids_in) return J.ids_in[i];
    if (J.col >= 0) return J.edges[i * J.cols + J.col];
    if (i < J.B) return J.edges[i * J.cols];
    if (i < 2 * J.B) return J.edges[(i - J.B) * J.cols + J.cols - 1];
    const int64_t t = i - 2 * J.B;
    if (J.src_neg) return t < J.CN ? J.src_neg[t] : J.dst_neg[t - J.CN];
    return J.dst_neg[t];
}

union PmSmem {
    uint32_t h[RS_MAX_PASSES][RS_RADIX];
    SweepSmem sw;
    EmitSmem<SW_THREADS> em;
};

// one work item.  (J and ph are picked by STATIC index under uniform branches in the kernel: indexing the kernel-argument struct with a run-time
// value makes hipcc copy it into scratch — the same finding as the grouped segment update, DESIGN_HISTORY 4.1)
__device__ __forceinline__ void pm_item(const PmJob& J, const PmPhase& ph, const int g, const int tile, PmSmem& sm, const PmJob& J0, const PmJob& J1, RsErr* err) {
    const int tid = threadIdx.x;
    if (ph.kind == PM_INIT) {  // what must be zero before any atomic lands: every job's global digit histograms
        if (J0.n > 0)
            for (int b = tid; b < RS_MAX_PASSES * RS_RADIX; b += SW_THREADS) __hip_atomic_store(J0.ghist + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (J1.n > 0)
            for (int b = tid; b < RS_MAX_PASSES * RS_RADIX; b += SW_THREADS) __hip_atomic_store(J1.ghist + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (ph.kind == PM_HIST) {
        // ids of this tile (assembled here when they are not given), the digit histograms of all passes, and the zero fills the later
        // phases expect: this tile's granules of every pass, its emit granules, uniq[]
        for (int b = tid; b < RS_MAX_PASSES * RS_RADIX; b += SW_THREADS) (&sm.h[0][0])[b] = 0;
        __syncthreads();
        const int64_t base = (int64_t)tile * RS_TILE;
#pragma unroll
        for (int r = 0; r < SW_ITEMS; ++r) {
            const int64_t i = base + r * SW_THREADS + tid;
            if (i < J.n) {
                const uint64_t k = (uint64_t)pm_id(J, i);
                if (!J.ids_in) rs_st<true>(J.ids + i, (int64_t)k);
                rs_st<true>(J.uniq + i, (int64_t)0);
                for (int ps = 0; ps < J.passes; ++ps) atomicAdd(&sm.h[ps][(int)((k >> (ps * RS_BITS)) & (RS_RADIX - 1))], 1u);
            }
        }
        constexpr int EPT = RS_TILE / PM_EM_TILE;
        if (tid < EPT && tile * EPT + tid < J.etiles) rs_st<true>(J.tile_state + (tile * EPT + tid), 0u);
        for (int ps = 0; ps < J.passes; ++ps) {
            const size_t row = ((size_t)ps * J.ntiles + tile) * RS_RADIX;
            for (int b = tid; b < RS_RADIX; b += SW_THREADS) rs_st<true>(J.state + (row + b), 0u);
        }
        __syncthreads();
        for (int b = tid; b < J.passes * RS_RADIX; b += SW_THREADS) {
            const uint32_t v = (&sm.h[0][0])[b];
            if (v) atomicAdd(&J.ghist[b], v);
        }
    } else if (ph.kind == PM_SWEEP) {
        // ping-pong so that the last pass lands in (keys, perm)
        const int ps = ph.pass;
        const bool to_keys = ((J.passes - 1 - ps) % 2) == 0;
        const uint64_t* kin = ps == 0 ? (const uint64_t*)(J.ids_in ? J.ids_in : J.ids) : (to_keys ? J.keys2 : J.keys);
        const int32_t* pin = ps == 0 ? nullptr : (to_keys ? J.payB : J.payA);
        uint64_t* kout = to_keys ? J.keys : J.keys2;
        int32_t* pout = (ps == J.passes - 1) ? J.perm : (to_keys ? J.payA : J.payB);
        rs_sweep_tile<true>(sm.sw, tile, kin, pin, J.n, ps * RS_BITS, J.ghist + (size_t)ps * RS_RADIX, J.state + (size_t)ps * J.ntiles * RS_RADIX, kout, pout, err, g);
    } else if (ph.kind == PM_EMIT) {
        rs_emit_tile<SW_THREADS, true>(sm.em, tile, J.keys, J.perm, J.n, J.tile_state, J.uniq, J.inverse, J.seg, J.count, err, g);
    } else {  // PM_PLAN: the batch's edges in batch-local ids (marius_remap_edges) and the index plan of the segmented update (marius_segment_plan)
        const int64_t U = rs_ld<true>(J.count);
        const int64_t base = (int64_t)tile * RS_TILE;
#pragma unroll
        for (int r = 0; r < SW_ITEMS; ++r) {
            const int64_t k = base + r * SW_THREADS + tid;
            if (k < J.n) {
                if (J.has_plan) seg_plan_position<true>(k, J.n, U, J.perm, J.inverse, J.seg, J.uniq, J.plan);
                if (J.edges_out && k < J.B) {  // dataloader.cpp:460-466
                    J.edges_out[k * J.cols] = rs_ld<true>(J.inverse + k);
                    if (J.cols == 3) J.edges_out[k * J.cols + 1] = J.edges[k * J.cols + 1];
                    J.edges_out[k * J.cols + J.cols - 1] = rs_ld<true>(J.inverse + (J.B + k));
                }
            }
        }
    }
}

__global__ __launch_bounds__(SW_THREADS) void prepare_maps_kernel(const PmArgs A) {
    __shared__ PmSmem sm;
    __shared__ int s_item;
    const int tid = threadIdx.x;
    PmCtl* ctl = A.ctl;
    // Loop shape (found the hard way): everything thread 0 does alone — completing the previous item, drawing the next — sits in ONE block at
    // the head, and the drawn item is made a wave-uniform value (readfirstlane) before anything branches on it.  With `item` left as the VGPR an LDS read
    // gives, the exit test is a divergent branch to the compiler; its structurizer then peeled "thread 0: done[g]++ ... draw" into an outer loop and sent
    // the other 63 lanes of wave 0 round the inner loop (barrier, read s_item) without lane 0: the barrier was passed with the OLD s_item and the
    // workgroup repeated its item for ever (one workgroup, one phase was enough to hang the device).
    int g_prev = -1;
    for (;;) {
        if (tid == 0) {
            if (g_prev >= 0) __hip_atomic_fetch_add(&ctl->done[g_prev], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every thread's release fence precedes the barrier that ended the item)
            s_item = (int)atomicAdd(&ctl->next, 1u);
        }
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_item);
        if (item >= A.total) break;
        PmPhase ph = A.ph[0];
        int g = 0;
#pragma unroll
        for (int q = 1; q < PM_MAX_PHASES; ++q)
            if (q < A.nph && item >= A.ph[q].first) {
                ph = A.ph[q];
                g = q;
            }
        const int tile = item - ph.first;
        if (ph.dep >= 0) {
            if (tid == 0) {
                const uint32_t want = (uint32_t)ph.dep_items;
                RsWatch watch(&ctl->err, g, tile);
                uint32_t seen;
                while ((seen = __hip_atomic_load(&ctl->done[ph.dep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (watch.expired(0, ph.dep, seen)) break;
                }
            }
            __syncthreads();  // (no cache invalidate: what the item reads of earlier phases it reads with agent-scope loads — rs_ld<true>)
        }
        if (ph.job == 0) pm_item(A.job[0], ph, g, tile, sm, A.job[0], A.job[1], &ctl->err);
        else pm_item(A.job[1], ph, g, tile, sm, A.job[0], A.job[1], &ctl->err);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // every wave: its (write-through) stores have completed — s_waitcnt vmcnt(0), no cache write-back
        __syncthreads();  // the item is complete; s_item has been read by everybody
        g_prev = g;
    }
    if (tid == 0) {
        // (relaxed throughout: the control block is only ever touched by agent-scope atomics, and the next launch starts after this one has ended)
        const uint32_t e = __hip_atomic_fetch_add(&ctl->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e == gridDim.x - 1) {  // everybody else has left: nobody reads the control block any more
            if (__hip_atomic_load(&ctl->err.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {  // a wait gave up: the maps are not to be used
                rs_st<true>(A.job[0].count, (int64_t)-1);
                if (A.job[1].n > 0) rs_st<true>(A.job[1].count, (int64_t)-1);
            }
            for (int q = 0; q < PM_MAX_PHASES; ++q) __hip_atomic_store(&ctl->done[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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
    p.keys_off = SORT_CTL_BYTES;  // [0, 256): SortCtl, at the same place whatever n a call has
    p.scan_off = p.keys_off + align_up((size_t)nn * 8, 256);
    p.temp_off = p.scan_off + align_up((size_t)nn * 4 * 2, 256);  // flags + scan
    p.temp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    {   // the hand-written sort: second key buffer, two payload buffers, histogram matrix, per-tile head counts
        const size_t tiles = (size_t)(nn + RS_TILE - 1) / RS_TILE;
        const size_t own = align_up((size_t)nn * 8, 256) + 2 * align_up((size_t)nn * 4, 256) +
                           align_up(((size_t)RS_MAX_PASSES * RS_RADIX + (size_t)RS_MAX_PASSES * tiles * RS_RADIX + tiles * (RS_TILE / EM_TILE) + 4) * 4, 256);
        if (tiles <= (size_t)RS_MAX_TILES && own > p.temp_bytes) p.temp_bytes = own;
    }
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
    // uniq[U..n) reads as id 0 so that capacity-sized gathers downstream stay in bounds without a host sync on U (own path: zeroed by
    // rs_ghist_kernel; library path: the memset below)
    char* ws = (char*)workspace;
    uint64_t* keys = (uint64_t*)(ws + p.keys_off);
    int32_t* flags = (int32_t*)(ws + p.scan_off);
    int32_t* scan = flags + n;
    void* tmp = ws + p.temp_off;
    size_t tmp_bytes = p.temp_bytes;
    {
        // (MARIUS_SORT=rocprim: the library chain — A/B runs)
        if (key_bits <= RS_MAX_KEY_BITS && n <= (int64_t)RS_MAX_TILES * RS_TILE && !kernel_env().sort_rocprim) {
            const int ntiles = (int)cdiv(n, RS_TILE);
            const int passes = (key_bits + RS_BITS - 1) / RS_BITS;
            char* t = (char*)tmp;
            uint64_t* keys2 = (uint64_t*)t;
            t += align_up((size_t)n * 8, 256);
            int32_t* payA = (int32_t*)t;
            t += align_up((size_t)n * 4, 256);
            int32_t* payB = (int32_t*)t;
            t += align_up((size_t)n * 4, 256);
            // [passes][RADIX] global histograms, [passes][ntiles][RADIX] granules, [etiles] head-count granules
            uint32_t* ghist = (uint32_t*)t;
            uint32_t* state = ghist + (size_t)RS_MAX_PASSES * RS_RADIX;
            uint32_t* tile_state = state + (size_t)passes * ntiles * RS_RADIX;
            static std::atomic<unsigned long long> tickets{1};
            SortCtl* ctl = (SortCtl*)ws;
            rs_ghist_kernel<<<dim3((unsigned)ntiles), dim3(SW_THREADS), 0, st>>>((const uint64_t*)ids, n, passes, ntiles, ghist, state, tile_state, uniq, ctl, tickets.fetch_add(1));
            // ping-pong so that the last pass lands in (keys, perm)
            const uint64_t* kin = (const uint64_t*)ids;
            const int32_t* pin = nullptr;
            for (int ps_ = 0; ps_ < passes; ++ps_) {
                const bool to_keys = ((passes - 1 - ps_) % 2) == 0;
                uint64_t* kout = to_keys ? keys : keys2;
                int32_t* pout = (ps_ == passes - 1) ? perm : (to_keys ? payA : payB);
                rs_sweep_kernel<<<dim3((unsigned)ntiles), dim3(SW_THREADS), 0, st>>>(kin, pin, n, ps_ * RS_BITS, ghist + (size_t)ps_ * RS_RADIX,
                                                                                     state + (size_t)ps_ * ntiles * RS_RADIX, kout, pout, &ctl->ctr[1 + ps_]);
                kin = kout;
                pin = pout;
            }
            rs_emit_kernel<<<dim3((unsigned)cdiv(n, EM_TILE)), dim3(RS_THREADS), 0, st>>>(keys, perm, n, tile_state, uniq, inverse, seg_offsets, num_unique_dev, ctl, 1 + passes);
            return check_launch("sort_unique");
        }
    }
    if (hipMemsetAsync(uniq, 0, (size_t)n * sizeof(int64_t), st) != hipSuccess) {
        set_last_error("sort_unique: memset failed");
        return MARIUS_ERR_HIP;
    }
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

// ---- marius_prepare_maps: the fused map launch (prepare_maps_kernel above)
static bool pm_supported(const marius_map_job& j) {
    return j.n > 0 && j.n <= (int64_t)RS_MAX_TILES * RS_TILE && j.key_bits > 0 && j.key_bits <= RS_MAX_KEY_BITS;
}

extern "C" int marius_prepare_maps_supported(const marius_map_job* jobs, int32_t num_jobs) {
    if (!jobs || num_jobs < 1 || num_jobs > PM_MAX_JOBS || kernel_env().sort_rocprim) return 0;
    for (int j = 0; j < num_jobs; ++j)
        if (!pm_supported(jobs[j])) return 0;
    return 1;
}

extern "C" int marius_prepare_maps_preferred(void) { return kernel_env().maps_fused ? 1 : 0; }

extern "C" int marius_prepare_maps(const marius_map_job* jobs, int32_t num_jobs, marius_stream_t stream) {
    MARIUS_REQUIRE(jobs && num_jobs >= 1 && num_jobs <= PM_MAX_JOBS, "prepare_maps: one or two jobs");
    PmArgs A = {};
    int width = 0;
    for (int j = 0; j < num_jobs; ++j) {
        const marius_map_job& in = jobs[j];
        MARIUS_REQUIRE(pm_supported(in), "prepare_maps: job %d: n = %ld / key_bits = %d outside the fused launch's range (marius_prepare_maps_supported)", j, (long)in.n, in.key_bits);
        MARIUS_REQUIRE(in.uniq && in.inverse && in.perm && in.seg_offsets && in.num_unique_dev && in.workspace, "prepare_maps: job %d: null output", j);
        MARIUS_REQUIRE(in.ids_in || in.ids_out, "prepare_maps: job %d: ids are neither given (ids_in) nor is there room to assemble them (ids_out)", j);
        if (!in.ids_in) {
            MARIUS_REQUIRE(in.edges && in.B > 0 && (in.edge_cols == 2 || in.edge_cols == 3), "prepare_maps: job %d: assembling ids needs the batch's edges", j);
            if (in.col >= 0) MARIUS_REQUIRE(in.col < in.edge_cols && in.n == in.B, "prepare_maps: job %d: a column job has one id per edge", j);
            else MARIUS_REQUIRE(in.n == 2 * in.B + (in.src_neg ? in.CN : 0) + (in.dst_neg ? in.CN : 0), "prepare_maps: job %d: n != 2 B + negatives", j);
        }
        MARIUS_REQUIRE(!in.edges_out || (in.edges && in.n >= 2 * in.B), "prepare_maps: job %d: edges_out needs the edges and the endpoints' positions", j);
        SortPlan p;
        if (make_plan(in.n, p) != MARIUS_OK) return MARIUS_ERR_HIP;
        MARIUS_REQUIRE(in.workspace_bytes >= p.total, "prepare_maps: job %d: workspace too small (%zu < %zu)", j, in.workspace_bytes, p.total);
        PmJob& J = A.job[j];
        J.ids_in = in.ids_in;
        J.ids = in.ids_out;
        J.edges = in.edges;
        J.src_neg = in.src_neg;
        J.dst_neg = in.dst_neg;
        J.B = in.B;
        J.CN = in.CN;
        J.n = in.n;
        J.cols = in.edge_cols;
        J.col = in.col;
        J.passes = (in.key_bits + RS_BITS - 1) / RS_BITS;
        J.ntiles = (int)cdiv(in.n, RS_TILE);
        J.etiles = (int)cdiv(in.n, PM_EM_TILE);
        J.ptiles = J.ntiles;
        J.has_plan = in.plan ? 1 : 0;
        char* ws = (char*)in.workspace;
        J.keys = (uint64_t*)(ws + p.keys_off);
        char* t = ws + p.temp_off;
        J.keys2 = (uint64_t*)t;
        t += align_up((size_t)in.n * 8, 256);
        J.payA = (int32_t*)t;
        t += align_up((size_t)in.n * 4, 256);
        J.payB = (int32_t*)t;
        t += align_up((size_t)in.n * 4, 256);
        J.ghist = (uint32_t*)t;
        J.state = J.ghist + (size_t)RS_MAX_PASSES * RS_RADIX;
        J.tile_state = J.state + (size_t)J.passes * J.ntiles * RS_RADIX;
        J.perm = in.perm;
        J.uniq = in.uniq;
        J.inverse = in.inverse;
        J.count = in.num_unique_dev;
        J.seg = in.seg_offsets;
        if (in.plan) J.plan = seg_plan_ptrs(in.plan, in.n);
        J.edges_out = in.edges_out;
        if (J.ntiles > width) width = J.ntiles;
    }
    // phases in dependency order, the jobs interleaved; every phase depends on the previous phase of its own job
    int last[PM_MAX_JOBS] = {-1, -1};
    auto add = [&](int job, int kind, int pass, int nitems) {
        PmPhase& ph = A.ph[A.nph];
        const int dep = kind == PM_HIST ? 0 : last[job];
        ph = {job, kind, pass, A.total, nitems, dep, dep >= 0 ? A.ph[dep].nitems : 0};
        last[job] = A.nph++;
        A.total += nitems;
    };
    add(0, PM_INIT, 0, 1);
    last[0] = last[1] = 0;
    for (int j = 0; j < num_jobs; ++j) add(j, PM_HIST, 0, A.job[j].ntiles);
    int maxp = 0;
    for (int j = 0; j < num_jobs; ++j) maxp = A.job[j].passes > maxp ? A.job[j].passes : maxp;
    for (int ps = 0; ps < maxp; ++ps)
        for (int j = 0; j < num_jobs; ++j)
            if (ps < A.job[j].passes) add(j, PM_SWEEP, ps, A.job[j].ntiles);
    for (int j = num_jobs - 1; j >= 0; --j) add(j, PM_EMIT, 0, A.job[j].etiles);   // (the shorter chain first: its plan phase then fills the wait for the longer one's emit)
    for (int j = num_jobs - 1; j >= 0; --j)
        if (A.job[j].has_plan || A.job[j].edges_out) add(j, PM_PLAN, 0, A.job[j].ptiles);
    MARIUS_REQUIRE(A.nph <= PM_MAX_PHASES, "prepare_maps: too many phases");
    if (const char* e = getenv("MARIUS_PM_MAXPH")) {  // debugging: only the first k phases
        const int k = atoi(e);
        if (k >= 1 && k < A.nph) {
            A.nph = k;
            A.total = A.ph[k].first;
        }
    }
    A.ctl = (PmCtl*)((char*)jobs[0].workspace + 64);
    // one workgroup per tile of the widest phase (a pass of the larger job) plus the other job's share, capped: the queue needs no particular count
    int nwg = width + (num_jobs > 1 ? A.job[1].ntiles : 0);
    if (nwg > 96) nwg = 96;
    if (kernel_env().pm_nwg > 0) nwg = kernel_env().pm_nwg;
    if (nwg < 1) nwg = 1;
    hipStream_t st = as_stream(stream);
    ProfScope ps(PROF_SORT_UNIQUE, st);
    prepare_maps_kernel<<<dim3((unsigned)nwg), dim3(SW_THREADS), 0, st>>>(A);
    return check_launch("prepare_maps");
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
    char* ws = (char*)workspace;
    uint64_t* keys = (uint64_t*)(ws + p.keys_off);
    int32_t* flags = (int32_t*)(ws + p.scan_off);
    int32_t* scan = flags + n;
    int64_t blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    {   // two launches: ranks by binary search (+ the zero fills), then the radix sort's emit kernel (heads, prefix by granule hand-off, outputs)
        const int64_t etiles = cdiv(n, EM_TILE);
        if (n <= (int64_t)RS_MAX_TILES * RS_TILE && (size_t)(etiles + 4) * 4 <= p.temp_bytes && !kernel_env().sort_rocprim) {
            uint32_t* tile_state = (uint32_t*)(ws + p.temp_off);
            merge_rank_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(ids, n, ro, num_runs, keys, perm, uniq, tile_state, etiles, &((SortCtl*)ws)->ctr[1]);
            rs_emit_kernel<<<dim3((unsigned)etiles), dim3(RS_THREADS), 0, st>>>(keys, perm, n, tile_state, uniq, inverse, seg_offsets, num_unique_dev, (SortCtl*)ws, 1);
            return check_launch("merge_unique_runs");
        }
    }
    if (hipMemsetAsync(uniq, 0, (size_t)n * sizeof(int64_t), st) != hipSuccess) {
        set_last_error("merge_unique_runs: memset failed");
        return MARIUS_ERR_HIP;
    }
    merge_rank_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(ids, n, ro, num_runs, keys, perm, nullptr, nullptr, 0, nullptr);
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
