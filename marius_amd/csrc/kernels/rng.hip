// ATen-CPU-generator-compatible sampling on the device.
//
// The reference draws negatives with torch::randint on the CPU generator (at::mt19937); "bit-exact sampled ids"
// therefore means reproducing that MT19937 stream.  mt19937_fill_kernel advances a DEVICE-resident 624-word state
// and emits raw tempered words; sample_negatives_kernel maps them to ids exactly as ATen's
// uniform_int_from_to_distribution does (draw % range, two words per draw when range >= 2^28).
//
// MT19937 regeneration is a 3-phase recurrence (new[i] needs new[i-227]); one 256-thread workgroup keeps both
// state generations in LDS and pays 3 barriers per 624 words (~45 us for the 100k words of a Freebase86m batch).
// The kernel is stream-ordered and touches one CU, so the trainer runs it on a side stream one batch ahead.
#include "common.h"

namespace marius {

constexpr int MT_N = 624;
constexpr int MT_M = 397;

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// T threads.  T = 64 (one wave, tried in round 6): the three phases of a twist are ordered by the wave's own in-order LDS queue — the barrier
// of a single-wave workgroup costs nothing — so a 624-word block is three dependent LDS round trips of 4 / 4 / 3 elements per lane instead of
// three 4-wave barriers.  Measured SLOWER (203 vs 107 us per 100,000 words): 256 threads stays the default (MARIUS_MT_THREADS selects).
template <int T>
__global__ __launch_bounds__(T) void mt19937_fill_kernel(uint32_t* __restrict__ state, uint32_t* __restrict__ out, int64_t n) {
    __shared__ uint32_t buf[2][MT_N + 8];
    constexpr int P = MT_N - MT_M;  // 227
    const int tid = threadIdx.x;
    for (int i = tid; i < MT_N; i += T) buf[0][i] = state[i];
    int idx = (int)state[MT_N];
    int cur = 0;
    __syncthreads();
    int64_t produced = 0;
    while (produced < n) {
        if (idx >= MT_N) {
            uint32_t* o = buf[cur];
            uint32_t* w = buf[cur ^ 1];
            // phase A: i in [0, 227): new[i] = old[i+397] ^ f(old[i], old[i+1])
#pragma unroll
            for (int i = tid; i < P; i += T) w[i] = mt_mix(o[i], o[i + 1], o[i + MT_M]);
            __syncthreads();
            // phase B: i in [227, 454): new[i] = new[i-227] ^ f(old[i], old[i+1])
#pragma unroll
            for (int k = tid; k < P; k += T) {
                const int i = k + P;
                w[i] = mt_mix(o[i], o[i + 1], w[k]);
            }
            __syncthreads();
            // phase C: i in [454, 624): new[i] = new[i-227] ^ f(old[i], i == 623 ? new[0] : old[i+1])
#pragma unroll
            for (int k = tid; k < MT_N - 2 * P; k += T) {
                const int i = k + 2 * P;
                const uint32_t nxt = (i == MT_N - 1) ? w[0] : o[i + 1];
                w[i] = mt_mix(o[i], nxt, w[i - P]);
            }
            __syncthreads();
            cur ^= 1;
            idx = 0;
        }
        int64_t left = n - produced;
        int take = MT_N - idx;
        if ((int64_t)take > left) take = (int)left;
        const uint32_t* s = buf[cur];
        for (int i = tid; i < take; i += T) out[produced + i] = mt_temper(s[idx + i]);
        idx += take;
        produced += take;
        // the next twist writes buf[cur^1] only, and reads buf[cur]: no hazard with the tempering reads above
    }
    __syncthreads();
    for (int i = tid; i < MT_N; i += T) state[i] = buf[cur][i];
    if (tid == 0) state[MT_N] = (uint32_t)idx;
}

__global__ __launch_bounds__(256) void sample_negatives_kernel(const uint32_t* __restrict__ raw,
                                                               const int64_t* __restrict__ edges, uint64_t B, int edge_cols,
                                                               int col, uint64_t num_nodes, int C, int N, int n_deg,
                                                               int w_uni, int w_deg, int64_t* __restrict__ out_ids,
                                                               int64_t* __restrict__ deg_pos) {
    const int64_t total = (int64_t)C * N;
    const int n_uni = N - n_deg;
    const int64_t words_per_chunk = (int64_t)n_uni * w_uni + (int64_t)n_deg * w_deg;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t / N);
        const int k = (int)(t - (int64_t)c * N);
        const uint32_t* r = raw + c * words_per_chunk;
        int64_t id;
        if (k >= n_deg) {  // uniform part, drawn first
            const uint32_t* p = r + (int64_t)(k - n_deg) * w_uni;
            uint64_t v = (w_uni == 2) ? (((uint64_t)p[0] << 32) | (uint64_t)p[1]) : (uint64_t)p[0];
            id = (int64_t)(v % num_nodes);
        } else {  // degree-based part: edge position drawn after the uniform ids of this chunk
            const uint32_t* p = r + (int64_t)n_uni * w_uni + (int64_t)k * w_deg;
            uint64_t v = (w_deg == 2) ? (((uint64_t)p[0] << 32) | (uint64_t)p[1]) : (uint64_t)p[0];
            int64_t pos = (int64_t)(v % B);
            deg_pos[(int64_t)c * n_deg + k] = pos;
            id = edges[pos * edge_cols + col];
        }
        out_ids[t] = id;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void select_edges_kernel(const T* __restrict__ in, int cols, const int64_t* __restrict__ perm,
                                                           int64_t start, int64_t B, int64_t* __restrict__ out) {
    const int64_t total = B * cols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = t / cols;
        int c = (int)(t - i * cols);
        int64_t src = perm ? perm[start + i] : (start + i);
        out[t] = (int64_t)in[src * cols + c];
    }
}

// all_ids = cat(src, dst, src_neg.flat, dst_neg.flat)   (DataLoader::edgeSample, dataloader.cpp:400-409)
__global__ __launch_bounds__(256) void assemble_ids_kernel(const int64_t* __restrict__ edges, int64_t B, int cols,
                                                           const int64_t* __restrict__ src_neg, const int64_t* __restrict__ dst_neg,
                                                           int64_t CN, int64_t* __restrict__ out) {
    const int64_t nsrc = src_neg ? CN : 0;
    const int64_t total = 2 * B + nsrc + (dst_neg ? CN : 0);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t v;
        if (t < B) v = edges[t * cols];
        else if (t < 2 * B) v = edges[(t - B) * cols + cols - 1];
        else if (t < 2 * B + nsrc) v = src_neg[t - 2 * B];
        else v = dst_neg[t - 2 * B - nsrc];
        out[t] = v;
    }
}

// edges_ = stack({src_mapping, rel, dst_mapping})   (dataloader.cpp:460-466)
__global__ __launch_bounds__(256) void remap_edges_kernel(const int64_t* __restrict__ edges, const int64_t* __restrict__ inverse,
                                                          int64_t B, int cols, int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        out[i * cols] = inverse[i];
        if (cols == 3) out[i * cols + 1] = edges[i * cols + 1];
        out[i * cols + cols - 1] = inverse[B + i];
    }
}

// deg_negative_local_filter (negative.cpp:21-39), uncompacted: row (c*n_deg + k) = (e, k) when the sampled edge position
// e = deg_pos[c][k] lies in chunk c, else (-1, -1) (ignored by the score filter; compaction keeps the reference's order).
__global__ __launch_bounds__(256) void deg_filter_kernel(const int64_t* __restrict__ deg_pos, int C, int n_deg, int64_t chunk_size,
                                                         int64_t* __restrict__ out) {
    const int64_t total = (int64_t)C * n_deg;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = t / n_deg, k = t - c * n_deg;
        const int64_t e = deg_pos[t];
        const bool hit = (e / chunk_size) == c;
        out[2 * t] = hit ? e : -1;
        out[2 * t + 1] = hit ? k : -1;
    }
}

// ---- host-side generator (same stream; used for the per-epoch randperm, which is a serial swap chain) ----
static inline uint32_t host_mix(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
// in-place regeneration in three index ranges (no wrap-around arithmetic inside the loops: they vectorise)
static void host_twist(uint32_t* p) {
    int i = 0;
    for (; i < MT_N - MT_M; i++) p[i] = host_mix(p[i], p[i + 1], p[i + MT_M]);
    for (; i < MT_N - 1; i++) p[i] = host_mix(p[i], p[i + 1], p[i + MT_M - MT_N]);
    p[MT_N - 1] = host_mix(p[MT_N - 1], p[0], p[MT_M - 1]);
}
static inline uint32_t host_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
// n tempered words into out, whole state blocks at a time
static void host_fill(uint32_t* st, uint32_t* out, int64_t n) {
    int64_t done = 0;
    while (done < n) {
        if (st[MT_N] >= (uint32_t)MT_N) {
            host_twist(st);
            st[MT_N] = 0;
        }
        const int idx = (int)st[MT_N];
        int64_t take = MT_N - idx;
        if (take > n - done) take = n - done;
        for (int64_t j = 0; j < take; j++) out[done + j] = host_temper(st[idx + j]);
        st[MT_N] = (uint32_t)(idx + take);
        done += take;
    }
}
static inline uint32_t host_next(uint32_t* st) {
    if (st[MT_N] >= (uint32_t)MT_N) {
        host_twist(st);
        st[MT_N] = 0;
    }
    return host_temper(st[st[MT_N]++]);
}

}  // namespace marius

using namespace marius;

extern "C" void marius_mt19937_seed_host(uint32_t* st, uint64_t seed) {
    st[0] = (uint32_t)(seed & 0xffffffffu);
    for (int j = 1; j < MT_N; j++) st[j] = 1812433253u * (st[j - 1] ^ (st[j - 1] >> 30)) + (uint32_t)j;
    st[MT_N] = MT_N;
}

extern "C" void marius_mt19937_fill_host(uint32_t* st, uint32_t* out, int64_t n) { host_fill(st, out, n); }

extern "C" int marius_mt19937_randperm_host(uint32_t* st, int64_t* out, int64_t n) {
    MARIUS_REQUIRE(n >= 0 && (n == 0 || out), "randperm: bad arguments");
    // ATen (TensorFactories.cpp, randperm_cpu): Fisher-Yates with 32-bit draws below 2^32 / 20 elements; above, the inside-out variant
    // with random64() = (first draw << 32) | second draw
    // Both loops are chains of dependent random accesses into `out` (80 MB at 10 M edges: every access misses the caches).  The draws do not
    // depend on the array, so they are generated a block ahead and their targets prefetched: same result, memory-level parallelism
    // instead of one miss at a time.
    constexpr int AHEAD = 512;
    int64_t zs[AHEAD];
    uint32_t w[2 * AHEAD];
    if ((uint64_t)n >= (0xffffffffull / 20ull)) {
        for (int64_t i0 = 0; i0 < n; i0 += AHEAD) {
            const int m = (int)((n - i0) < AHEAD ? (n - i0) : AHEAD);
            host_fill(st, w, 2 * (int64_t)m);
            for (int j = 0; j < m; j++) {
                const uint64_t r = ((uint64_t)w[2 * j] << 32) | (uint64_t)w[2 * j + 1];
                zs[j] = (int64_t)(r % (uint64_t)(i0 + j + 1));
                __builtin_prefetch(out + zs[j], 1);
            }
            for (int j = 0; j < m; j++) {
                const int64_t i = i0 + j, z = zs[j];
                out[i] = out[z];
                out[z] = i;
            }
        }
        return MARIUS_OK;
    }
    for (int64_t i = 0; i < n; i++) out[i] = i;
    const uint32_t n32 = (uint32_t)n;  // n < 2^32 / 20 here: 32-bit remainders
    for (int64_t i0 = 0; i0 < n - 1; i0 += AHEAD) {
        const int m = (int)((n - 1 - i0) < AHEAD ? (n - 1 - i0) : AHEAD);
        host_fill(st, w, m);
        for (int j = 0; j < m; j++) {
            const uint32_t i = (uint32_t)(i0 + j);
            zs[j] = (int64_t)(w[j] % (n32 - i)) + (int64_t)i;
            __builtin_prefetch(out + zs[j], 1);
        }
        for (int j = 0; j < m; j++) {
            const int64_t i = i0 + j;
            const int64_t sav = out[i];
            out[i] = out[zs[j]];
            out[zs[j]] = sav;
        }
    }
    return MARIUS_OK;
}

extern "C" int marius_mt19937_fill(uint32_t* state_dev, uint32_t* out_dev, int64_t n, marius_stream_t stream) {
    MARIUS_REQUIRE(state_dev && n >= 0 && (n == 0 || out_dev), "mt19937_fill: bad arguments");
    if (n == 0) return MARIUS_OK;
    ProfScope ps(PROF_MT_FILL, as_stream(stream));
    const int t = kernel_env().mt_threads;
    // default 256 threads: measured 107 / 150 / 203 us per 100,000 words with 256 / 128 / 64 threads (tools/bench_mt.py, profiles/r6_mt_threads.txt) — the
    // single-wave form saves the barriers but leaves each lane 10 elements per twist, and the twist is bound by that
    if (t == 64) mt19937_fill_kernel<64><<<dim3(1), dim3(64), 0, as_stream(stream)>>>(state_dev, out_dev, n);
    else if (t == 128) mt19937_fill_kernel<128><<<dim3(1), dim3(128), 0, as_stream(stream)>>>(state_dev, out_dev, n);
    else mt19937_fill_kernel<256><<<dim3(1), dim3(256), 0, as_stream(stream)>>>(state_dev, out_dev, n);
    return check_launch("mt19937_fill");
}

static inline int draw_words(uint64_t range) { return range >= (1ull << 28) ? 2 : 1; }

extern "C" int64_t marius_negatives_raw_words(int64_t num_nodes, int64_t B, int32_t C, int32_t N, int32_t n_deg) {
    int64_t n_uni = N - n_deg;
    return (int64_t)C * (n_uni * draw_words((uint64_t)num_nodes) + (int64_t)n_deg * draw_words((uint64_t)B));
}

extern "C" int marius_sample_negatives(const uint32_t* raw, const int64_t* edges, int64_t B, int32_t edge_cols,
                                       int32_t inverse, int64_t num_nodes, int32_t C, int32_t N, int32_t n_deg,
                                       int64_t* out_ids, int64_t* deg_pos, marius_stream_t stream) {
    MARIUS_REQUIRE(raw && out_ids && C > 0 && N > 0 && n_deg >= 0 && n_deg <= N && num_nodes > 0,
                   "sample_negatives: bad arguments");
    MARIUS_REQUIRE(n_deg == 0 || (edges && deg_pos && B > 0 && (edge_cols == 2 || edge_cols == 3)),
                   "sample_negatives: degree sampling needs edges/deg_pos");
    int64_t total = (int64_t)C * N;
    int64_t blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    int col = inverse ? 0 : (edge_cols - 1);
    sample_negatives_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(
        raw, edges, (uint64_t)B, edge_cols, col, (uint64_t)num_nodes, C, N, n_deg, draw_words((uint64_t)num_nodes),
        draw_words((uint64_t)(B > 0 ? B : 1)), out_ids, deg_pos);
    return check_launch("sample_negatives");
}

extern "C" int marius_select_edges(const void* edges_in, int32_t in_is_int64, int32_t cols, const int64_t* perm,
                                   int64_t start, int64_t B, int64_t* out, marius_stream_t stream) {
    MARIUS_REQUIRE(B >= 0 && (cols == 2 || cols == 3) && start >= 0, "select_edges: bad arguments");
    if (B == 0) return MARIUS_OK;
    MARIUS_REQUIRE(edges_in && out, "select_edges: null pointer");
    int64_t blocks = cdiv(B * cols, 256);
    if (blocks > 4096) blocks = 4096;
    if (in_is_int64)
        select_edges_kernel<int64_t><<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(
            (const int64_t*)edges_in, cols, perm, start, B, out);
    else
        select_edges_kernel<int32_t><<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(
            (const int32_t*)edges_in, cols, perm, start, B, out);
    return check_launch("select_edges");
}

extern "C" int marius_assemble_ids(const int64_t* edges, int64_t B, int32_t cols, const int64_t* src_neg, const int64_t* dst_neg,
                                   int64_t CN, int64_t* out, marius_stream_t stream) {
    MARIUS_REQUIRE(edges && out && B > 0 && (cols == 2 || cols == 3) && CN >= 0, "assemble_ids: bad arguments");
    int64_t total = 2 * B + (src_neg ? CN : 0) + (dst_neg ? CN : 0);
    int64_t blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    assemble_ids_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(edges, B, cols, src_neg, dst_neg, CN, out);
    return check_launch("assemble_ids");
}

extern "C" int marius_remap_edges(const int64_t* edges, const int64_t* inverse, int64_t B, int32_t cols, int64_t* out,
                                  marius_stream_t stream) {
    MARIUS_REQUIRE(edges && inverse && out && B > 0 && (cols == 2 || cols == 3), "remap_edges: bad arguments");
    int64_t blocks = cdiv(B, 256);
    if (blocks > 4096) blocks = 4096;
    remap_edges_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(edges, inverse, B, cols, out);
    return check_launch("remap_edges");
}

extern "C" int marius_deg_filter(const int64_t* deg_pos, int32_t C, int32_t n_deg, int64_t B, int64_t* out, marius_stream_t stream) {
    MARIUS_REQUIRE(C > 0 && n_deg >= 0 && B > 0, "deg_filter: bad arguments");
    if (n_deg == 0) return MARIUS_OK;
    MARIUS_REQUIRE(deg_pos && out, "deg_filter: null pointer");
    const int64_t chunk_size = (B + C - 1) / C;  // ceil((double)B / C), negative.cpp:28
    int64_t blocks = cdiv((int64_t)C * n_deg, 256);
    if (blocks > 1024) blocks = 1024;
    deg_filter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(deg_pos, C, n_deg, chunk_size, out);
    return check_launch("deg_filter");
}
