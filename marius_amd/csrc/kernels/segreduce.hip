// Atomic-free segmented row reduction in sorted-id order, optionally fused with the sparse Adagrad update.
//
// Replaces autograd's index_select backward (index_add_ with atomics on the device, nn/model.cpp:324) and, in the
// fused form, Batch::accumulateGradients (data/batch.cpp:62-79) + the two Storage::indexAdd calls of
// DataLoader::updateEmbeddings (data/dataloader.cpp:550-564).
//
// Input: rows[n, ld] (one gradient row per occurrence), perm/seg_offsets from marius_sort_unique (stable sort, so the
// summation order inside a segment is the input order => bit-reproducible).  The sorted positions are cut into
// chunks of SEG_R; one wave owns one chunk, so work is balanced regardless of how skewed the segment lengths are
// (a Zipf hub node or relation can own thousands of rows).  A segment fully inside a chunk is finished by that wave;
// a segment that crosses chunk boundaries leaves per-chunk partials in `carry` and is finished by the wave of the
// chunk where it starts (second launch).  Every output row has exactly one writer: no atomics.
#include "common.h"
#include "seg_plan.h"

namespace marius {

// (SEG_R = 32 sorted positions per wave: seg_plan.h)
// (tools/build_variant.py + tools/ab_variants.sh time variants of these constants on one box.  Round 4, Freebase86m step: 8 row loads in flight per
// lane 151 us for the reduce + update pair, 4 -> 144, 2 -> 141, 1 -> 141, 16 -> 167: most segments are singletons that are skipped, and the batch's
// registers cost occupancy; Adagrad rows per thread 8 -> 200 us, 4 / 2 / 1 equal once a row's pieces map one to one onto lanes; with the endpoint
// singletons updated by the edge backward (a third of the row slots skipped) 4 -> 116, 3 -> 113, 2 -> 112.5, 1 -> 119 us)
#ifndef MARIUS_SEG_BATCH
#define MARIUS_SEG_BATCH 2
#endif
#ifndef MARIUS_ADAGRAD_UNR
#define MARIUS_ADAGRAD_UNR 2
#endif
constexpr int SEG_BATCH = MARIUS_SEG_BATCH;  // row loads in flight per lane

struct SegArgs {
    const float* rows;
    int64_t rows_ld;
    const int32_t* perm;
    const int64_t* inverse;  // unique index of each input position
    const int32_t* seg_offsets;
    int64_t n;
    int d;
    float* carry;  // [nchunks][2][dpad]
    int dpad;
    int skip_singletons;  // 1: segments of length 1 are neither loaded nor written (their consumer reads the occurrence row itself)
    // optional plan (marius_segment_plan): what the kernels otherwise derive through chains of dependent index loads, precomputed off the
    // critical path (the ids are known a step before the gradients exist)
    const int4* pos_plan;    // [n]        per sorted position: {occurrence row, unique index, complete-in-chunk, singleton}
    const int4* chunk_plan;  // [nchunks]  per chunk: {owns a boundary-crossing segment, its unique index, carry slot of its first partial, last chunk}
};

struct ApplySum {
    float* out;
    int64_t out_ld;
    const int64_t* out_rows;  // optional row indirection
    __device__ __forceinline__ void finish() const {}
    template <int VEC>
    __device__ __forceinline__ void operator()(int u, int col, const float (&g)[VEC]) const {
        const int64_t r = out_rows ? out_rows[u] : (int64_t)u;
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[r * out_ld + col + e] = g[e];
    }
};

// Optional magnitude tracking (marius_lp_desc.absmax): *absmax >= every |w| the update has written.  The bound is monotone, so a racy read
// filters almost every lane out once it has settled and the atomic (non-negative floats order like their bit patterns) is rare.
__device__ __forceinline__ void track_absmax(float* absmax, float mx) {
    if (!absmax) return;
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0 && mx > *reinterpret_cast<volatile float*>(absmax)) atomicMax(reinterpret_cast<unsigned int*>(absmax), __float_as_uint(mx));
}

struct ApplyAdagrad {
    const int64_t* uniq;
    float* table;
    float* state;
    int64_t ld;
    float lr, eps;
    float* absmax;  // optional: see track_absmax (the caller reduces over the wave)
    mutable float seen = 0.f;
    template <int VEC>
    __device__ __forceinline__ void operator()(int u, int col, const float (&g)[VEC]) const {
#pragma clang fp contract(off)
        const int64_t r = uniq[u];
        float* w = table + r * ld + col;
        float* s = state + r * ld + col;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float ds = g[e] * g[e];
            const float sn = s[e] + ds;
            const float dw = -lr * (g[e] / (sqrtf(sn) + eps));
            s[e] = sn;
            const float wn = w[e] + dw;
            w[e] = wn;
            seen = fmaxf(seen, fabsf(wn));
        }
    }
    __device__ __forceinline__ void finish() const { track_absmax(absmax, seen); }
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
        v[0] = *p;
    }
}

// phase 1: one wave per chunk
template <int VEC, int NIT, class Apply>
__device__ __forceinline__ void seg_reduce_body(const SegArgs& a, const Apply& apply, int64_t block) {
    const int lane = threadIdx.x & 63;
    const int64_t chunk = block * 4 + (threadIdx.x >> 6);
    const int64_t k0 = chunk * SEG_R;
    if (k0 >= a.n) return;
    const int cnt = (int)min((int64_t)SEG_R, a.n - k0);
    const int64_t k1 = k0 + cnt;
    // lane r < cnt owns sorted position k0 + r: its occurrence row, its segment, and whether that segment lies inside the chunk
    int p = 0, u = -1, complete = 0, single = 0;
    if (lane < cnt) {
        if (a.pos_plan) {
            const int4 q = a.pos_plan[k0 + lane];
            p = q.x;
            u = q.y;
            complete = q.z;
            single = (a.skip_singletons && q.w) || q.w == 2;  // 2: a padding slot (negative id), dead in every form
        } else {
            p = a.perm[k0 + lane];
            u = (int)a.inverse[p];
            const int s0 = a.seg_offsets[u], s1 = a.seg_offsets[u + 1];
            complete = (s0 >= k0) && (s1 <= k1);
            single = a.skip_singletons && (s1 - s0 == 1);
        }
    }
    const int u_first = __shfl(u, 0, 64);
    float acc[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[it][e] = 0.f;
    int cur = u_first, cur_complete = __shfl(complete, 0, 64), cur_single = __shfl(single, 0, 64);

    auto flush = [&](int useg, int is_complete, int is_single) {
        if (is_single) return;  // nothing was accumulated and nothing is stored
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (lane + it * 64) * VEC;
            if (col < a.d) {
                if (is_complete) {
                    apply.template operator()<VEC>(useg, col, acc[it]);
                } else {
                    float* c = a.carry + ((chunk * 2) + (useg == u_first ? 0 : 1)) * a.dpad + col;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) c[e] = acc[it][e];
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[it][e] = 0.f;
        }
    };

    for (int b0 = 0; b0 < cnt; b0 += SEG_BATCH) {
        float v[SEG_BATCH][NIT][VEC];
        int ub[SEG_BATCH], cb[SEG_BATCH], sb[SEG_BATCH];
#pragma unroll
        for (int j = 0; j < SEG_BATCH; ++j) {
            const int r = b0 + j;
            const int rr = r < cnt ? r : 0;
            const int pr = __shfl(p, rr, 64);
            ub[j] = (r < cnt) ? __shfl(u, rr, 64) : -1;
            cb[j] = __shfl(complete, rr, 64);
            sb[j] = __shfl(single, rr, 64);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int col = (lane + it * 64) * VEC;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[j][it][e] = 0.f;
                if (r < cnt && col < a.d && !sb[j]) load_vec<VEC>(a.rows + (int64_t)pr * a.rows_ld + col, v[j][it]);
            }
        }
#pragma unroll
        for (int j = 0; j < SEG_BATCH; ++j) {
            if (ub[j] < 0) break;
            if (ub[j] != cur) {
                flush(cur, cur_complete, cur_single);
                cur = ub[j];
                cur_complete = cb[j];
                cur_single = sb[j];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[it][e] += v[j][it][e];
        }
    }
    flush(cur, cur_complete, cur_single);
}

template <int VEC, int NIT, class Apply>
__global__ __launch_bounds__(256) void seg_reduce_kernel(SegArgs a, Apply apply) {
    seg_reduce_body<VEC, NIT, Apply>(a, apply, (int64_t)blockIdx.x);
}

// phase 2: the chunk in which a boundary-crossing segment STARTS finishes it from the carries
template <int VEC, int NIT, class Apply>
__device__ __forceinline__ void seg_fixup_body(const SegArgs& a, const Apply& apply, int64_t block) {
    const int lane = threadIdx.x & 63;
    const int64_t chunk = block * 4 + (threadIdx.x >> 6);
    const int64_t k0 = chunk * SEG_R;
    if (k0 >= a.n) return;
    const int64_t k1 = min(k0 + SEG_R, a.n);
    int u_first, u_last;
    int64_t last_chunk;
    if (a.chunk_plan) {
        const int4 q = a.chunk_plan[chunk];
        if (!q.x) return;  // not the owner of a crossing segment
        u_last = q.y;
        u_first = q.z ? -1 : q.y;  // only (u_last == u_first) is used below: slot 0 iff the segment opens the chunk
        last_chunk = q.w;
    } else {
        u_first = (int)a.inverse[a.perm[k0]];
        u_last = (int)a.inverse[a.perm[k1 - 1]];
        const int64_t s0 = a.seg_offsets[u_last], s1 = a.seg_offsets[u_last + 1];
        if (!(s0 >= k0 && s1 > k1)) return;  // not the owner of a crossing segment
        last_chunk = (s1 - 1) / SEG_R;
    }
    float acc[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (lane + it * 64) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[it][e] = 0.f;
        if (col < a.d) load_vec<VEC>(a.carry + ((chunk * 2) + (u_last == u_first ? 0 : 1)) * a.dpad + col, acc[it]);
    }
    // a hub segment can span hundreds of chunks: keep FIX_U carry loads in flight per lane instead of one dependent load per chunk
    constexpr int FIX_U = 8;
    int64_t ch = chunk + 1;
    for (; ch + FIX_U - 1 <= last_chunk; ch += FIX_U) {
        float t[FIX_U][NIT][VEC];
#pragma unroll
        for (int j = 0; j < FIX_U; ++j)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int col = (lane + it * 64) * VEC;
                if (col < a.d) load_vec<VEC>(a.carry + ((ch + j) * 2) * a.dpad + col, t[j][it]);
            }
#pragma unroll
        for (int j = 0; j < FIX_U; ++j)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int col = (lane + it * 64) * VEC;
                if (col < a.d) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[it][e] += t[j][it][e];
                }
            }
    }
    for (; ch <= last_chunk; ++ch) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (lane + it * 64) * VEC;
            if (col < a.d) {
                float t[VEC];
                load_vec<VEC>(a.carry + (ch * 2) * a.dpad + col, t);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[it][e] += t[e];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (lane + it * 64) * VEC;
        if (col < a.d) apply.template operator()<VEC>(u_last, col, acc[it]);
    }
    apply.finish();
}

template <int VEC, int NIT, class Apply>
__global__ __launch_bounds__(256) void seg_fixup_kernel(SegArgs a, Apply apply) {
    seg_fixup_body<VEC, NIT, Apply>(a, apply, (int64_t)blockIdx.x);
}

// ---- plan: everything the three kernels derive from perm / inverse / seg_offsets alone (seg_plan.h: seg_plan_position)
__global__ __launch_bounds__(256) void seg_plan_kernel(const int32_t* __restrict__ perm, const int64_t* __restrict__ inverse, const int32_t* __restrict__ seg_offsets,
                                                       const int64_t* __restrict__ uniq, int64_t n, SegPlanPtrs P) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t U = inverse[perm[n - 1]] + 1;
    seg_plan_position(k, n, U, perm, inverse, seg_offsets, uniq, P);
}

static inline int dpad_of(int d) { return (d + 3) / 4 * 4; }
// threads along a row in adagrad_rows_body: every VEC-float piece of a row gets its own lane when the row has at most 64 pieces
// (MARIUS_ADAGRAD_TX_EXACT; d = 100: 25 lanes per row, 10 rows per pass and 6 idle threads — rounded up to a power of two it is 32 lanes with 7 idle in each row)
// (measured: 144 -> 137 us for the reduce + update pair at d = 100)
#ifndef MARIUS_ADAGRAD_TX_EXACT
#define MARIUS_ADAGRAD_TX_EXACT 1
#endif
static inline int adagrad_tx(int vpr) {
    if (MARIUS_ADAGRAD_TX_EXACT && vpr >= 1 && vpr <= 64) return vpr;
    int tx = 1;
    while (tx < vpr && tx < 64) tx <<= 1;
    return tx;
}

// Row-parallel sparse Adagrad over the per-unique-row gradients g[U, dpad] (U = inverse[perm[n-1]] + 1 read on the device):
//   ds = g*g; s = state[id] + ds; table[id] += -lr * (g / (sqrt(s) + eps)); state[id] = s        (batch.cpp:67-69 op order)
struct AdagradRowsArgs {
    const float* g;
    int64_t g_ld;
    const int32_t* perm;
    const int64_t* inverse;
    int64_t n;
    const int64_t* uniq;
    float* table;
    float* state;
    int64_t ld;
    int vpr, tx_n;  // VEC-float pieces per row; threads along a row (power of two <= 64): 256 / tx_n rows per block pass
    float lr, eps;
    const float* occ;
    int64_t occ_ld;
    const int32_t* seg_offsets;
    const int4* row_plan;
    int skip_crossing;  // 1: rows whose segment crosses a chunk boundary are updated by the fix-up workgroups of the same launch
    float* absmax;      // optional: track_absmax
    int fused_below;    // > 0 (planned form): singleton rows whose occurrence row is < fused_below were updated by the rows' producer (marius_segment_update.fused_below)
};

template <int VEC>
__device__ __forceinline__ void adagrad_rows_body(const AdagradRowsArgs& A, int64_t block) {
#pragma clang fp contract(off)
    const float* __restrict__ g = A.g;
    const int64_t g_ld = A.g_ld, n = A.n, ld = A.ld, occ_ld = A.occ_ld;
    const int32_t* __restrict__ perm = A.perm;
    const int64_t* __restrict__ inverse = A.inverse;
    const int64_t* __restrict__ uniq = A.uniq;
    float* __restrict__ table = A.table;
    float* __restrict__ state = A.state;
    const int vpr = A.vpr;
    const float lr = A.lr, eps = A.eps;
    const float* __restrict__ occ = A.occ;
    const int32_t* __restrict__ seg_offsets = A.seg_offsets;
    const int4* __restrict__ row_plan = A.row_plan;
    const int64_t U = row_plan ? n : inverse[perm[n - 1]] + 1;  // planned: rows past U carry id -1
    const int TX = A.tx_n, TY = 256 / TX;
    const int ty = threadIdx.x / TX, tx = threadIdx.x - ty * TX;  // TX need not be a power of two: the 256 - TX TY threads left over idle
    constexpr int UNR = MARIUS_ADAGRAD_UNR;
    int64_t rows[UNR], ids[UNR];
    const float* grow[UNR];  // the row's gradient: the reduced sum, or (segment of one occurrence) that occurrence's row itself
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
        rows[k] = (block * UNR + k) * TY + ty;
        grow[k] = g + rows[k] * g_ld;
        if (ty >= TY) {
            ids[k] = -1;
        } else if (row_plan) {
            ids[k] = -1;
            if (rows[k] < U) {
                const int4 q = row_plan[rows[k]];
                ids[k] = ((int64_t)q.y << 32) | (int64_t)(uint32_t)q.x;
                if (occ && q.z >= 0) grow[k] = occ + (int64_t)q.z * occ_ld;
                if (A.skip_crossing && q.w) ids[k] = -1;
                if (q.z >= 0 && q.z < A.fused_below) ids[k] = -1;
            }
        } else {
            ids[k] = rows[k] < U ? uniq[rows[k]] : -1;
            if (ids[k] >= 0 && occ) {
                const int s0 = seg_offsets[rows[k]], s1 = seg_offsets[rows[k] + 1];
                if (s1 - s0 == 1) grow[k] = occ + (int64_t)perm[s0] * occ_ld;
            }
        }
    }
    float seen = 0.f;
    for (int c = tx; c < vpr; c += TX) {
        float gv[UNR][VEC], wv[UNR][VEC], sv[UNR][VEC];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (ids[k] >= 0) {
                load_vec<VEC>(grow[k] + c * VEC, gv[k]);
                load_vec<VEC>(table + ids[k] * ld + c * VEC, wv[k]);
                load_vec<VEC>(state + ids[k] * ld + c * VEC, sv[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (ids[k] >= 0) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float ds = gv[k][e] * gv[k][e];
                    const float sn = sv[k][e] + ds;
                    const float dw = -lr * (gv[k][e] / (sqrtf(sn) + eps));
                    sv[k][e] = sn;
                    wv[k][e] = wv[k][e] + dw;
                    seen = fmaxf(seen, fabsf(wv[k][e]));
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    state[ids[k] * ld + c * VEC + e] = sv[k][e];
                    table[ids[k] * ld + c * VEC + e] = wv[k][e];
                }
            }
        }
    }
    track_absmax(A.absmax, seen);
}

// Reduce-only job of a group (marius_segment_update.sum_out): the rows that occur once are COPIED — occurrence row -> output row — by the row
// blocks of the second launch (the reduce launch skipped them, exactly as it does for the Adagrad jobs); every other row was written by the reduce
// launch or is finished by the fix-up blocks.  Same thread layout as adagrad_rows_body.
template <int VEC>
__device__ __forceinline__ void copy_single_rows_body(const AdagradRowsArgs& A, const ApplySum& S, int64_t block) {
    const int TX = A.tx_n, TY = 256 / TX;
    const int ty = threadIdx.x / TX, tx = threadIdx.x - ty * TX;
    constexpr int UNR = MARIUS_ADAGRAD_UNR;
    const float* src[UNR];
    float* dst[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
        const int64_t row = (block * UNR + k) * TY + ty;
        src[k] = nullptr;
        dst[k] = nullptr;
        if (ty < TY && row < A.n) {
            const int4 q = A.row_plan[row];
            if (q.z >= 0 && !(q.x == -1 && q.y == -1)) {  // a real row (not past the unique count, not padding) with a single occurrence
                src[k] = A.occ + (int64_t)q.z * A.occ_ld;
                dst[k] = S.out + (S.out_rows ? S.out_rows[row] : row) * S.out_ld;
            }
        }
    }
    for (int c = tx; c < A.vpr; c += TX) {
        float v[UNR][VEC];
#pragma unroll
        for (int k = 0; k < UNR; ++k)
            if (src[k]) load_vec<VEC>(src[k] + c * VEC, v[k]);
#pragma unroll
        for (int k = 0; k < UNR; ++k)
            if (src[k]) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) dst[k][c * VEC + e] = v[k][e];
            }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void adagrad_unique_rows_kernel(AdagradRowsArgs A) {
    adagrad_rows_body<VEC>(A, (int64_t)blockIdx.x);
}

// One launch, two kinds of workgroups (planned form): the first `nfix` finish the segments that cross chunk boundaries from their carries
// and apply Adagrad to those rows themselves; the rest update every other row.  The two sets of rows are disjoint and both only need the
// reduce launch before them, so the fix-up no longer stands between the reduction and the update.
template <int VEC, int NIT>
__global__ __launch_bounds__(256) void adagrad_with_fixup_kernel(SegArgs sa, ApplyAdagrad ap, AdagradRowsArgs A, int nfix) {
    if ((int)blockIdx.x < nfix)
        seg_fixup_body<VEC, NIT, ApplyAdagrad>(sa, ap, (int64_t)blockIdx.x);
    else
        adagrad_rows_body<VEC>(A, (int64_t)blockIdx.x - nfix);
}

// Several independent tables (the node table and the relation tables of one training step) in ONE pair of launches: workgroup ranges
// select the job.  Per job the work and its order are those of the single-table launches above, so the results are theirs bit for bit; what
// goes away are four launches, the side stream they ran on and its fork / join events (13 + 18 us of the step's tail, trace in DESIGN.md 5).
constexpr int SEG_GROUP_MAX = 4;
struct SegJob {
    SegArgs sa;
    ApplySum sum;
    ApplyAdagrad ada;
    AdagradRowsArgs A;
    unsigned red0, upd0, nfix;  // first workgroup of the job in the reduce / update grid; fix-up workgroups at the head of its update range
    int sum_only;               // 1: a reduce-only job (marius_segment_update.sum_out): `sum` is the destination itself, the row blocks copy the singletons
};
struct SegGroup {
    SegJob job[SEG_GROUP_MAX];
    int njobs;
};
// (the job is picked by static index under uniform branches: indexing the kernel argument with a run-time value makes the compiler copy
// the whole group into scratch — the grouped launches then took 0.8 ms instead of 0.13)
template <int VEC, int NIT>
__global__ __launch_bounds__(256) void seg_reduce_group_kernel(SegGroup G) {
    const unsigned b = blockIdx.x;
#define SEG_JOB(J)                                                                                          \
    if (J == G.njobs - 1 || b < G.job[J + 1 < SEG_GROUP_MAX ? J + 1 : J].red0) {                              \
        seg_reduce_body<VEC, NIT, ApplySum>(G.job[J].sa, G.job[J].sum, (int64_t)(b - G.job[J].red0));        \
        return;                                                                                             \
    }
    SEG_JOB(0) SEG_JOB(1) SEG_JOB(2) SEG_JOB(3)
#undef SEG_JOB
}
template <int VEC, int NIT>
__global__ __launch_bounds__(256) void adagrad_with_fixup_group_kernel(SegGroup G) {
    const unsigned b0 = blockIdx.x;
#define SEG_JOB(J)                                                                                          \
    if (J == G.njobs - 1 || b0 < G.job[J + 1 < SEG_GROUP_MAX ? J + 1 : J].upd0) {                             \
        const unsigned b = b0 - G.job[J].upd0;                                                              \
        if (G.job[J].sum_only) {                                                                            \
            if (b < G.job[J].nfix)                                                                          \
                seg_fixup_body<VEC, NIT, ApplySum>(G.job[J].sa, G.job[J].sum, (int64_t)b);                   \
            else                                                                                            \
                copy_single_rows_body<VEC>(G.job[J].A, G.job[J].sum, (int64_t)(b - G.job[J].nfix));          \
        } else if (b < G.job[J].nfix)                                                                       \
            seg_fixup_body<VEC, NIT, ApplyAdagrad>(G.job[J].sa, G.job[J].ada, (int64_t)b);                   \
        else                                                                                                \
            adagrad_rows_body<VEC>(G.job[J].A, (int64_t)(b - G.job[J].nfix));                                \
        return;                                                                                             \
    }
    SEG_JOB(0) SEG_JOB(1) SEG_JOB(2) SEG_JOB(3)
#undef SEG_JOB
}

template <class Apply>
static int launch_seg(const SegArgs& a, const Apply& apply, int vec, hipStream_t st, bool fixup = true) {
    const int64_t nchunks = cdiv(a.n, SEG_R);
    dim3 grid((unsigned)cdiv(nchunks, 4)), block(256);
    const int per = cdiv(a.d, 64 * vec);  // column iterations per lane
#define SEG_LAUNCH(V, N)                                             \
    do {                                                             \
        seg_reduce_kernel<V, N, Apply><<<grid, block, 0, st>>>(a, apply); \
        if (fixup) seg_fixup_kernel<V, N, Apply><<<grid, block, 0, st>>>(a, apply);  \
    } while (0)
    if (vec == 4) {
        if (per <= 1) SEG_LAUNCH(4, 1);
        else if (per <= 2) SEG_LAUNCH(4, 2);
        else { set_last_error("segment reduce: d=%d too large", a.d); return MARIUS_ERR_UNSUPPORTED; }
    } else if (vec == 2) {
        if (per <= 1) SEG_LAUNCH(2, 1);
        else if (per <= 2) SEG_LAUNCH(2, 2);
        else if (per <= 4) SEG_LAUNCH(2, 4);
        else { set_last_error("segment reduce: d=%d too large", a.d); return MARIUS_ERR_UNSUPPORTED; }
    } else {
        if (per <= 1) SEG_LAUNCH(1, 1);
        else if (per <= 2) SEG_LAUNCH(1, 2);
        else if (per <= 4) SEG_LAUNCH(1, 4);
        else if (per <= 8) SEG_LAUNCH(1, 8);
        else { set_last_error("segment reduce: d=%d too large", a.d); return MARIUS_ERR_UNSUPPORTED; }
    }
#undef SEG_LAUNCH
    return check_launch("segment_reduce");
}

static int fill_args(SegArgs& a, const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                     const int32_t* seg_offsets, int64_t n, int d, void* carry) {
    MARIUS_REQUIRE(n >= 0 && d > 0 && d <= 512 && rows_ld >= d, "segment reduce: bad sizes n=%ld d=%d", (long)n, d);
    MARIUS_REQUIRE(n == 0 || (rows && perm && inverse && seg_offsets && carry), "segment reduce: null pointer");
    a.rows = rows;
    a.rows_ld = rows_ld;
    a.perm = perm;
    a.inverse = inverse;
    a.seg_offsets = seg_offsets;
    a.n = n;
    a.d = d;
    a.carry = (float*)carry;
    a.dpad = dpad_of(d);
    a.skip_singletons = 0;
    a.pos_plan = nullptr;
    a.chunk_plan = nullptr;
    return MARIUS_OK;
}

}  // namespace marius

using namespace marius;

static inline size_t carry_only_bytes(int64_t n, int d) { return ((size_t)cdiv(n > 0 ? n : 1, SEG_R) * 2 * dpad_of(d) * sizeof(float) + 255) / 256 * 256; }

extern "C" size_t marius_segment_carry_bytes(int64_t n, int32_t d) {
    // chunk carries + (for the fused Adagrad form) the per-unique-row gradient scratch [n, dpad]
    return carry_only_bytes(n, d) + (size_t)(n > 0 ? n : 1) * dpad_of(d) * sizeof(float) + 256;
}

static int segment_sum_rows_impl(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse, const int32_t* seg_offsets, int64_t n, int32_t d,
                                 const int64_t* out_rows, float* out, int64_t out_ld, void* carry, const void* plan, marius_stream_t stream);

extern "C" int marius_segment_sum_rows(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                       const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* out_rows, float* out,
                                       int64_t out_ld, void* carry, marius_stream_t stream) {
    return segment_sum_rows_impl(rows, rows_ld, perm, inverse, seg_offsets, n, d, out_rows, out, out_ld, carry, nullptr, stream);
}

extern "C" int marius_segment_sum_rows_planned(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                               const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* out_rows, float* out,
                                               int64_t out_ld, void* carry, const void* plan, marius_stream_t stream) {
    MARIUS_REQUIRE(plan || n == 0, "segment_sum_rows_planned: null plan");
    return segment_sum_rows_impl(rows, rows_ld, perm, inverse, seg_offsets, n, d, out_rows, out, out_ld, carry, plan, stream);
}

// [pos_plan | chunk_plan | row_plan | occ_single]
extern "C" size_t marius_segment_plan_bytes(int64_t n) { return 2 * plan_pos_bytes(n) + plan_chunk_bytes(n) + plan_occ_bytes(n); }
extern "C" const uint8_t* marius_segment_plan_occ_single(const void* plan, int64_t n) {
    return plan ? (const uint8_t*)plan + 2 * plan_pos_bytes(n) + plan_chunk_bytes(n) : nullptr;
}

extern "C" int marius_segment_plan(const int32_t* perm, const int64_t* inverse, const int32_t* seg_offsets, const int64_t* uniq_ids, int64_t n, void* plan,
                                   marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && (n == 0 || (perm && inverse && seg_offsets && uniq_ids && plan)), "segment_plan: bad arguments");
    if (n == 0) return MARIUS_OK;
    seg_plan_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream)>>>(perm, inverse, seg_offsets, uniq_ids, n, seg_plan_ptrs(plan, n));
    return check_launch("segment_plan");
}

static int segment_sum_rows_impl(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse, const int32_t* seg_offsets, int64_t n, int32_t d,
                                 const int64_t* out_rows, float* out, int64_t out_ld, void* carry, const void* plan, marius_stream_t stream) {
    SegArgs a;
    int rc = fill_args(a, rows, rows_ld, perm, inverse, seg_offsets, n, d, carry);
    if (rc) return rc;
    if (n == 0) return MARIUS_OK;
    MARIUS_REQUIRE(out && out_ld >= d, "segment_sum_rows: bad output");
    int vec = row_vec_width(rows, rows_ld, d);
    int v2 = row_vec_width(out, out_ld, d);
    vec = vec < v2 ? vec : v2;
    if (plan) {
        const char* pp = (const char*)plan;
        a.pos_plan = (const int4*)pp;
        a.chunk_plan = (const int4*)(pp + plan_pos_bytes(n));
    }
    ApplySum ap{out, out_ld, out_rows};
    return launch_seg(a, ap, vec, as_stream(stream));
}


static int segment_adagrad_scatter_impl(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse, const int32_t* seg_offsets, int64_t n,
                                        int32_t d, const int64_t* uniq_ids, float* table, float* state, int64_t table_ld, float lr, float eps, void* carry,
                                        const void* plan, float* absmax, marius_stream_t stream, int64_t fused_below = 0);

extern "C" int marius_segment_adagrad_scatter(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                              const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* uniq_ids,
                                              float* table, float* state, int64_t table_ld, float lr, float eps, void* carry,
                                              marius_stream_t stream) {
    return segment_adagrad_scatter_impl(rows, rows_ld, perm, inverse, seg_offsets, n, d, uniq_ids, table, state, table_ld, lr, eps, carry, nullptr, nullptr, stream);
}

extern "C" int marius_segment_adagrad_scatter_planned(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                                      const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* uniq_ids, float* table, float* state,
                                                      int64_t table_ld, float lr, float eps, void* carry, const void* plan, marius_stream_t stream) {
    MARIUS_REQUIRE(plan || n == 0, "segment_adagrad_scatter_planned: null plan");
    return segment_adagrad_scatter_impl(rows, rows_ld, perm, inverse, seg_offsets, n, d, uniq_ids, table, state, table_ld, lr, eps, carry, plan, nullptr, stream);
}

extern "C" int marius_segment_adagrad_scatter_tracked(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                                      const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* uniq_ids, float* table, float* state,
                                                      int64_t table_ld, float lr, float eps, void* carry, const void* plan, float* absmax, marius_stream_t stream) {
    return segment_adagrad_scatter_impl(rows, rows_ld, perm, inverse, seg_offsets, n, d, uniq_ids, table, state, table_ld, lr, eps, carry, plan, absmax, stream);
}

// max |x| over a [rows, d] table (row pitch ld), max'ed into *absmax (the caller zero-initialises it, or keeps a running bound): the starting
// point of marius_lp_desc.absmax.  HBM-bound: contiguous rows (ld == d, 16-B aligned) are read as one flat float4 stream, four loads in flight
// per lane; rows_dev (optional) = device count that replaces `rows`.
template <bool FLAT>
__global__ __launch_bounds__(256) void table_absmax_kernel(const float* __restrict__ t, int64_t rows, const int64_t* __restrict__ rows_dev, int64_t ld, int d,
                                                           float* __restrict__ absmax) {
    if (rows_dev) rows = min(rows, *rows_dev);
    float mx = 0.f;
    const int64_t n = rows * d, stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (FLAT) {
        const int64_t n4 = n >> 2;
        const float4* t4 = reinterpret_cast<const float4*>(t);
        int64_t i = i0;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float4 a = t4[i], b = t4[i + stride], c = t4[i + 2 * stride], e = t4[i + 3 * stride];
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))), fmaxf(fmaxf(fabsf(e.x), fabsf(e.y)), fmaxf(fabsf(e.z), fabsf(e.w)))));
        }
        for (; i < n4; i += stride) {
            const float4 a = t4[i];
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
        for (int64_t k = (n4 << 2) + i0; k < n; k += stride) mx = fmaxf(mx, fabsf(t[k]));
    } else {
        for (int64_t i = i0; i < n; i += stride) {
            const int64_t r = i / d;
            mx = fmaxf(mx, fabsf(t[r * ld + (i - r * d)]));
        }
    }
    track_absmax(absmax, mx);
}
static int table_absmax_launch(const float* table, int64_t rows, const int64_t* rows_dev, int64_t ld, int32_t d, float* absmax, marius_stream_t stream) {
    MARIUS_REQUIRE(rows >= 0 && d > 0 && ld >= d && absmax && (rows == 0 || table), "table_absmax: bad arguments");
    if (rows == 0) return MARIUS_OK;
    const bool flat = ld == d && (reinterpret_cast<uintptr_t>(table) & 15) == 0;
    int64_t blocks = cdiv(rows * d, 256 * 16);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (flat) table_absmax_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(table, rows, rows_dev, ld, d, absmax);
    else table_absmax_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(table, rows, rows_dev, ld, d, absmax);
    return check_launch("table_absmax");
}
extern "C" int marius_table_absmax(const float* table, int64_t rows, int64_t ld, int32_t d, float* absmax, marius_stream_t stream) {
    return table_absmax_launch(table, rows, nullptr, ld, d, absmax, stream);
}
extern "C" int marius_table_absmax_counted(const float* table, int64_t capacity, const int64_t* num_rows_dev, int64_t ld, int32_t d, float* absmax,
                                           marius_stream_t stream) {
    MARIUS_REQUIRE(num_rows_dev, "table_absmax_counted: the device row count is required");
    return table_absmax_launch(table, capacity, num_rows_dev, ld, d, absmax, stream);
}

static int segment_adagrad_scatter_impl(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse, const int32_t* seg_offsets, int64_t n,
                                        int32_t d, const int64_t* uniq_ids, float* table, float* state, int64_t table_ld, float lr, float eps, void* carry,
                                        const void* plan, float* absmax, marius_stream_t stream, int64_t fused_below) {
    MARIUS_REQUIRE(fused_below <= 0 || plan, "segment_adagrad_scatter: fused_below needs a plan");
    SegArgs a;
    int rc = fill_args(a, rows, rows_ld, perm, inverse, seg_offsets, n, d, carry);
    if (rc) return rc;
    if (n == 0) return MARIUS_OK;
    MARIUS_REQUIRE(uniq_ids && table && state && table_ld >= d, "segment_adagrad_scatter: bad table arguments");
    int vec = row_vec_width(rows, rows_ld, d);
    int v2 = row_vec_width(table, table_ld, d), v3 = row_vec_width(state, table_ld, d);
    vec = vec < v2 ? vec : v2;
    vec = vec < v3 ? vec : v3;
    hipStream_t st = as_stream(stream);
    ProfScope ps(PROF_SEG_ADAGRAD, st);
    // (1) per-unique-row gradient sums into scratch (stores only: nothing in the reduction waits on the tables)
    float* gsum = (float*)((char*)carry + carry_only_bytes(n, d));
    const int64_t g_ld = dpad_of(d);
    int vsum = row_vec_width(rows, rows_ld, d);
    ApplySum ap{gsum, g_ld, nullptr};
    // most rows of a batch occur once (uniform negatives over a large table): those segments skip the reduction entirely and the
    // update kernel reads their gradient straight from the occurrence row — no 80 MB round trip through gsum
    const bool skip = vec <= vsum;
    a.skip_singletons = skip ? 1 : 0;
    const int4* row_plan = nullptr;
    if (plan) {
        const char* pp = (const char*)plan;
        a.pos_plan = (const int4*)pp;
        a.chunk_plan = (const int4*)(pp + plan_pos_bytes(n));
        row_plan = (const int4*)(pp + plan_pos_bytes(n) + plan_chunk_bytes(n));
    }
    const int vpr = d / vec;
    const int tx = adagrad_tx(vpr);
    const int ty = 256 / tx;
    const unsigned row_blocks = (unsigned)cdiv(n, (int64_t)ty * MARIUS_ADAGRAD_UNR);
    const float* occ = skip ? rows : nullptr;
    MARIUS_REQUIRE(fused_below <= 0 || skip, "segment_adagrad_scatter: fused_below needs the singleton-skipping form (rows at least as aligned as the tables)");
    AdagradRowsArgs A{gsum, g_ld, perm, inverse, n, uniq_ids, table, state, table_ld, vpr, tx, lr, eps, occ, rows_ld, seg_offsets, row_plan, 0, absmax, (int)(fused_below > 0 ? fused_below : 0)};
    const int per = cdiv(d, 64 * vec);
    // (MARIUS_SEG_FUSED_FIXUP=0: the fix-up as its own launch between reduction and update — A/B runs)
    if (plan && vec == vsum && vec == 4 && per <= 2 && !kernel_env().seg_fused_fixup_off) {
        rc = launch_seg(a, ap, vsum, st, /*fixup=*/false);
        if (rc) return rc;
        A.skip_crossing = 1;
        ApplyAdagrad aa{uniq_ids, table, state, table_ld, lr, eps, absmax};
        const unsigned nfix = (unsigned)cdiv(cdiv(n, SEG_R), 4);
        dim3 grid(nfix + row_blocks), block(256);
        if (per <= 1) adagrad_with_fixup_kernel<4, 1><<<grid, block, 0, st>>>(a, aa, A, (int)nfix);
        else adagrad_with_fixup_kernel<4, 2><<<grid, block, 0, st>>>(a, aa, A, (int)nfix);
        return check_launch("segment_adagrad_scatter");
    }
    rc = launch_seg(a, ap, vsum, st);
    if (rc) return rc;
    // (2) row-parallel Adagrad + scatter (ids ascending and unique: race-free, fully pipelined loads)
    dim3 block(256), grid(row_blocks);
    if (vec == 4)
        adagrad_unique_rows_kernel<4><<<grid, block, 0, st>>>(A);
    else if (vec == 2)
        adagrad_unique_rows_kernel<2><<<grid, block, 0, st>>>(A);
    else
        adagrad_unique_rows_kernel<1><<<grid, block, 0, st>>>(A);
    return check_launch("segment_adagrad_scatter");
}

extern "C" int marius_segment_adagrad_scatter_group(const marius_segment_update* jobs, int32_t njobs, marius_stream_t stream) {
    MARIUS_REQUIRE(jobs && njobs >= 1, "segment_adagrad_scatter_group: no jobs");
    hipStream_t st = as_stream(stream);
    bool grouped = njobs <= SEG_GROUP_MAX && !kernel_env().seg_group_off;  // (MARIUS_SEG_GROUP=0: one launch pair per table — A/B runs)
    int per0 = 0;
    for (int j = 0; j < njobs && grouped; ++j) {  // the single-launch planned form's conditions (segment_adagrad_scatter_impl), for every job
        const marius_segment_update& u = jobs[j];
        const bool so = u.sum_out != nullptr;
        if (!(u.n > 0 && u.plan && u.d > 0 && u.d <= 512 && u.rows && u.perm && u.inverse && u.seg_offsets && u.carry && u.rows_ld >= u.d &&
              (so ? u.sum_out_ld >= u.d : (u.uniq_ids && u.table && u.state && u.table_ld >= u.d)))) { grouped = false; break; }
        const int v = row_vec_width(u.rows, u.rows_ld, u.d);
        const int v2 = so ? row_vec_width(u.sum_out, u.sum_out_ld, u.d) : row_vec_width(u.table, u.table_ld, u.d), v3 = so ? 4 : row_vec_width(u.state, u.table_ld, u.d);
        const int per = cdiv(u.d, 256);
        if (v != 4 || v2 != 4 || v3 != 4 || per > 2 || (j > 0 && per != per0)) grouped = false;
        for (int i = 0; i < j; ++i)  // jobs run side by side: a shared scratch or a shared table would race
            if (jobs[i].carry == u.carry || (!so && !jobs[i].sum_out && (jobs[i].table == u.table || jobs[i].state == u.state)) || (so && jobs[i].sum_out == u.sum_out))
                grouped = false;
        per0 = per;
    }
    if (!grouped) {
        for (int j = 0; j < njobs; ++j) {
            const marius_segment_update& u = jobs[j];
            int rc = u.sum_out ? segment_sum_rows_impl(u.rows, u.rows_ld, u.perm, u.inverse, u.seg_offsets, u.n, u.d, u.sum_out_rows, u.sum_out, u.sum_out_ld, u.carry, u.plan, stream)
                               : segment_adagrad_scatter_impl(u.rows, u.rows_ld, u.perm, u.inverse, u.seg_offsets, u.n, u.d, u.uniq_ids, u.table, u.state, u.table_ld, u.lr, u.eps,
                                                              u.carry, u.plan, u.absmax, stream, u.fused_below);
            if (rc) return rc;
        }
        return MARIUS_OK;
    }
    ProfScope ps(PROF_SEG_ADAGRAD, st);
    SegGroup G;
    G.njobs = njobs;
    unsigned red = 0, upd = 0;
    for (int j = 0; j < njobs; ++j) {
        const marius_segment_update& u = jobs[j];
        SegJob& J = G.job[j];
        int rc = fill_args(J.sa, u.rows, u.rows_ld, u.perm, u.inverse, u.seg_offsets, u.n, u.d, u.carry);
        if (rc) return rc;
        float* gsum = (float*)((char*)u.carry + carry_only_bytes(u.n, u.d));
        const int64_t g_ld = dpad_of(u.d);
        J.sum_only = u.sum_out ? 1 : 0;
        J.sum = u.sum_out ? ApplySum{u.sum_out, u.sum_out_ld, u.sum_out_rows} : ApplySum{gsum, g_ld, nullptr};
        J.sa.skip_singletons = 1;
        const char* pp = (const char*)u.plan;
        J.sa.pos_plan = (const int4*)pp;
        J.sa.chunk_plan = (const int4*)(pp + plan_pos_bytes(u.n));
        const int4* row_plan = (const int4*)(pp + plan_pos_bytes(u.n) + plan_chunk_bytes(u.n));
        const int vpr = u.d / 4;
        const int tx = adagrad_tx(vpr);
        const int ty = 256 / tx;
        const unsigned row_blocks = (unsigned)cdiv(u.n, (int64_t)ty * MARIUS_ADAGRAD_UNR);
        J.A = AdagradRowsArgs{gsum, g_ld, u.perm, u.inverse, u.n, u.uniq_ids, u.table, u.state, u.table_ld, vpr, tx, u.lr, u.eps, u.rows, u.rows_ld, u.seg_offsets, row_plan, 1, u.absmax,
                              (int)(u.fused_below > 0 ? u.fused_below : 0)};
        J.ada = ApplyAdagrad{u.uniq_ids, u.table, u.state, u.table_ld, u.lr, u.eps, u.absmax};
        J.nfix = (unsigned)cdiv(cdiv(u.n, SEG_R), 4);
        J.red0 = red;
        J.upd0 = upd;
        red += J.nfix;  // the reduce grid has one workgroup per four chunks too
        upd += J.nfix + row_blocks;
    }
    if (per0 <= 1) {
        seg_reduce_group_kernel<4, 1><<<dim3(red), dim3(256), 0, st>>>(G);
        adagrad_with_fixup_group_kernel<4, 1><<<dim3(upd), dim3(256), 0, st>>>(G);
    } else {
        seg_reduce_group_kernel<4, 2><<<dim3(red), dim3(256), 0, st>>>(G);
        adagrad_with_fixup_group_kernel<4, 2><<<dim3(upd), dim3(256), 0, st>>>(G);
    }
    return check_launch("segment_adagrad_scatter_group");
}
