// The index plan of the segmented update (marius_segment_plan): layout and the per-position rule, shared by the stand-alone plan launch
// (segreduce.hip) and the fused map launch (sort_unique.hip: marius_prepare_maps computes the plan as its last phase).
#pragma once
#include <type_traits>

#include "common.h"

namespace marius {

constexpr int SEG_R = 32;  // sorted positions per wave of the segmented reduction (segreduce.hip)

inline size_t plan_pos_bytes(int64_t n) { return ((size_t)(n > 0 ? n : 1) * sizeof(int4) + 255) / 256 * 256; }
inline size_t plan_chunk_bytes(int64_t n) { return ((size_t)cdiv(n > 0 ? n : 1, SEG_R) * sizeof(int4) + 255) / 256 * 256; }
inline size_t plan_occ_bytes(int64_t n) { return ((size_t)(n > 0 ? n : 1) + 255) / 256 * 256; }

// [pos_plan | chunk_plan | row_plan | occ_single]
struct SegPlanPtrs {
    int4* pos_plan;
    int4* chunk_plan;
    int4* row_plan;
    uint8_t* occ_single;
};
inline SegPlanPtrs seg_plan_ptrs(void* plan, int64_t n) {
    char* p = (char*)plan;
    return {(int4*)p, (int4*)(p + plan_pos_bytes(n)), (int4*)(p + plan_pos_bytes(n) + plan_chunk_bytes(n)), (uint8_t*)(p + 2 * plan_pos_bytes(n) + plan_chunk_bytes(n))};
}

// everything the reduce / fix-up / update kernels derive from perm / inverse / seg_offsets alone, for sorted position k (k < n) and unique row k:
//   pos_plan[k]   {occurrence row, unique index, segment inside its SEG_R chunk, 2 = dead (padding id) | 1 = singleton | 0}
//   chunk_plan[c] (written by the chunk's last position) {owns a boundary-crossing segment, its unique index, segment starts inside the chunk, last chunk it reaches}
//   row_plan[k]   per unique row k < U {table row id lo, hi, occurrence row of a singleton or -1, segment crosses a chunk boundary}; k >= U: (-1, -1, -1, 0)
//   occ_single[p] the singleton flag by occurrence row (marius_lp_desc.upd_occ_single)
// A NEGATIVE id is a padding slot, not a row (the unused slots of a fixed-capacity exchange block, exchange.hip): its positions are dead — never
// loaded, never stored, whatever the form — and the segment they make up owns no fix-up and no table row.
// COH: the inputs were written earlier in the SAME launch (the fused map launch): agent-scope loads (sort_unique.hip: rs_ld)
template <bool COH = false>
__device__ __forceinline__ void seg_plan_position(int64_t k, int64_t n, int64_t U, const int32_t* __restrict__ perm, const int64_t* __restrict__ inverse,
                                                  const int32_t* __restrict__ seg_offsets, const int64_t* __restrict__ uniq, const SegPlanPtrs& P) {
    auto ld = [](auto* q) {
        typedef std::remove_cv_t<std::remove_pointer_t<decltype(q)>> T;
        if constexpr (COH) return __hip_atomic_load(const_cast<T*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *q;
    };
    const int64_t k0 = k / SEG_R * SEG_R, k1 = min(k0 + SEG_R, n);
    const int p = ld(perm + k);
    const int u = (int)ld(inverse + p);
    const int s0 = ld(seg_offsets + u), s1 = ld(seg_offsets + u + 1);
    const bool dead = ld(uniq + u) < 0;
    P.pos_plan[k] = make_int4(p, u, (dead || (s0 >= k0 && s1 <= k1)) ? 1 : 0, dead ? 2 : ((s1 - s0 == 1) ? 1 : 0));
    P.occ_single[p] = (!dead && s1 - s0 == 1) ? 1 : 0;
    if (k == k1 - 1) {  // last position of its chunk: does the chunk own a boundary-crossing segment (the one its last position belongs to)?
        const bool owner = !dead && (s0 >= k0) && (s1 > k1);
        P.chunk_plan[k / SEG_R] = make_int4(owner ? 1 : 0, u, (s0 != k0) ? 1 : 0, (int)((s1 - 1) / SEG_R));
    }
    if (k < U) {  // position k also describes unique row k
        const int64_t id = ld(uniq + k);
        const int t0 = ld(seg_offsets + k), t1 = ld(seg_offsets + k + 1);
        if (id < 0) P.row_plan[k] = make_int4(-1, -1, -1, 0);
        else P.row_plan[k] = make_int4((int)(id & 0xffffffffll), (int)(id >> 32), (t1 - t0 == 1) ? ld(perm + t0) : -1, (t0 / SEG_R != (t1 - 1) / SEG_R) ? 1 : 0);
    } else {
        P.row_plan[k] = make_int4(-1, -1, -1, 0);
    }
}

}  // namespace marius
