// Shared definitions for the link-prediction decoder kernels (lp_*.hip).
#pragma once
#include "common.h"

namespace marius {

typedef float v16f __attribute__((ext_vector_type(16)));

// MFMA f32 32x32x2 fragment conventions used throughout (guide: cdna_hip_programming §3):
//   acc = mfma(a, b, acc):  D[m][n] += sum_k A[m][k] * B[k][n]
//   lane l supplies A[m = l & 31][k = l >> 5] and B[k = l >> 5][n = l & 31]
//   lane l holds D[m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][n = l & 31] in register r (0..15)
// K is consumed four at a time ("quad" q): half-wave h = l >> 5 owns k = 4q + 2h + e for step e in {0,1}; both
// operands use the same assignment, so any K permutation is legal (the contraction is a plain sum).
__device__ __forceinline__ v16f mfma32(float a, float b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Guarded 4-float read of row[k .. k+3] (zero beyond `limit`); vec = widest aligned access the row allows.
__device__ __forceinline__ float4 load_row4(const float* __restrict__ row, int k, int limit, int vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k + 3 < limit) {
        if (vec == 4) {
            v = *reinterpret_cast<const float4*>(row + k);
        } else if (vec == 2) {
            float2 a = *reinterpret_cast<const float2*>(row + k);
            float2 b = *reinterpret_cast<const float2*>(row + k + 2);
            v = make_float4(a.x, a.y, b.x, b.y);
        } else {
            v = make_float4(row[k], row[k + 1], row[k + 2], row[k + 3]);
        }
    } else {
        if (k < limit) v.x = row[k];
        if (k + 1 < limit) v.y = row[k + 1];
        if (k + 2 < limit) v.z = row[k + 2];
    }
    return v;
}

struct LpDims {
    int64_t B, Bp;
    int Bc, C, N, d, ndir, edge_cols, relop, cmp;
    int64_t n_ld, d_ld;
    float gscale;  // 1 (SUM) or 1/Bp (MEAN)
};

}  // namespace marius
