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
    float gscale;  // d(reduced loss)/d(loss term): 1 (SUM); MEAN: 1/Bp (SoftmaxCE: one term per row), 1/(Bp N) (Ranking), 1/(Bp (1+N)) (others)
    int loss;      // MARIUS_LOSS_* with CROSS_ENTROPY folded into SOFTMAX_CE
    float margin;
};

// ---- MFMA contraction kernels: argument blocks and tile constants (shared by lp_decoder.hip and lp_fast.hip)
struct ScoreArgs {
    const float* adj;  // [ndir][Bp, d_ld]
    const float* emb;
    int64_t emb_ld;
    int emb_vec;
    const int64_t* negmap[2];  // [C, N] batch-local
    float* S;                  // [ndir][Bp, n_ld]
    const float* x2;           // L2
    const float* y2;           // L2
    int KC, KS, nkc, dk;
    int ablate;  // debug only (MARIUS_ABLATE): 1 = skip S stores, 2 = skip MFMAs, 4 = skip negative-tile staging
    unsigned long long* dbg;  // debug only: per-phase s_memtime stamps of the first workgroups (MARIUS_DBG_TIMELINE)
    float* lse_part;  // optional [ngroups][ndir][Bp][2]: per (negative-tile group, row) running (max, sum exp) from the score epilogue
    LpDims D;
};

constexpr int F_TM = 128, F_TN = 128;

struct GradArgs {
    const float* S;
    const float* lse;  // [ndir][Bp]
    const float* adj;  // [ndir][Bp, d_ld]
    const float* emb;
    int64_t emb_ld;
    int emb_vec;
    const int64_t* negmap[2];
    float* dadj;            // [ndir][Bp, d_ld]
    float* gocc;            // [L, d_ld]
    int64_t negocc_off[2];  // first gocc row of dir's negatives
    int ncols;              // useful columns per n-block (128, or 127 when the ones column is appended for L2)
    int ablate;             // debug only (MARIUS_ABLATE): 2 = skip MFMAs, 4 = skip V staging, 8 = skip B staging, 16 = skip exp
    unsigned long long* dbg;  // debug only: cycle stamps (marius_debug_set_timeline)
    LpDims D;
};

constexpr int G_TM = 64, G_TN = 128, G_KC = 64;
constexpr int G_KSA = G_KC + 2;   // [m][k] layout, b64 fragment reads: stride/2 odd
constexpr int G_TMS = G_TM + 4;   // [k][m] layout
constexpr int G_TNS = G_TN + 4;   // [k][n] layout

// ---- the loss family of loss.cpp:50-187 on scores.  term(x, y): the loss of one score x with label y (1 = positive, 0 = negative);
// dterm: its derivative.  `rowval` is the per-row scalar the loss needs (SoftmaxCE: lse; Ranking: pos - margin).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
// binary_cross_entropy(sigmoid(x), y) exactly as ATen evaluates it: log terms clamped at -100, backward (p - y) / max(p (1 - p), 1e-12)
// times sigmoid'(x) = p (1 - p)  (Loss.cpp: binary_cross_entropy_out_cpu / _backward)
__device__ __forceinline__ float bce_sig_term(float x, float y) {
    const float p = sigmoidf_(x);
    const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.f - p), -100.f);
    return -(y * lp + (1.f - y) * lq);
}
__device__ __forceinline__ float bce_sig_dterm(float x, float y) {
    const float p = sigmoidf_(x);
    const float pq = p * (1.f - p);
    return (p - y) / fmaxf(pq, 1e-12f) * pq;
}
__device__ __forceinline__ float loss_term(int loss, float x, float y) {
    switch (loss) {
        case MARIUS_LOSS_BCE_AFTER_SIGMOID: return bce_sig_term(x, y);
        case MARIUS_LOSS_BCE_WITH_LOGITS:   // (1 - y) x + log(1 + e^-|x|) + max(-x, 0)   (Loss.cpp: binary_cross_entropy_with_logits)
            return (1.f - y) * x + (fmaxf(-x, 0.f) + log1pf(__expf(-fabsf(x))));
        case MARIUS_LOSS_MSE: return (x - y) * (x - y);
        case MARIUS_LOSS_SOFTPLUS: {        // softplus(-(2y - 1) x), beta 1, threshold 20
            const float z = -(2.f * y - 1.f) * x;
            return z > 20.f ? z : log1pf(__expf(z));
        }
        default: return 0.f;
    }
}
__device__ __forceinline__ float loss_dterm(int loss, float x, float y) {
    switch (loss) {
        case MARIUS_LOSS_BCE_AFTER_SIGMOID: return bce_sig_dterm(x, y);
        case MARIUS_LOSS_BCE_WITH_LOGITS: return sigmoidf_(x) - y;
        case MARIUS_LOSS_MSE: return 2.f * (x - y);
        case MARIUS_LOSS_SOFTPLUS: {
            const float sg = -(2.f * y - 1.f);
            const float z = sg * x;
            return sg * (z > 20.f ? 1.f : sigmoidf_(z));
        }
        default: return 0.f;
    }
}
// dL/dS of a negative score (label 0) for any loss; Ranking: 1[s - pos + margin > 0]
__device__ __forceinline__ float loss_dneg(int loss, float s, float rowval, float gscale) {
    if (loss == MARIUS_LOSS_SOFTMAX_CE) return gscale * __expf(s - rowval);
    if (loss == MARIUS_LOSS_RANKING) return (s - rowval > 0.f) ? gscale : 0.f;
    return gscale * loss_dterm(loss, s, 0.f);
}
// generic-loss form of dscore (used by the generic backward kernels when D.loss != SOFTMAX_CE)
template <bool L2>
__device__ __forceinline__ float dscore_any(const LpDims& D, float s, float rowval) {
    const float q = loss_dneg(D.loss, s, rowval, D.gscale);
    if (L2) return (s > 1.0000001e-4f) ? (-q / s) : 0.f;
    return q;
}

template <bool L2>
__device__ __forceinline__ float dscore(float s, float lse, float gscale) {
    // dL/dS (Dot) or dL/d(x.y) (L2: S = sqrt(max(t,1e-8)), t = x2 + y2 - 2 x.y  =>  -q / S, zero where clamped)
    if (L2) {
        const float q = gscale * __expf(s - lse);
        return (s > 1.0000001e-4f) ? (-q / s) : 0.f;
    }
    return gscale * __expf(s - lse);
}

// fast variants (lp_fast.hip): require d % 4 == 0 and 16-B aligned embedding rows; return false if not applicable
bool launch_scores_fast(const ScoreArgs& a, bool l2, hipStream_t st);
bool launch_grad_adj_fast(const GradArgs& a, bool l2, hipStream_t st);
bool launch_grad_neg_fast(const GradArgs& a, bool l2, hipStream_t st);
// geometry of the resident-operand score kernel: 64-column negative tiles, `ntpg` tiles per workgroup, `ngroups` groups per chunk
inline void scores_res_geometry(int N, int& ntpg, int& ngroups) {
    const int ntiles = (N + 63) / 64;
    ntpg = ntiles >= 8 ? 4 : ntiles;
    ngroups = (ntiles + ntpg - 1) / ntpg;
}
bool scores_res_applicable(const float* emb, int64_t emb_ld, int d);
bool scores_a_applicable(const float* emb, int64_t emb_ld, int d);
// adj-in-registers score kernel, persistent workgroups: units of two 32-column tiles, one SoftmaxCE partial per (row, unit)
inline int scores_ap_groups(int N) { return ((N + 31) / 32 + 1) / 2; }
bool launch_scores_ap(const ScoreArgs& a, bool l2, hipStream_t st);
// resident-operand / 16x16x4 variants (lp_res.hip): additionally d <= 128 for the score kernel
bool launch_scores_res(const ScoreArgs& a, bool l2, hipStream_t st);
bool launch_grad16(const GradArgs& a, bool l2, int which, hipStream_t st);  // which: 0 both (one launch), 1 dAdj, 2 dNeg

// flash-style training path (lp_flash.hip): operand records, SoftmaxCE row statistics, recomputing backward
bool flash_applicable(const marius_lp_desc* desc, const LpDims& D);
size_t flash_adjrec_bytes(const LpDims& D);
size_t flash_negrec_bytes(const LpDims& D);
size_t flash_part_bytes(const LpDims& D);
const float* flash_part_weights(const LpDims& D, const float2* part);  // [2][ndir Bp] g exp(m_k - lse), written by flash_merge
// adj_packed: lp_prep2_kernel already wrote the adj records (and zeroed dadj); otherwise they are packed from `adj` here and dadj_zero
// (if given) is zeroed.  The negatives' gocc rows are always zeroed by the negative pack kernel.
bool flash_fused();  // forward statistics + dAdj in one sweep (default); false: MARIUS_FLASH_FUSED=0
int flash_forward(const marius_lp_desc* desc, const LpDims& D, const float* adj, char* adjrec, char* negrec, float2* part, float* S, bool adj_packed,
                  float* gocc, const int64_t negocc_off[2], float* dadj_zero, const float* pos, float* dadj, float* dadj2, hipStream_t st);
int flash_merge(const LpDims& D, const float2* part, const float* pos, float* lse, float* rowloss, float* dpos, float* blocksum, char* adjrec, bool f16,
                hipStream_t st);
int flash_backward(const marius_lp_desc* desc, const LpDims& D, char* adjrec, char* negrec, float* dadj, float* gocc, const int64_t negocc_off[2],
                   const float2* part, bool filtered, float* S, hipStream_t st);
int flash_chunks(int d);   // column chunks of the contraction index: 1 for d <= 128, ceil(d / 128) equal ones above, 0 = not representable
bool flash_chunked(int d); // d > 128: stored scores + per-chunk launches
bool flash_tail4(int d);   // the records of this width fold their last four columns into the hi region (lp_flash.hip: fl_pitch)
size_t flash_tiled_scores_bytes(const LpDims& D);  // the stored scores in tile order (internal layout; row-major [Bp, n_ld] with MARIUS_LP_STORE_SCORES)
// fp16 operand records (lp_flash.hip): the scale an operand set is packed with, derived on the device from marius_lp_desc.absmax
struct FlRange {
    const float* absmax;      // bound on |node rows|; nullptr: bf16 records, scales 1
    const float* absmax_rel;  // bound on |relation rows| (marius_lp_desc.absmax_rel, or absmax + 1)
    int adj_bound;            // FL_ADJ_*: how the bound on |adj| = |op(e, r)| follows from the two
};
// |op(e, r)| per relation operator (relation_operators.cpp:7-42): no relation / NoOp: M_e; Hadamard: M_e M_r; ComplexHadamard: each output
// is a sum or difference of two products: 2 M_e M_r; Translation: M_e + M_r
enum { FL_ADJ_NODE = 0, FL_ADJ_PRODUCT = 1, FL_ADJ_PRODUCT2 = 2, FL_ADJ_SUM = 3 };
FlRange flash_range(const marius_lp_desc* desc, const LpDims& D);
template <bool F16>
__device__ __forceinline__ unsigned short fl_cvt16(float x) {
    if constexpr (F16) {
        const float c = fminf(fmaxf(x, -65504.f), 65504.f);
        return __builtin_bit_cast(unsigned short, (_Float16)c);
    } else {
        return __builtin_bit_cast(unsigned short, (__bf16)x);
    }
}
// two values at once, no saturation (for magnitudes known to fit: the V factors of the flash kernels are <= 2^14 by construction):
// v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round to nearest even
template <bool F16>
__device__ __forceinline__ unsigned fl_cvt16x2(float x, float y) {
    typedef float f2_ __attribute__((ext_vector_type(2)));
    const f2_ v = {x, y};
    if constexpr (F16) {
        typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2_));
    } else {
        typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_));
    }
}
// (w0, w1) - (the two 16-bit values packed in `hi`): the low halves of a split.  With fp16 halves one v_fma_mix_f32 each — an fp32 FMA that reads
// its fp16 operand straight from either half of the packed register — instead of a conversion and a subtraction (hipcc does not form it itself).
template <bool F16>
__device__ __forceinline__ void fl_lo_pair(float w0, float w1, unsigned hi, float& lo0, float& lo1) {
    if constexpr (F16) {
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo0) : "v"(hi), "v"(w0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo1) : "v"(hi), "v"(w1));
    } else {
        lo0 = w0 - __builtin_bit_cast(float, hi << 16);
        lo1 = w1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    }
}
template <bool F16>
__device__ __forceinline__ float fl_back16(unsigned short b) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, b);
    else return (float)__builtin_bit_cast(__bf16, b);
}
// power-of-two scale that puts a magnitude bound M at [2^11, 2^12); 1 for M == 0 (or a non-finite bound)
__host__ __device__ __forceinline__ float fl_scale_of(float M) {
    if (!(M > 0.f) || !(M < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(M, &e);  // M = m 2^e, m in [0.5, 1)
    return ldexpf(1.f, 12 - e);
}
struct FlScales {
    float s_adj, s_neg;  // operand scales of the adj records and of the negative-row records (1, 1 on the bf16 path)
};
// the scales every kernel of a step derives, on the device, from the same two bounds (so they agree without a host read-back)
__device__ __forceinline__ FlScales fl_scales(const FlRange& rg) {
    FlScales f;
    f.s_adj = f.s_neg = 1.f;
    if (rg.absmax) {
        const float me = rg.absmax[0];
        f.s_neg = fl_scale_of(me);
        float ma = me;
        if (rg.adj_bound != FL_ADJ_NODE) {
            const float mr = rg.absmax_rel[0];
            ma = rg.adj_bound == FL_ADJ_SUM ? me + mr : me * mr * (rg.adj_bound == FL_ADJ_PRODUCT2 ? 2.f : 1.f);
        }
        f.s_adj = fl_scale_of(ma);
    }
    return f;
}

}  // namespace marius
