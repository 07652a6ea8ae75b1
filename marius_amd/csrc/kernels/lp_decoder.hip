// CORRUPT_NODE link-prediction decoder: forward scores, SoftmaxCE, hand-derived backward.  gfx950 only.
//
// Reference path replaced (src/cpp/src): nn/decoders/edge/decoder_methods.cpp:57-114 (node_corrupt_forward),
// relation_operators.cpp:7-47, comparators.cpp:7-73, data/samplers/negative.cpp:306-311 (apply_score_filter),
// nn/loss.cpp:50-67 (SoftmaxCrossEntropy), and the autograd backward of all of these (nn/model.cpp:324).
//
// Work split per batch (dir 0 = (src,rel)->dst with dst negatives, dir 1 = (dst,inv_rel)->src with src negatives):
//   prep      adj = op(e, r) rows [Bp, d] (zero rows for the pad_and_reshape padding), pos = cmp(adj, other)      HBM-bound
//   scores    S_c = adj_c [Bc x d] . Neg_c^T [d x N] per (chunk, dir): batched NT contraction, FP32 MFMA 32x32x2   MFMA-bound
//   filter    S[e][k] = -1e9                                                                                       tiny
//   lse       per-row logsumexp over {pos, S row}, row loss                                                        HBM-bound (reads S once)
//   grad_adj  dAdj_c [Bc x d] = V_c [Bc x N] . Neg_c [N x d],  V = dL/dS (exp(S - lse), L2: -q/S), made on the fly MFMA-bound
//   grad_neg  dNeg_c [N x d]  = V_c^T [N x Bc] . adj_c [Bc x d]                                                     MFMA-bound
//   edge_bwd  positive-score gradient + relation-operator backward -> per-occurrence node grads, per-edge rel grads HBM-bound
// The negative rows are gathered by batch-local index straight into LDS (the "LDS-staged transpose": fragments are
// read from LDS in MFMA operand order, no transposed copy ever exists in HBM).  No atomics anywhere: every output
// element has exactly one owner workgroup.
#include <cstdlib>

#include "lp_common.h"

namespace marius {

// =========================================================================================== prep
struct PrepArgs {
    const float* emb;
    int64_t emb_ld;
    const int64_t* edges;
    const float* rel[2];
    int64_t rel_ld;
    float* adj;  // [ndir][Bp, d_ld]
    float* pos;  // [ndir][Bp]
    float* x2;   // [ndir][Bp] (L2) or null
    // flash path (lp_flash.hip), fused into lp_prep2_kernel: adj operand records (hi | lo bf16 planes, permuted rows) and the zero fill
    // of dadj that the two-contributor accumulation of the backward needs.  frec == nullptr: not the flash path.
    char* frec;
    int fKP, fXR;
    int fP, fTail;  // record pitch in bytes; 1: the folded column tail of lp_flash.hip (fl_pitch) — the last four columns' hi and lo halves share one 16-byte piece
    float* fdadj;  // [ndir][Bp, d_ld]
    FlRange frg;   // fp16 records: the adj scale is derived from it (lp_common.h); absmax == nullptr: bf16 records
    LpDims D;
};

__device__ __forceinline__ float relop_fwd(int relop, const float* e, const float* r, int c, int d) {
#pragma clang fp contract(off)
    switch (relop) {
        case MARIUS_OP_HADAMARD: return e[c] * r[c];
        case MARIUS_OP_TRANSLATION: return e[c] + r[c];
        case MARIUS_OP_COMPLEX_HADAMARD: {
            const int h = d / 2;
            if (c < h) return (e[c] * r[c]) - (e[c + h] * r[c + h]);
            return (e[c - h] * r[c]) + (e[c] * r[c - h]);
        }
        default: return e[c];
    }
}

// one wave per (row, dir); lanes stride over d
__global__ __launch_bounds__(256) void lp_prep_kernel(PrepArgs a) {
#pragma clang fp contract(off)
    const LpDims& D = a.D;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t total = D.Bp * D.ndir;
    if (wave >= total) return;
    const int dir = (int)(wave / D.Bp);
    const int64_t i = wave - (int64_t)dir * D.Bp;
    float* adj = a.adj + ((int64_t)dir * D.Bp + i) * D.d_ld;
    if (i >= D.B) {  // pad_and_reshape zero rows; F.pad zero positives
        for (int c = lane; c < D.d_ld; c += 64) adj[c] = 0.f;
        if (lane == 0) {
            a.pos[(int64_t)dir * D.Bp + i] = 0.f;
            if (a.x2) a.x2[(int64_t)dir * D.Bp + i] = 0.f;
        }
        return;
    }
    const int64_t* ed = a.edges + i * D.edge_cols;
    const int64_t s = ed[0], t = ed[D.edge_cols - 1];
    const float* e = a.emb + (dir == 0 ? s : t) * a.emb_ld;
    const float* o = a.emb + (dir == 0 ? t : s) * a.emb_ld;
    const bool has_rel = (D.edge_cols == 3) && (a.rel[dir] != nullptr);
    const float* r = has_rel ? a.rel[dir] + ed[1] * a.rel_ld : nullptr;
    float acc = 0.f, nrm = 0.f;
    for (int c = lane; c < D.d_ld; c += 64) {
        float v = 0.f;
        if (c < D.d) {
            v = has_rel ? relop_fwd(D.relop, e, r, c, D.d) : e[c];
            const float ov = o[c];
            if (D.cmp == MARIUS_CMP_L2) {
                const float df = (v - ov) + 1e-6f;  // pairwise_distance: ||x1 - x2 + eps||
                acc += df * df;
                nrm += v * v;
            } else {
                acc += v * ov;
            }
        }
        adj[c] = v;
    }
    acc = wave_sum(acc);
    if (D.cmp == MARIUS_CMP_L2) {
        nrm = wave_sum(nrm);
        acc = sqrtf(acc);
    }
    if (lane == 0) {
        a.pos[(int64_t)dir * D.Bp + i] = acc;
        if (a.x2) a.x2[(int64_t)dir * D.Bp + i] = nrm;
    }
}

// Vectorised prep: half a wave (32 lanes) per edge, BOTH directions at once (they share the two endpoint rows), lane l owns the
// element pairs {2l, 2l+1} of the first half of the row and {d/2 + 2l, d/2 + 2l + 1} of the second half — for ComplEx exactly the
// (re, im) partners.  Four times fewer waves than lp_prep_kernel and four independent 8-B loads per row and lane instead of a
// dependent chain of 4-B ones (that kernel is latency-bound: 62 us for 120 MB).  Requires d % 4 == 0, d <= 128, even row strides.
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned prep_pack_bf16x2(float x, float y) {
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    v2bf p;
    p[0] = (__bf16)x;
    p[1] = (__bf16)y;
    return __builtin_bit_cast(unsigned, p);
}
// record of row i of direction dir: hi / lo pairs of the lane's four elements, zero K padding, zero tail; dadj row zeroed
__device__ __forceinline__ void prep_flash_store(const PrepArgs& a, int dir, int64_t i, int l, int c0, int c1, const float (&v)[4], bool act) {
#pragma clang fp contract(off)
    const LpDims& D = a.D;
    const int P = a.fP;
    const int64_t c = i / D.Bc;
    const int x = (int)(i - c * D.Bc);
    const int xp = 4 * (x & 3) + ((x >> 2) & 3) + (x & ~15);  // fl_rho
    char* rec = a.frec + (((int64_t)dir * D.C + c) * a.fXR + xp) * (int64_t)P;
    // folded column tail: the lo halves of columns d - 4 .. d - 1 sit 8 bytes behind their hi halves instead of in the lo region
    const int lo0 = (a.fTail && c0 >= D.d - 4) ? 2 * c0 + 8 : 2 * a.fKP + 2 * c0;
    const int lo1 = (a.fTail && c1 >= D.d - 4) ? 2 * c1 + 8 : 2 * a.fKP + 2 * c1;
    if (act) {
        // x s = h + l with both halves in the record's element type (fp16 when the caller gave magnitude bounds — s = the adj scale, a power
        // of two — else bf16 with s = 1): the same split flash_pack_adj_kernel makes (fl_write_piece)
        unsigned short H[4], L[4];
        if (a.frg.absmax) {
            const float sc = fl_scales(a.frg).s_adj;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xs = v[k] * sc;
                H[k] = fl_cvt16<true>(xs);
                L[k] = fl_cvt16<true>(xs - fl_back16<true>(H[k]));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                H[k] = fl_cvt16<false>(v[k]);
                L[k] = fl_cvt16<false>(v[k] - fl_back16<false>(H[k]));
            }
        }
        *reinterpret_cast<unsigned*>(rec + 2 * c0) = (unsigned)H[0] | ((unsigned)H[1] << 16);
        *reinterpret_cast<unsigned*>(rec + 2 * c1) = (unsigned)H[2] | ((unsigned)H[3] << 16);
        *reinterpret_cast<unsigned*>(rec + lo0) = (unsigned)L[0] | ((unsigned)L[1] << 16);
        *reinterpret_cast<unsigned*>(rec + lo1) = (unsigned)L[2] | ((unsigned)L[3] << 16);
        if (a.fdadj) {
            float* z = a.fdadj + ((int64_t)dir * D.Bp + i) * D.d_ld;
            *reinterpret_cast<float2*>(z + c0) = make_float2(0.f, 0.f);
            *reinterpret_cast<float2*>(z + c1) = make_float2(0.f, 0.f);
        }
    } else {
        const int q = l - D.d / 4;             // idle lanes write the zero K padding, one element pair each
        const int col = D.d + 2 * q;
        if (a.fTail) {  // the eight zeros behind T = [h | l]: bytes 2 d + 8 .. 2 d + 23 of the hi region; the lo region has no padding
            if (q < 4) *reinterpret_cast<unsigned*>(rec + 2 * D.d + 8 + 4 * q) = 0u;
        } else if (col < a.fKP) {
            *reinterpret_cast<unsigned*>(rec + 2 * col) = 0u;
            *reinterpret_cast<unsigned*>(rec + 2 * a.fKP + 2 * col) = 0u;
        }
    }
    if (l == 0) *reinterpret_cast<float4*>(rec + P - 16) = make_float4(0.f, 0.f, 0.f, 0.f);  // lsec: patched in by the merge kernel
}

// adj = op(e, r) on the four elements {c0, c0 + 1, c1, c1 + 1} a lane owns (ComplEx: (re, im) = (k, k + 2)); the SAME expressions, with
// contraction off, in the forward (prep2) and — when the flash path does not keep an fp32 copy of adj — in the edge backward
__device__ __forceinline__ void relop4(int relop, bool has_rel, const float (&e)[4], const float (&r)[4], float (&v)[4]) {
#pragma clang fp contract(off)
    if (!has_rel) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = e[k];
    } else if (relop == MARIUS_OP_HADAMARD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = e[k] * r[k];
    } else if (relop == MARIUS_OP_TRANSLATION) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = e[k] + r[k];
    } else if (relop == MARIUS_OP_COMPLEX_HADAMARD) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            v[k] = (e[k] * r[k]) - (e[k + 2] * r[k + 2]);
            v[k + 2] = (e[k] * r[k + 2]) + (e[k + 2] * r[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = e[k];
    }
}

// One edge row of the prep: its three phases, so that a half-wave can walk two rows with the loads of both in flight (lp_prep2_kernel<2>).
#ifndef MARIUS_PREP_EDGES_PER_HALF_WAVE
#define MARIUS_PREP_EDGES_PER_HALF_WAVE 2
#endif
struct Prep2Row {
    int64_t s, t, e1;
    float x[2][4];  // [0] = src row, [1] = dst row; elements {c0, c0+1, c1, c1+1}
    float r[2][4];
};
__device__ __forceinline__ void prep2_ids(const PrepArgs& a, int64_t i, Prep2Row& R) {
    const LpDims& D = a.D;
    const int64_t* ed = a.edges + i * D.edge_cols;
    R.s = ed[0];
    R.t = ed[D.edge_cols - 1];
    R.e1 = D.edge_cols == 3 ? ed[1] : 0;
}
__device__ __forceinline__ void prep2_rows(const PrepArgs& a, int l, Prep2Row& R) {
    const LpDims& D = a.D;
    const int h2 = D.d / 2;
    const bool act = 4 * l < D.d;
    const int c0 = 2 * l, c1 = h2 + 2 * l;
    const float* es = a.emb + R.s * a.emb_ld;
    const float* et = a.emb + R.t * a.emb_ld;
    float (&x)[2][4] = R.x;
    float (&r)[2][4] = R.r;
    const bool has_rel0 = (D.edge_cols == 3) && a.rel[0], has_rel1 = (D.edge_cols == 3) && a.rel[1];
    if (act) {
        const float2 a0 = *reinterpret_cast<const float2*>(es + c0), a1 = *reinterpret_cast<const float2*>(es + c1);
        const float2 b0 = *reinterpret_cast<const float2*>(et + c0), b1 = *reinterpret_cast<const float2*>(et + c1);
        x[0][0] = a0.x; x[0][1] = a0.y; x[0][2] = a1.x; x[0][3] = a1.y;
        x[1][0] = b0.x; x[1][1] = b0.y; x[1][2] = b1.x; x[1][3] = b1.y;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const bool hr = dir == 0 ? has_rel0 : has_rel1;
            if (hr && dir < D.ndir) {
                const float* rr = a.rel[dir] + R.e1 * a.rel_ld;
                const float2 q0 = *reinterpret_cast<const float2*>(rr + c0), q1 = *reinterpret_cast<const float2*>(rr + c1);
                r[dir][0] = q0.x; r[dir][1] = q0.y; r[dir][2] = q1.x; r[dir][3] = q1.y;
            } else {
                r[dir][0] = r[dir][1] = r[dir][2] = r[dir][3] = 0.f;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[0][k] = x[1][k] = r[0][k] = r[1][k] = 0.f;
    }
}
__device__ __forceinline__ void prep2_finish(const PrepArgs& a, int64_t i, int l, const Prep2Row& R) {
#pragma clang fp contract(off)
    const LpDims& D = a.D;
    const int h2 = D.d / 2;
    const bool act = 4 * l < D.d;
    const int c0 = 2 * l, c1 = h2 + 2 * l;
    const float (&x)[2][4] = R.x;
    const float (&r)[2][4] = R.r;
    const bool has_rel0 = (D.edge_cols == 3) && a.rel[0], has_rel1 = (D.edge_cols == 3) && a.rel[1];
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        if (dir >= D.ndir) break;
        const float* e = x[dir];        // operand that goes through the relation operator
        const float* o = x[dir ^ 1];    // the other endpoint
        const bool hr = dir == 0 ? has_rel0 : has_rel1;
        float v[4];
        {
            const float ee[4] = {e[0], e[1], e[2], e[3]};
            relop4(D.relop, hr, ee, r[dir], v);
        }
        float acc = 0.f, nrm = 0.f;
        if (act) {
            if (D.cmp == MARIUS_CMP_L2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float df = (v[k] - o[k]) + 1e-6f;  // pairwise_distance: ||x1 - x2 + eps||
                    acc += df * df;
                    nrm += v[k] * v[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += v[k] * o[k];
            }
            if (a.adj) {  // null: the flash path keeps adj only as operand records (the edge backward recomputes it from the same rows)
                float* adj = a.adj + ((int64_t)dir * D.Bp + i) * D.d_ld;
                *reinterpret_cast<float2*>(adj + c0) = make_float2(v[0], v[1]);
                *reinterpret_cast<float2*>(adj + c1) = make_float2(v[2], v[3]);
            }
        }
        if (a.frec) prep_flash_store(a, dir, i, l, c0, c1, v, act);
        acc = half_sum(acc);
        if (D.cmp == MARIUS_CMP_L2) {
            nrm = half_sum(nrm);
            acc = sqrtf(acc);
        }
        if (l == 0) {
            a.pos[(int64_t)dir * D.Bp + i] = acc;
            if (a.x2) a.x2[(int64_t)dir * D.Bp + i] = nrm;
        }
    }
}
// rows past the edges: the zero rows of pad_and_reshape and the records that pad every chunk block to a multiple of 32 rows
__device__ __forceinline__ void prep2_pad(const PrepArgs& a, int64_t i, int l) {
#pragma clang fp contract(off)
    const LpDims& D = a.D;
    if (i >= D.Bp) {
        // flash path: blocks past the edge rows zero the records that pad every chunk block to a multiple of 32 rows
        if (a.frec) {
            const int64_t r = i - D.Bp;  // pad record index over (dir, chunk, x in [Bc, XR))
            const int npad = a.fXR - D.Bc;
            if (npad > 0 && r < (int64_t)D.ndir * D.C * npad) {
                const int64_t cd = r / npad;
                const int x = D.Bc + (int)(r - cd * npad);
                const int xp = 4 * (x & 3) + ((x >> 2) & 3) + (x & ~15);
                const int P = a.fP;
                char* rec = a.frec + (cd * a.fXR + xp) * (int64_t)P;
                for (int o = 16 * l; o < P; o += 16 * 32) *reinterpret_cast<float4*>(rec + o) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        return;
    }
    const int h2 = D.d / 2;
    const bool act = 4 * l < D.d;          // lanes that own elements
    const int c0 = 2 * l, c1 = h2 + 2 * l;  // first-half pair, second-half pair
    if (i >= D.B) {  // pad_and_reshape zero rows; F.pad zero positives
        for (int dir = 0; dir < D.ndir; ++dir) {
            if (act && a.adj) {
                float* adj = a.adj + ((int64_t)dir * D.Bp + i) * D.d_ld;
                *reinterpret_cast<float2*>(adj + c0) = make_float2(0.f, 0.f);
                *reinterpret_cast<float2*>(adj + c1) = make_float2(0.f, 0.f);
            }
            if (a.frec) {
                const float zz[4] = {0.f, 0.f, 0.f, 0.f};
                prep_flash_store(a, dir, i, l, c0, c1, zz, act);
            }
            if (l == 0) {
                a.pos[(int64_t)dir * D.Bp + i] = 0.f;
                if (a.x2) a.x2[(int64_t)dir * D.Bp + i] = 0.f;
            }
        }
        return;
    }
}
// EP edge rows per half-wave: the row loads of EP edges in flight behind one wait (the dependent chain edge ids -> rows -> stores is what a
// half-wave spends its time on).  Same-box A/B at the bench shape: 1 -> 43.2 us, 2 -> 39.8 us (MARIUS_PREP_EDGES_PER_HALF_WAVE)
template <int EP>
__global__ __launch_bounds__(256) void lp_prep2_kernel(PrepArgs a) {
    const LpDims& D = a.D;
    const int l = threadIdx.x & 31;
    const int64_t i0 = (int64_t)blockIdx.x * (8 * EP) + (threadIdx.x >> 5);
    if constexpr (EP > 1) {
        if (i0 + 8 * (EP - 1) < D.B) {  // all of them are edge rows: ids of all, rows of all, then the arithmetic and the stores
            Prep2Row R[EP];
#pragma unroll
            for (int e = 0; e < EP; ++e) prep2_ids(a, i0 + 8 * e, R[e]);
#pragma unroll
            for (int e = 0; e < EP; ++e) prep2_rows(a, l, R[e]);
#pragma unroll
            for (int e = 0; e < EP; ++e) prep2_finish(a, i0 + 8 * e, l, R[e]);
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int64_t i = i0 + 8 * e;
        if (i >= D.B) {
            prep2_pad(a, i, l);
        } else {
            Prep2Row R;
            prep2_ids(a, i, R);
            prep2_rows(a, l, R);
            prep2_finish(a, i, l, R);
        }
    }
}

// L2 only: y2[dir][c*N + j] = ||neg row||^2
__global__ __launch_bounds__(256) void lp_negnorm_kernel(const float* emb, int64_t emb_ld, const int64_t* neg0, const int64_t* neg1,
                                                         int64_t CN, int d, int ndir, float* y2) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= CN * ndir) return;
    const int dir = (int)(wave / CN);
    const int64_t j = wave - dir * CN;
    const float* row = emb + (dir == 0 ? neg0 : neg1)[j] * emb_ld;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += row[c] * row[c];
    s = wave_sum(s);
    if (lane == 0) y2[wave] = s;
}

// =========================================================================================== scores (F)

template <bool L2>
__global__ __launch_bounds__(256) void lp_scores_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int dir = blockIdx.z / D.C, c = blockIdx.z - dir * D.C;
    const int m0 = blockIdx.y * F_TM, n0 = blockIdx.x * F_TN;
    const int KS = a.KS;
    float* As = smem;
    float* Bs = smem + F_TM * KS;
    const float* adj = a.adj + ((int64_t)dir * D.Bp + (int64_t)c * D.Bc) * D.d_ld;
    const int64_t* negmap = a.negmap[dir] + (int64_t)c * D.N;

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int piece = tid & 15, rbase = tid >> 4;  // 16 threads per row, 16 rows per pass
    // negative row pointers for the 8 rows this thread stages
    const float* nrow[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int n = n0 + rbase + 16 * it;
        nrow[it] = (n < D.N) ? a.emb + negmap[n] * a.emb_ld : nullptr;
    }

    for (int kc = 0; kc < a.nkc; ++kc) {
        const int k0 = kc * a.KC;
        const int kcur = min(a.KC, a.dk - k0);
        if (4 * piece < kcur) {
            const int k = k0 + 4 * piece;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + 16 * it;
                const int m = m0 + row;
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < D.Bc) va = *reinterpret_cast<const float4*>(adj + (int64_t)m * D.d_ld + k);
                float4 vb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nrow[it]) vb = load_row4(nrow[it], k, D.d, a.emb_vec);
                float* pa = As + row * KS + 4 * piece;
                float* pb = Bs + row * KS + 4 * piece;
                *reinterpret_cast<float2*>(pa) = make_float2(va.x, va.y);
                *reinterpret_cast<float2*>(pa + 2) = make_float2(va.z, va.w);
                *reinterpret_cast<float2*>(pb) = make_float2(vb.x, vb.y);
                *reinterpret_cast<float2*>(pb + 2) = make_float2(vb.z, vb.w);
            }
        }
        __syncthreads();
        const float* ap = As + (wm * 64 + l31) * KS + 2 * h;
        const float* bp = Bs + (wn * 64 + l31) * KS + 2 * h;
        const int nq = kcur >> 2;
        for (int q = 0; q < nq; ++q) {
            const float2 a0 = *reinterpret_cast<const float2*>(ap + 4 * q);
            const float2 a1 = *reinterpret_cast<const float2*>(ap + 32 * KS + 4 * q);
            const float2 b0 = *reinterpret_cast<const float2*>(bp + 4 * q);
            const float2 b1 = *reinterpret_cast<const float2*>(bp + 32 * KS + 4 * q);
            acc[0][0] = mfma32(a0.x, b0.x, acc[0][0]);
            acc[0][1] = mfma32(a0.x, b1.x, acc[0][1]);
            acc[1][0] = mfma32(a1.x, b0.x, acc[1][0]);
            acc[1][1] = mfma32(a1.x, b1.x, acc[1][1]);
            acc[0][0] = mfma32(a0.y, b0.y, acc[0][0]);
            acc[0][1] = mfma32(a0.y, b1.y, acc[0][1]);
            acc[1][0] = mfma32(a1.y, b0.y, acc[1][0]);
            acc[1][1] = mfma32(a1.y, b1.y, acc[1][1]);
        }
        __syncthreads();
    }

    float* S = a.S + ((int64_t)dir * D.Bp + (int64_t)c * D.Bc) * D.n_ld;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int n = n0 + wn * 64 + tn * 32 + l31;
            float yy = 0.f;
            if (L2 && n < D.N) yy = a.y2[(int64_t)dir * D.C * D.N + (int64_t)c * D.N + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + acc_row(r, h);
                if (m < D.Bc && n < D.N) {
                    float v = acc[tm][tn][r];
                    if (L2) {
#pragma clang fp contract(off)
                        const float xx = a.x2[(int64_t)dir * D.Bp + (int64_t)c * D.Bc + m];
                        const float t = (xx + yy) - 2.f * v;  // comparators.cpp:37 x2 + y2 - 2*xy
                        v = sqrtf(fmaxf(t, 1e-8f));
                    }
                    S[(int64_t)m * D.n_ld + n] = v;
                }
            }
        }
}

// =========================================================================================== filter / lse / loss
__global__ __launch_bounds__(256) void lp_filter_kernel(float* S, int64_t n_ld, int64_t Bp, int N, const int64_t* filt, int64_t nf) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nf; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = filt[2 * t], k = filt[2 * t + 1];
        if (e >= 0 && e < Bp && k >= 0 && k < N) S[e * n_ld + k] = -1e9f;
    }
}

// one wave per (row, dir): lse = log(e^pos + sum_j e^S_j); rowloss = lse - pos
__global__ __launch_bounds__(256) void lp_lse_kernel(const float* __restrict__ S, int64_t n_ld, const float* __restrict__ pos,
                                                     int64_t rows, int N, float* __restrict__ lse, float* __restrict__ rowloss,
                                                     float* __restrict__ dpos, float gscale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * n_ld;
    const float p = pos[row];
    float m = p;
    const int n4 = N >> 2;
    for (int c = lane; c < n4; c += 64) {
        const float4 v = reinterpret_cast<const float4*>(s)[c];
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int c = (n4 << 2) + lane; c < N; c += 64) m = fmaxf(m, s[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < n4; c += 64) {
        const float4 v = reinterpret_cast<const float4*>(s)[c];
        sum += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m));
    }
    for (int c = (n4 << 2) + lane; c < N; c += 64) sum += __expf(s[c] - m);
    sum = wave_sum(sum);
    if (lane == 0) {
        sum += __expf(p - m);
        const float l = m + __logf(sum);
        lse[row] = l;
        rowloss[row] = l - p;
        if (dpos) dpos[row] = (__expf(p - l) - 1.f) * gscale;
    }
}

// Every other loss of loss.cpp on the materialised scores: one wave per (row, dir).  rowloss = sum of the row's terms, dpos = dL/dpos,
// rowval (stored in the lse array) = what the backward contractions need per row (Ranking: pos - margin).
__global__ __launch_bounds__(256) void lp_loss_terms_kernel(const float* __restrict__ S, int64_t n_ld, const float* __restrict__ pos, int64_t rows, int N,
                                                            int loss, float margin, float gscale, float* __restrict__ rowval,
                                                            float* __restrict__ rowloss, float* __restrict__ dpos, float* __restrict__ vlog) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * n_ld;
    const float p = pos[row];
    float sum = 0.f, cnt = 0.f;
    float* vl = vlog ? vlog + row * n_ld : nullptr;
    // per element: (loss term, log of dL/dS or -inf).  16-B accesses: n_ld is a multiple of 4 and rows are 16-B aligned.
    auto one = [&](float x, bool valid, float& lg) {
        lg = -INFINITY;
        if (!valid) return;
        if (loss == MARIUS_LOSS_RANKING) {
            const float t = x - p + margin;
            if (t > 0.f) {
                sum += t;
                cnt += 1.f;
                lg = 0.f;
            }
        } else if (loss == MARIUS_LOSS_BCE_WITH_LOGITS || loss == MARIUS_LOSS_SOFTPLUS) {
            // label 0: term = softplus(x) = max(x, 0) + log1p(e^-|x|); dterm = sigmoid(x), log sigmoid(x) = x - softplus(x)
            // (SoftPlusLoss: torch's threshold 20 -> term = x, dterm = 1)
            const float sp = fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x)));  // log1p's extra accuracy (< 6e-8 absolute) is below the sum's rounding
            const bool lin = (loss == MARIUS_LOSS_SOFTPLUS) && x > 20.f;
            sum += lin ? x : sp;
            lg = lin ? 0.f : x - sp;
        } else {
            sum += loss_term(loss, x, 0.f);
            const float v = loss_dterm(loss, x, 0.f);  // >= 0 for the losses that use the buffer
            lg = v > 0.f ? __logf(v) : -INFINITY;
        }
    };
    for (int c4 = lane; 4 * c4 < n_ld; c4 += 64) {
        const int c = 4 * c4;
        const float4 x = *reinterpret_cast<const float4*>(s + c);
        float4 lg;
        one(x.x, c < N, lg.x);
        one(x.y, c + 1 < N, lg.y);
        one(x.z, c + 2 < N, lg.z);
        one(x.w, c + 3 < N, lg.w);
        if (vl) *reinterpret_cast<float4*>(vl + c) = lg;
    }
    sum = wave_sum(sum);
    cnt = wave_sum(cnt);
    if (lane == 0) {
        if (loss == MARIUS_LOSS_RANKING) {
            rowval[row] = vlog ? 0.f : p - margin;
            rowloss[row] = sum;
            if (dpos) dpos[row] = -cnt * gscale;
        } else {
            rowval[row] = 0.f;
            rowloss[row] = sum + loss_term(loss, p, 1.f);
            if (dpos) dpos[row] = loss_dterm(loss, p, 1.f) * gscale;
        }
    }
}

// merge the per-group partials of the fused score epilogue with the positive score: lse = log(e^pos + sum_g l_g e^{m_g})
__device__ __forceinline__ float lp_lse_merge_row(const float* __restrict__ part, int ng, const float* __restrict__ pos, int64_t rows,
                                                  float* __restrict__ lse, float* __restrict__ rowloss, float* __restrict__ dpos, float gscale, int64_t row) {
    const float p = pos[row];
    // part is [group][row][2]: consecutive threads read consecutive 8-B pairs
    const float2* pr = reinterpret_cast<const float2*>(part) + row;
    float m = p, sum;
    if (ng <= 16) {  // the common case (N <= 1024): all partials in flight at once, one pass over memory
        float2 v[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) v[g] = g < ng ? pr[(int64_t)g * rows] : make_float2(-INFINITY, 0.f);
#pragma unroll
        for (int g = 0; g < 16; ++g) m = fmaxf(m, v[g].x);
        sum = __expf(p - m);
#pragma unroll
        for (int g = 0; g < 16; ++g) sum += v[g].y * __expf(v[g].x - m);
    } else {
        for (int g = 0; g < ng; ++g) m = fmaxf(m, pr[(int64_t)g * rows].x);
        sum = __expf(p - m);
        for (int g = 0; g < ng; ++g) {
            const float2 v = pr[(int64_t)g * rows];
            sum += v.y * __expf(v.x - m);
        }
    }
    const float l = m + __logf(sum);
    lse[row] = l;
    rowloss[row] = l - p;
    dpos[row] = (__expf(p - l) - 1.f) * gscale;
    return l - p;
}

// blocksum[block] = sum of the block's row losses in a fixed order, so that the final reduction adds a few hundred numbers instead of
// re-reading every row (blocks never straddle the two directions: the launch covers each direction with its own blocks)
__global__ __launch_bounds__(256) void lp_lse_merge_kernel(const float* __restrict__ part, int ng, const float* __restrict__ pos, int64_t rows,
                                                           float* __restrict__ lse, float* __restrict__ rowloss, float* __restrict__ dpos, float gscale,
                                                           int64_t Bp, float* __restrict__ blocksum) {
    __shared__ float red[256];
    const int64_t bpd = (Bp + 255) / 256;                       // blocks per direction
    const int64_t dir = blockIdx.x / bpd, blk = blockIdx.x - dir * bpd;
    const int64_t r = blk * 256 + threadIdx.x;
    const int64_t row = dir * Bp + r;
    float mine = 0.f;
    if (r < Bp) mine = lp_lse_merge_row(part, ng, pos, rows, lse, rowloss, dpos, gscale, row);
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) blocksum[blockIdx.x] = red[0];
}

// sum of the per-block partials: loss[1 + dir], loss[0] = rhs + lhs (model.cpp:309-312)
__global__ __launch_bounds__(256) void lp_loss_reduce_blocks_kernel(const float* __restrict__ blocksum, int64_t bpd, int ndir, float scale, float* loss) {
    __shared__ float red[256];
    float tot[2] = {0.f, 0.f};
    for (int dir = 0; dir < ndir; ++dir) {
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < bpd; i += 256) s += blocksum[dir * bpd + i];
        __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        tot[dir] = red[0] * scale;
    }
    if (threadIdx.x == 0) {
        loss[1] = tot[0];
        loss[2] = ndir == 2 ? tot[1] : 0.f;
        loss[0] = tot[0] + (ndir == 2 ? tot[1] : 0.f);
        loss[3] = 0.f;
    }
}


// deterministic sum of rowloss: ONE block reduces each direction in turn -> loss[1 + dir], then loss[0] = rhs + lhs (model.cpp:309-312)
__global__ __launch_bounds__(1024) void lp_loss_reduce_kernel(const float* rowloss, int64_t Bp, int ndir, float scale, float* loss) {
    __shared__ float red[1024];
    float tot[2] = {0.f, 0.f};
    for (int dir = 0; dir < ndir; ++dir) {
        const float* r = rowloss + (int64_t)dir * Bp;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int64_t i = threadIdx.x;
        for (; i + 3 * 1024 < Bp; i += 4 * 1024) {
            s0 += r[i];
            s1 += r[i + 1024];
            s2 += r[i + 2048];
            s3 += r[i + 3072];
        }
        for (; i < Bp; i += 1024) s0 += r[i];
        __syncthreads();  // red[] of the previous direction fully consumed
        red[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        tot[dir] = red[0] * scale;
    }
    if (threadIdx.x == 0) {
        loss[1] = tot[0];
        loss[2] = ndir == 2 ? tot[1] : 0.f;
        loss[0] = tot[0] + (ndir == 2 ? tot[1] : 0.f);
        loss[3] = 0.f;
    }
}

// =========================================================================================== backward contractions
// dAdj_c[m, n] = sum_j V[m, j] * Neg_c[j, n]     (M = rows of the chunk, K = negatives, N = embedding columns)
template <bool L2>
__global__ __launch_bounds__(256) void lp_grad_adj_kernel(GradArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[G_TM * G_KSA];
    __shared__ __attribute__((aligned(16))) float Bs[G_KC * G_TNS];
    __shared__ float rsum[G_TM];
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int dir = blockIdx.z / D.C, c = blockIdx.z - dir * D.C;
    const int m0 = blockIdx.y * G_TM, n0 = blockIdx.x * a.ncols;
    const int64_t rowbase = (int64_t)dir * D.Bp + (int64_t)c * D.Bc;
    const float* S = a.S + rowbase * D.n_ld;
    const int64_t* negmap = a.negmap[dir] + (int64_t)c * D.N;

    v16f acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // staging roles
    const int qpiece = tid & 15, qrow = tid >> 4;   // Q: 16 threads x float4 = 64 k per row, 16 rows per pass, 4 passes
    const int bpiece = tid & 31, brow = tid >> 5;   // B: 32 threads x float4 = 128 n per row, 8 rows per pass, 8 passes
    float lse_r[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int m = m0 + qrow + 16 * it;
        lse_r[it] = (m < D.Bc) ? a.lse[rowbase + m] : 0.f;
    }

    for (int j0 = 0; j0 < D.N; j0 += G_KC) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = qrow + 16 * it;
            const int m = m0 + row;
            const int j = j0 + 4 * qpiece;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < D.Bc && j < D.N) {
                const float4 s = *reinterpret_cast<const float4*>(S + (int64_t)m * D.n_ld + j);
                v.x = dscore_any<L2>(D, s.x, lse_r[it]);
                if (j + 1 < D.N) v.y = dscore_any<L2>(D, s.y, lse_r[it]);
                if (j + 2 < D.N) v.z = dscore_any<L2>(D, s.z, lse_r[it]);
                if (j + 3 < D.N) v.w = dscore_any<L2>(D, s.w, lse_r[it]);
            }
            float* p = Qs + row * G_KSA + 4 * qpiece;
            *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(p + 2) = make_float2(v.z, v.w);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = brow + 8 * it;
            const int j = j0 + row;
            const int nl = 4 * bpiece;  // local column
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < D.N) {
                const float* nr = a.emb + negmap[j] * a.emb_ld;
                if (nl < a.ncols) {
                    const int lim = min(D.d, n0 + a.ncols);
                    v = load_row4(nr, n0 + nl, lim, (n0 & 3) ? 1 : a.emb_vec);
                }
                if (L2 && nl == G_TN - 4) v.w = 1.f;  // ones column -> row sums of V
            }
            *reinterpret_cast<float4*>(Bs + row * G_TNS + nl) = v;
        }
        __syncthreads();
        const float* ap = Qs + (wm * 32 + l31) * G_KSA + 2 * h;
        const float* bp = Bs + (2 * h) * G_TNS + wn * 64 + l31;
#pragma unroll 4
        for (int q = 0; q < G_KC / 4; ++q) {
            const float2 av = *reinterpret_cast<const float2*>(ap + 4 * q);
            const float b00 = bp[(4 * q) * G_TNS], b01 = bp[(4 * q) * G_TNS + 32];
            const float b10 = bp[(4 * q + 1) * G_TNS], b11 = bp[(4 * q + 1) * G_TNS + 32];
            acc[0] = mfma32(av.x, b00, acc[0]);
            acc[1] = mfma32(av.x, b01, acc[1]);
            acc[0] = mfma32(av.y, b10, acc[0]);
            acc[1] = mfma32(av.y, b11, acc[1]);
        }
        __syncthreads();
    }

    if (L2) {  // local column 127 holds sum_j V[m, j]
        if (wn == 1 && l31 == 31) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rsum[wm * 32 + acc_row(r, h)] = acc[1][r];
        }
        __syncthreads();
    }
    float* out = a.dadj + rowbase * D.d_ld;
    const float* adj = a.adj + rowbase * D.d_ld;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int nl = wn * 64 + tn * 32 + l31;
        const int n = n0 + nl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 + acc_row(r, h);
            const int m = m0 + ml;
            if (m < D.Bc && nl < a.ncols && n < D.d) {
                float v = acc[tn][r];
                if (L2) v -= adj[(int64_t)m * D.d_ld + n] * rsum[ml];  // + a_i * sum_j (q/S)
                out[(int64_t)m * D.d_ld + n] = v;
            }
        }
    }
}

// dNeg_c[m, n] = sum_i V[i, m] * adj_c[i, n]     (M = negatives of the chunk, K = rows of the chunk, N = embedding columns)
template <bool L2>
__global__ __launch_bounds__(256) void lp_grad_neg_kernel(GradArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[G_KC * G_TMS];
    __shared__ __attribute__((aligned(16))) float Bs[G_KC * G_TNS];
    __shared__ float csum[G_TM];
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int dir = blockIdx.z / D.C, c = blockIdx.z - dir * D.C;
    const int m0 = blockIdx.y * G_TM, n0 = blockIdx.x * a.ncols;
    const int64_t rowbase = (int64_t)dir * D.Bp + (int64_t)c * D.Bc;
    const float* S = a.S + rowbase * D.n_ld;
    const float* adj = a.adj + rowbase * D.d_ld;
    const float* lse = a.lse + rowbase;

    v16f acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int qpiece = tid & 15, qrow = tid >> 4;  // Q: row = k (= i), 16 threads x float4 = 64 m (= j)
    const int bpiece = tid & 31, brow = tid >> 5;

    for (int i0 = 0; i0 < D.Bc; i0 += G_KC) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = qrow + 16 * it;
            const int i = i0 + row;
            const int j = m0 + 4 * qpiece;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < D.Bc && j < D.N) {
                const float l = lse[i];
                const float4 s = *reinterpret_cast<const float4*>(S + (int64_t)i * D.n_ld + j);
                v.x = dscore_any<L2>(D, s.x, l);
                if (j + 1 < D.N) v.y = dscore_any<L2>(D, s.y, l);
                if (j + 2 < D.N) v.z = dscore_any<L2>(D, s.z, l);
                if (j + 3 < D.N) v.w = dscore_any<L2>(D, s.w, l);
            }
            *reinterpret_cast<float4*>(Qs + row * G_TMS + 4 * qpiece) = v;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = brow + 8 * it;
            const int i = i0 + row;
            const int nl = 4 * bpiece;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < D.Bc) {
                if (nl < a.ncols) {
                    const int lim = min(D.d, n0 + a.ncols);
                    v = load_row4(adj + (int64_t)i * D.d_ld, n0 + nl, lim, (n0 & 3) ? 1 : 4);
                }
                if (L2 && nl == G_TN - 4) v.w = 1.f;
            }
            *reinterpret_cast<float4*>(Bs + row * G_TNS + nl) = v;
        }
        __syncthreads();
        const float* ap = Qs + (2 * h) * G_TMS + wm * 32 + l31;
        const float* bp = Bs + (2 * h) * G_TNS + wn * 64 + l31;
#pragma unroll 4
        for (int q = 0; q < G_KC / 4; ++q) {
            const float a0 = ap[(4 * q) * G_TMS], a1 = ap[(4 * q + 1) * G_TMS];
            const float b00 = bp[(4 * q) * G_TNS], b01 = bp[(4 * q) * G_TNS + 32];
            const float b10 = bp[(4 * q + 1) * G_TNS], b11 = bp[(4 * q + 1) * G_TNS + 32];
            acc[0] = mfma32(a0, b00, acc[0]);
            acc[1] = mfma32(a0, b01, acc[1]);
            acc[0] = mfma32(a1, b10, acc[0]);
            acc[1] = mfma32(a1, b11, acc[1]);
        }
        __syncthreads();
    }

    if (L2) {
        if (wn == 1 && l31 == 31) {
#pragma unroll
            for (int r = 0; r < 16; ++r) csum[wm * 32 + acc_row(r, h)] = acc[1][r];
        }
        __syncthreads();
    }
    const int64_t* negmap = a.negmap[dir] + (int64_t)c * D.N;
    float* out = a.gocc + (a.negocc_off[dir] + (int64_t)c * D.N) * D.d_ld;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int nl = wn * 64 + tn * 32 + l31;
        const int n = n0 + nl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 + acc_row(r, h);
            const int m = m0 + ml;
            if (m < D.N && nl < a.ncols && n < D.d) {
                float v = acc[tn][r];
                if (L2) v -= a.emb[negmap[m] * a.emb_ld + n] * csum[ml];
                out[(int64_t)m * D.d_ld + n] = v;
            }
        }
    }
}

// =========================================================================================== edge backward
struct EdgeBwdArgs {
    const float* emb;
    int64_t emb_ld;
    const int64_t* edges;
    const float* rel[2];
    int64_t rel_ld;
    const float* adj;
    const float* pos;
    const float* dpos;  // [ndir][Bp] dL/dpos (written by marius_lp_loss)
    const float* dadj;
    // flash path, fused form (dadj2 != nullptr): dadj / dadj2 hold the UNNORMALISED partials of a row's (at most two) contributors, `part` their
    // statistics (mref, sum V) [2][ndir Bp]: dL/dadj = g (exp(m0 - lse) O0 + exp(m1 - lse) O1).  keep_dadj: store that back into dadj.
    const float* dadj2;
    const float* pw;   // [2][ndir Bp] the contributors' weights g exp(m_k - lse) (flash_merge_kernel, float64)
    const float2* part;
    const float* lse;
    float gscale;
    int keep_dadj;
    float* gocc;     // rows [0,B) = src occurrences, [B,2B) = dst occurrences
    float* grel[2];  // [B, d_ld]
    // marius_lp_desc.upd_*: endpoint occurrences flagged in occ_single take their Adagrad step here (emb is then the table itself) and leave gocc unwritten
    const uint8_t* occ_single;
    float* upd_table;
    float* upd_state;
    float* upd_absmax;
    float upd_lr, upd_eps;
    LpDims D;
};

// the sparse Adagrad step of one lane's four elements of a row (batch.cpp:67-69 op order): the arithmetic of segreduce.hip's update kernels, bit for bit
__device__ __forceinline__ float edge_adagrad4(float* w_row, float* s_row, int c0, int c1, const float (&w)[4], const float (&g)[4], float lr, float eps) {
#pragma clang fp contract(off)
    const float2 s0 = *reinterpret_cast<const float2*>(s_row + c0), s1 = *reinterpret_cast<const float2*>(s_row + c1);
    const float sv[4] = {s0.x, s0.y, s1.x, s1.y};
    float sn[4], wn[4], seen = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ds = g[k] * g[k];
        sn[k] = sv[k] + ds;
        const float dw = -lr * (g[k] / (sqrtf(sn[k]) + eps));
        wn[k] = w[k] + dw;
        seen = fmaxf(seen, fabsf(wn[k]));
    }
    *reinterpret_cast<float2*>(s_row + c0) = make_float2(sn[0], sn[1]);
    *reinterpret_cast<float2*>(s_row + c1) = make_float2(sn[2], sn[3]);
    *reinterpret_cast<float2*>(w_row + c0) = make_float2(wn[0], wn[1]);
    *reinterpret_cast<float2*>(w_row + c1) = make_float2(wn[2], wn[3]);
    return seen;
}

// gradient wrt (e, r) of a = op(e, r) given g = dL/da, for column c (complex: c < h handles the pair (c, c + h))
// one wave per edge; handles both directions so every gocc element is written exactly once.
__global__ __launch_bounds__(256) void lp_edge_bwd_kernel(EdgeBwdArgs a) {
    const LpDims& D = a.D;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= D.B) return;
    const int64_t* ed = a.edges + i * D.edge_cols;
    const int64_t s = ed[0], t = ed[D.edge_cols - 1];
    const float* es = a.emb + s * a.emb_ld;
    const float* et = a.emb + t * a.emb_ld;
    float* gs = a.gocc + i * D.d_ld;
    float* gt = a.gocc + (D.B + i) * D.d_ld;
    const bool cplx = (D.relop == MARIUS_OP_COMPLEX_HADAMARD);
    const int h = D.d / 2;
    const int span = cplx ? h : D.d;  // complex: each lane-iteration handles columns c and c + h

    float coef[2], posv[2];
    for (int dir = 0; dir < D.ndir; ++dir) {
        coef[dir] = a.dpos[(int64_t)dir * D.Bp + i];  // dL/dpos
        posv[dir] = a.pos[(int64_t)dir * D.Bp + i];
    }

    for (int c = lane; c < span; c += 64) {
        float out_s[2] = {0.f, 0.f}, out_t[2] = {0.f, 0.f};  // [0]: column c, [1]: column c + h (complex)
        for (int dir = 0; dir < D.ndir; ++dir) {
            const float* e = dir == 0 ? es : et;   // operand that went through the relation operator
            const float* o = dir == 0 ? et : es;   // the other endpoint
            const bool has_rel = (D.edge_cols == 3) && (a.rel[dir] != nullptr);
            const float* r = has_rel ? a.rel[dir] + ed[1] * a.rel_ld : nullptr;
            const int64_t rowoff = ((int64_t)dir * D.Bp + i) * D.d_ld;
            const int ncol = cplx ? 2 : 1;
            float ga[2], go[2];
            for (int u = 0; u < ncol; ++u) {
                const int cc = c + u * h;
                const float av = a.adj[rowoff + cc];
                const float ov = o[cc];
                float g = a.dadj[rowoff + cc];
                if (a.dadj2) {
                    const int64_t row = (int64_t)dir * D.Bp + i, prows = (int64_t)D.ndir * D.Bp;
                    const float2 p1 = a.part[prows + row];
                    g *= a.pw[row];
                    if (p1.y > 0.f) g += a.pw[prows + row] * a.dadj2[rowoff + cc];
                    if (a.keep_dadj) const_cast<float*>(a.dadj)[rowoff + cc] = g;
                }
                if (D.cmp == MARIUS_CMP_L2) {
                    const float df = (av - ov) + 1e-6f;
                    const float w = (posv[dir] > 0.f) ? coef[dir] * df / posv[dir] : 0.f;
                    g += w;
                    go[u] = -w;
                } else {
                    g += coef[dir] * ov;
                    go[u] = coef[dir] * av;
                }
                ga[u] = g;
            }
            float ge[2] = {ga[0], ga[1]}, gr[2] = {0.f, 0.f};
            if (has_rel) {
                if (D.relop == MARIUS_OP_HADAMARD) {
                    ge[0] = ga[0] * r[c];
                    gr[0] = ga[0] * e[c];
                } else if (D.relop == MARIUS_OP_TRANSLATION) {
                    gr[0] = ga[0];
                } else if (cplx) {
                    const float er = e[c], ei = e[c + h], rr = r[c], ri = r[c + h];
                    ge[0] = ga[0] * rr + ga[1] * ri;
                    ge[1] = -ga[0] * ri + ga[1] * rr;
                    gr[0] = ga[0] * er + ga[1] * ei;
                    gr[1] = -ga[0] * ei + ga[1] * er;
                }
                for (int u = 0; u < ncol; ++u) a.grel[dir][i * D.d_ld + c + u * h] = gr[u];
            }
            for (int u = 0; u < ncol; ++u) {
                if (dir == 0) {
                    out_s[u] += ge[u];
                    out_t[u] += go[u];
                } else {
                    out_t[u] += ge[u];
                    out_s[u] += go[u];
                }
            }
        }
        const int ncol = cplx ? 2 : 1;
        for (int u = 0; u < ncol; ++u) {
            gs[c + u * h] = out_s[u];
            gt[c + u * h] = out_t[u];
        }
    }
}

// Vectorised edge backward: half a wave per edge, lane l owns the element pairs {2l, 2l+1} and {d/2 + 2l, d/2 + 2l + 1} (ComplEx:
// the (re, im) partners), every operand arrives as two independent 8-B loads.  Same arithmetic as lp_edge_bwd_kernel.
__global__ __launch_bounds__(256) void lp_edge_bwd2_kernel(EdgeBwdArgs a) {
    const LpDims& D = a.D;
    const int l = threadIdx.x & 31;
    const int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (i >= D.B || 4 * l >= D.d) return;
    const int c0 = 2 * l, c1 = D.d / 2 + 2 * l;
    const int64_t* ed = a.edges + i * D.edge_cols;
    const int64_t s = ed[0], t = ed[D.edge_cols - 1];
    auto ld4 = [&](const float* row, float (&v)[4]) {
        const float2 p = *reinterpret_cast<const float2*>(row + c0), q = *reinterpret_cast<const float2*>(row + c1);
        v[0] = p.x; v[1] = p.y; v[2] = q.x; v[3] = q.y;
    };
    auto st4 = [&](float* row, const float (&v)[4]) {
        *reinterpret_cast<float2*>(row + c0) = make_float2(v[0], v[1]);
        *reinterpret_cast<float2*>(row + c1) = make_float2(v[2], v[3]);
    };
    float x[2][4], av[2][4], dv[2][4], r[2][4];
    ld4(a.emb + s * a.emb_ld, x[0]);
    ld4(a.emb + t * a.emb_ld, x[1]);
    bool has_rel[2];
    float coef[2], posv[2];
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        has_rel[dir] = false;
        coef[dir] = posv[dir] = 0.f;
        if (dir >= D.ndir) continue;
        const int64_t rowoff = ((int64_t)dir * D.Bp + i) * D.d_ld;
        ld4(a.dadj + rowoff, dv[dir]);
        if (a.dadj2) {  // fused flash sweep: combine the contributors' unnormalised partials
            const int64_t row = (int64_t)dir * D.Bp + i, prows = (int64_t)D.ndir * D.Bp;
            const float2 p1 = a.part[prows + row];
            const float c0_ = a.pw[row];
#pragma unroll
            for (int k = 0; k < 4; ++k) dv[dir][k] *= c0_;
            if (p1.y > 0.f) {
                const float c1_ = a.pw[prows + row];
                float o1[4];
                ld4(a.dadj2 + rowoff, o1);
#pragma unroll
                for (int k = 0; k < 4; ++k) dv[dir][k] += c1_ * o1[k];
            }
            if (a.keep_dadj) st4(const_cast<float*>(a.dadj) + rowoff, dv[dir]);
        }
        has_rel[dir] = (D.edge_cols == 3) && (a.rel[dir] != nullptr);
        if (has_rel[dir]) ld4(a.rel[dir] + ed[1] * a.rel_ld, r[dir]);
        else r[dir][0] = r[dir][1] = r[dir][2] = r[dir][3] = 0.f;
        if (a.adj) ld4(a.adj + rowoff, av[dir]);
        else relop4(D.relop, has_rel[dir], x[dir], r[dir], av[dir]);  // bit for bit what the forward packed into the operand records
        coef[dir] = a.dpos[(int64_t)dir * D.Bp + i];  // dL/dpos
        posv[dir] = a.pos[(int64_t)dir * D.Bp + i];
    }
    float out_s[4] = {0.f, 0.f, 0.f, 0.f}, out_t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        if (dir >= D.ndir) continue;
        const float* e = x[dir];       // operand that went through the relation operator
        const float* o = x[dir ^ 1];   // the other endpoint
        float ga[4], go[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float g = dv[dir][k];
            if (D.cmp == MARIUS_CMP_L2) {
                const float df = (av[dir][k] - o[k]) + 1e-6f;
                const float w = (posv[dir] > 0.f) ? coef[dir] * df / posv[dir] : 0.f;
                g += w;
                go[k] = -w;
            } else {
                g += coef[dir] * o[k];
                go[k] = coef[dir] * av[dir][k];
            }
            ga[k] = g;
        }
        float ge[4] = {ga[0], ga[1], ga[2], ga[3]}, gr[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_rel[dir]) {
            if (D.relop == MARIUS_OP_HADAMARD) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ge[k] = ga[k] * r[dir][k];
                    gr[k] = ga[k] * e[k];
                }
            } else if (D.relop == MARIUS_OP_TRANSLATION) {
#pragma unroll
                for (int k = 0; k < 4; ++k) gr[k] = ga[k];
            } else if (D.relop == MARIUS_OP_COMPLEX_HADAMARD) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float er = e[k], ei = e[k + 2], rr = r[dir][k], ri = r[dir][k + 2];
                    ge[k] = ga[k] * rr + ga[k + 2] * ri;
                    ge[k + 2] = -ga[k] * ri + ga[k + 2] * rr;
                    gr[k] = ga[k] * er + ga[k + 2] * ei;
                    gr[k + 2] = -ga[k] * ei + ga[k + 2] * er;
                }
            }
            st4(a.grel[dir] + i * D.d_ld, gr);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dir == 0) {
                out_s[k] += ge[k];
                out_t[k] += go[k];
            } else {
                out_t[k] += ge[k];
                out_s[k] += go[k];
            }
        }
    }
    if (a.occ_single) {
        // a node that occurs once in the batch: its row (x[]), its whole gradient (out_*) and nothing else of it are right here — take the Adagrad
        // step now instead of a 400-byte store that the segment update reads back next to the same table row (its rows skip these: fused_below)
        float seen = 0.f;
        if (a.occ_single[i]) seen = edge_adagrad4(a.upd_table + s * a.emb_ld, a.upd_state + s * a.emb_ld, c0, c1, x[0], out_s, a.upd_lr, a.upd_eps);
        else st4(a.gocc + i * D.d_ld, out_s);
        if (a.occ_single[D.B + i]) seen = fmaxf(seen, edge_adagrad4(a.upd_table + t * a.emb_ld, a.upd_state + t * a.emb_ld, c0, c1, x[1], out_t, a.upd_lr, a.upd_eps));
        else st4(a.gocc + (D.B + i) * D.d_ld, out_t);
        // magnitude bound (marius_lp_desc.absmax): monotone, so the racy read filters almost every lane out once it has settled
        if (a.upd_absmax && seen > *reinterpret_cast<volatile float*>(a.upd_absmax)) atomicMax(reinterpret_cast<unsigned int*>(a.upd_absmax), __float_as_uint(seen));
        return;
    }
    st4(a.gocc + i * D.d_ld, out_s);
    st4(a.gocc + (D.B + i) * D.d_ld, out_t);
}

__global__ __launch_bounds__(256) void lp_zero_rows_kernel(float* p, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) p[t] = 0.f;
}

// =========================================================================================== ranks
__global__ __launch_bounds__(256) void lp_ranks_kernel(const float* __restrict__ pos, const float* __restrict__ neg, int64_t rows,
                                                       int N, int64_t neg_ld, int64_t* __restrict__ ranks) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float p = pos[row];
    const float* s = neg + row * neg_ld;
    int cnt = 0;
    for (int c = lane; c < N; c += 64) cnt += (s[c] >= p) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) ranks[row] = (int64_t)cnt + 1;
}

// =========================================================================================== host side
static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// Contraction kernel family: 3 = ping-pong persistent scores (experimental, MARIUS_KERNELS=pp; measured slower than 2),
// 2 = resident-operand scores + merged 16x16x4 grads (lp_res.hip, default),
// 1 = fast (lp_fast.hip), 0 = generic.
// MARIUS_KERNELS=generic|fast|res (or MARIUS_NO_FAST=1) selects a lower level for A/B runs and for tests of every code path;
// a level that does not apply to the shape falls through to the next lower one.
static int kernel_level();
static unsigned long long* g_dbg_timeline = nullptr;
// SoftmaxCE partial (max, sum exp) per (row, negative-tile group) come out of the score kernel's epilogue when nothing can change
// the scores afterwards (no score filter) and the resident-operand kernel runs; marius_lp_loss then only merges the partials.
// which level-2 score kernel runs: 'p' (adj fragments in registers, persistent workgroups: d / 4 in {8, 16, 25, 32}) or 'r' (adj tile in LDS,
// persistent workgroups: every other d <= 128 with d % 4 == 0; MARIUS_SCORES=res forces it where 'p' would apply — tests of that kernel at
// the bench shape).  The one-workgroup-per-unit forms and the bf16x6 split planes of rounds 1-2 lost their A/B runs and are gone.
static char scores_variant(const marius_lp_desc* d, const LpDims& D) {
    const bool want_res = kernel_env().scores == 'r';
    if (!want_res && scores_a_applicable(d->emb, d->emb_ld, D.d)) return 'p';
    if (scores_res_applicable(d->emb, d->emb_ld, D.d)) return 'r';
    return 0;
}
// number of column groups of the fused SoftmaxCE partials, 0 when the loss must be computed from the materialised scores
static int lse_fused_groups(const marius_lp_desc* d, const LpDims& D) {
    if (kernel_level() != 2) return 0;
    if (D.loss != MARIUS_LOSS_SOFTMAX_CE) return 0;  // the other losses read the materialised scores
    if ((d->dst_filter && d->n_dst_filter > 0) || (d->src_filter && d->n_src_filter > 0)) return 0;
    int ntpg, ng;
    const char v = scores_variant(d, D);
    if (v == 'p') ng = scores_ap_groups(D.N);
    else if (v == 'r') scores_res_geometry(D.N, ntpg, ng);
    else return 0;
    return ng;
}
static bool lse_fused(const marius_lp_desc* d, const LpDims& D) { return lse_fused_groups(d, D) > 0; }
static int kernel_level() {
    const KernelEnv& e = kernel_env();
    if (e.no_fast || e.kernels == 'g') return 0;
    if (e.kernels == 'f') return 1;
    return 2;
}

static int fill_dims(const marius_lp_desc* d, LpDims& D) {
    MARIUS_REQUIRE(d, "lp: null descriptor");
    MARIUS_REQUIRE(d->d > 0 && d->B > 0 && d->C > 0 && d->N > 0, "lp: bad sizes d=%d B=%ld C=%d N=%d", d->d, (long)d->B, d->C, d->N);
    MARIUS_REQUIRE(d->edge_cols == 2 || d->edge_cols == 3, "lp: Edge list must be a 3 or 2 column tensor");
    MARIUS_REQUIRE(d->cmp >= 0 && d->cmp <= 2 && d->relop >= 0 && d->relop <= 3, "lp: bad relop/cmp");
    MARIUS_REQUIRE(!(d->relop == MARIUS_OP_COMPLEX_HADAMARD && (d->d & 1)), "lp: ComplEx needs an even embedding dim");
    MARIUS_REQUIRE(!(d->use_inverse && (d->edge_cols != 3 || !d->inv_rel)), "lp: inverse relations need 3-column edges and inv_rel");
    MARIUS_REQUIRE(!(d->use_inverse && !d->src_neg), "lp: inverse direction needs src negatives");
    D.B = d->B;
    D.C = d->C;
    D.N = d->N;
    D.d = d->d;
    // comparators.cpp:9: (int)ceil((float)num_pos / num_chunks)
    D.Bc = (int)ceilf((float)d->B / (float)d->C);
    D.Bp = (int64_t)D.Bc * D.C;
    D.ndir = d->use_inverse ? 2 : 1;
    D.edge_cols = d->edge_cols;
    D.relop = d->relop;
    D.cmp = (d->cmp == MARIUS_CMP_COSINE) ? MARIUS_CMP_DOT : d->cmp;  // CosineCompare scores un-normalised tensors (comparators.cpp:43-60)
    D.n_ld = (d->N + 3) / 4 * 4;
    D.d_ld = (d->d + 3) / 4 * 4;
    MARIUS_REQUIRE(d->loss >= MARIUS_LOSS_SOFTMAX_CE && d->loss <= MARIUS_LOSS_SOFTPLUS, "lp: unknown loss type %d", d->loss);
    D.loss = (d->loss == MARIUS_LOSS_CROSS_ENTROPY) ? MARIUS_LOSS_SOFTMAX_CE : d->loss;  // same function of the same 1 + N scores
    D.margin = d->margin;
    {   // MEAN divides by the number of loss terms: rows (SoftmaxCE), Bp x N (Ranking: neg vs pos pairs), Bp x (1 + N) (elementwise losses)
        double terms = (double)D.Bp;
        if (D.loss == MARIUS_LOSS_RANKING) terms *= (double)D.N;
        else if (D.loss != MARIUS_LOSS_SOFTMAX_CE) terms *= (double)(D.N + 1);
        D.gscale = (d->reduction == MARIUS_REDUCE_MEAN) ? (float)(1.0 / terms) : 1.f;
    }
    // Bc * (C - 1) >= B leaves whole chunks of padding rows (a short last batch): legal, pad_and_reshape pads to Bc * C rows
    return MARIUS_OK;
}

// Losses whose dL/dS is non-negative (Ranking: 0 / g; the sigmoid family: g sigma(s)) reach the MFMA-tuned backward kernels through a
// second score-shaped buffer: marius_lp_loss writes log(dL/dS / g) per element (-inf for 0) next to the loss terms it computes anyway,
// and the backward runs on that buffer with lse = 0, i.e. exactly its SoftmaxCE form g exp(S' - 0).  Not for MSE (dL/dS changes sign)
// and not for the L2 comparator (its chain rule divides by the score itself).
static bool vlog_path(const LpDims& D) {
    if (kernel_env().no_vlog) return false;
    return D.loss != MARIUS_LOSS_SOFTMAX_CE && D.loss != MARIUS_LOSS_MSE && D.cmp == MARIUS_CMP_DOT && kernel_level() == 2;
}

static bool flash_store_scores(const marius_lp_desc* d) { return (d->flags & MARIUS_LP_STORE_SCORES) != 0; }

// what of the environment the flash buffers of a layout were sized for: (column chunks << 8) | folded tail | valid bit
static int flash_cfg_now(int d) { return 0x10000 | (flash_chunks(d) << 8) | (flash_tail4(d) ? 1 : 0); }
#define MARIUS_REQUIRE_FLASH_CFG(L, D, who)                                                                                                         \
    MARIUS_REQUIRE(!(L)->flash || (L)->flash_cfg == flash_cfg_now((D).d),                                                                            \
                   who ": the flash record layout changed between marius_lp_plan and this launch (MARIUS_FLASH_TAIL4 / MARIUS_FLASH_WIDE reloaded?): plan again")

static int make_layout(const marius_lp_desc* d, const LpDims& D, marius_lp_layout* L) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align256(bytes);
        return o;
    };
    L->Bp = D.Bp;
    L->n_ld = D.n_ld;
    L->d_ld = D.d_ld;
    const size_t rows = (size_t)D.Bp;
    for (int dir = 0; dir < 2; ++dir) L->adj[dir] = L->pos[dir] = L->neg[dir] = L->lse[dir] = L->rowloss[dir] = L->dadj[dir] = L->grel[dir] = 0;
    // per-dir arrays are contiguous [ndir][...] so kernels can index by dir
    size_t base = take(rows * D.d_ld * 4 * D.ndir);
    for (int dir = 0; dir < D.ndir; ++dir) L->adj[dir] = base + (size_t)dir * rows * D.d_ld * 4;
    base = take(rows * 4 * D.ndir);
    for (int dir = 0; dir < D.ndir; ++dir) L->pos[dir] = base + (size_t)dir * rows * 4;
    const bool flash = kernel_level() == 2 && flash_applicable(d, D);
    L->flash = flash ? 1 : 0;
    L->flash_cfg = flash ? flash_cfg_now(D.d) : 0;
    if (!flash || flash_store_scores(d) || flash_chunked(D.d)) {  // the flash path never materialises the scores — except for rows wider than 128 columns
        size_t sbytes = rows * D.n_ld * 4 * D.ndir;
        if (flash && flash_chunked(D.d) && flash_tiled_scores_bytes(D) > sbytes) sbytes = flash_tiled_scores_bytes(D);  // tile order needs whole tiles
        base = take(sbytes);
        for (int dir = 0; dir < D.ndir; ++dir) L->neg[dir] = base + (size_t)dir * rows * D.n_ld * 4;
    }
    base = take(rows * 4 * D.ndir);
    for (int dir = 0; dir < D.ndir; ++dir) L->lse[dir] = base + (size_t)dir * rows * 4;
    base = take(rows * 4 * D.ndir);
    for (int dir = 0; dir < D.ndir; ++dir) L->rowloss[dir] = base + (size_t)dir * rows * 4;
    L->loss = take(16);
    // flash path, fused form: a second [ndir][Bp, d_ld] block right behind the first holds the dAdj partial of a tile's second contributor
    base = take(rows * D.d_ld * 4 * D.ndir * (flash ? 2 : 1));
    for (int dir = 0; dir < D.ndir; ++dir) L->dadj[dir] = base + (size_t)dir * rows * D.d_ld * 4;
    const size_t nocc = (size_t)2 * D.B + (size_t)(d->src_neg ? 2 : 1) * D.C * D.N;
    L->gocc = take(nocc * D.d_ld * 4);
    base = take((size_t)D.B * D.d_ld * 4 * D.ndir);
    for (int dir = 0; dir < D.ndir; ++dir) L->grel[dir] = base + (size_t)dir * D.B * D.d_ld * 4;
    L->aux = take((rows + (size_t)D.C * D.N) * 4 * D.ndir);
    {
        const size_t ng = (size_t)(D.N + 63) / 64 + 1;  // upper bound over the score-kernel variants (finest: one partial per 64 columns)
        L->lsepart = take(rows * ng * 2 * 4 * D.ndir);
    }
    L->adjrec = L->negrec = L->fpart = 0;
    if (flash) {
        L->adjrec = take(flash_adjrec_bytes(D));
        L->negrec = take(flash_negrec_bytes(D));
        L->fpart = take(flash_part_bytes(D));
    }
    base = take(rows * 4 * D.ndir);
    L->dpos[0] = L->dpos[1] = 0;
    for (int dir = 0; dir < D.ndir; ++dir) L->dpos[dir] = base + (size_t)dir * rows * 4;
    L->vlog = (vlog_path(D) && !flash) ? take(rows * D.n_ld * 4 * D.ndir) : 0;
    L->total_bytes = off;
    return MARIUS_OK;
}

}  // namespace marius

using namespace marius;

extern "C" int marius_lp_plan(const marius_lp_desc* desc, marius_lp_layout* layout) {
    LpDims D;
    int rc = fill_dims(desc, D);
    if (rc) return rc;
    MARIUS_REQUIRE(layout, "lp_plan: null layout");
    return make_layout(desc, D, layout);
}

// the half-wave-per-edge prep / edge-backward kernels need 8-B aligned rows and d <= 128 (everything else: the any-shape kernels)
static bool lp_vec_ok(const marius_lp_desc* desc, const LpDims& D) {
    return (D.d % 4 == 0) && D.d <= 128 && (desc->emb_ld % 2 == 0) && (desc->rel_ld % 2 == 0 || !desc->rel) && (D.d == D.d_ld) &&
           ((reinterpret_cast<uintptr_t>(desc->emb) & 7) == 0) && ((reinterpret_cast<uintptr_t>(desc->rel) & 7) == 0) &&
           ((reinterpret_cast<uintptr_t>(desc->inv_rel) & 7) == 0);
}
// Flash path with the fused prep: adj exists as operand records only; the edge backward recomputes the four elements it needs from the rows
// it reads anyway (saves the 2 Bp d 4-byte store and its re-read).
static bool flash_adj_elided(const marius_lp_desc* desc, const LpDims& D, const marius_lp_layout* L) {
    return L->flash && lp_vec_ok(desc, D) && !(desc->flags & MARIUS_LP_STORE_SCORES);
}

// Endpoint singletons take their Adagrad step inside the edge backward (marius_lp_desc.upd_*): only the 16-byte-row kernel does that, only when
// training, and the state rows must be as aligned as the table's.
extern "C" int marius_lp_fuses_endpoint_update(const marius_lp_desc* desc) {
    if (!desc || !desc->upd_occ_single || !desc->upd_state || !(desc->flags & MARIUS_LP_TRAIN_ONLY) || (desc->flags & MARIUS_LP_KEEP_DADJ)) return 0;
    LpDims D;
    if (fill_dims(desc, D) != MARIUS_OK) return 0;
    return (lp_vec_ok(desc, D) && (reinterpret_cast<uintptr_t>(desc->upd_state) & 7) == 0) ? 1 : 0;
}

extern "C" int marius_lp_forward(const marius_lp_desc* desc, const marius_lp_layout* L, void* workspace, marius_stream_t stream) {
    LpDims D;
    int rc = fill_dims(desc, D);
    if (rc) return rc;
    MARIUS_REQUIRE(L && workspace && desc->emb && desc->edges && desc->dst_neg, "lp_forward: null pointer");
    MARIUS_REQUIRE(desc->emb_ld >= desc->d, "lp_forward: emb_ld < d");
    MARIUS_REQUIRE(desc->edge_cols == 2 || desc->rel, "lp_forward: relations missing for 3-column edges");
    MARIUS_REQUIRE_FLASH_CFG(L, D, "lp_forward");
    hipStream_t st = as_stream(stream);
    char* ws = (char*)workspace;
    const bool l2 = (D.cmp == MARIUS_CMP_L2);
    float* x2 = l2 ? (float*)(ws + L->aux) : nullptr;
    float* y2 = l2 ? x2 + (size_t)D.Bp * D.ndir : nullptr;

    const bool adj_elided = flash_adj_elided(desc, D, L);
    PrepArgs pa;
    pa.emb = desc->emb;
    pa.emb_ld = desc->emb_ld;
    pa.edges = desc->edges;
    pa.rel[0] = desc->rel;
    pa.rel[1] = desc->inv_rel;
    pa.rel_ld = desc->rel_ld;
    pa.adj = adj_elided ? nullptr : (float*)(ws + L->adj[0]);
    pa.pos = (float*)(ws + L->pos[0]);
    pa.x2 = x2;
    pa.D = D;
    pa.frec = nullptr;
    pa.fKP = pa.fXR = pa.fP = pa.fTail = 0;
    pa.fdadj = nullptr;
    pa.frg = FlRange{nullptr, nullptr, FL_ADJ_NODE};
    bool flash_fused_prep = false;
    {
        ProfScope ps(PROF_LP_PREP, st);
        const bool vec_ok = lp_vec_ok(desc, D);
        int64_t prep_rows = D.Bp;
        if (vec_ok && L->flash) {  // the adj records and the dadj zero fill ride along (lp_flash.hip)
            pa.fKP = (D.d + 15) / 16 * 16;
            pa.fXR = (D.Bc + 31) / 32 * 32;
            pa.fTail = flash_tail4(D.d) ? 1 : 0;
            pa.fP = pa.fTail ? 4 * pa.fKP - 16 : 4 * pa.fKP + 16;
            pa.frec = ws + L->adjrec;
            pa.frg = flash_range(desc, D);
            pa.fdadj = flash_fused() ? nullptr : (float*)(ws + L->dadj[0]);  // fused form: partials are stored, never accumulated
            prep_rows += (int64_t)D.ndir * D.C * (pa.fXR - D.Bc);  // one half-wave per chunk-padding record
            flash_fused_prep = true;
        }
        if (vec_ok)
            lp_prep2_kernel<MARIUS_PREP_EDGES_PER_HALF_WAVE><<<dim3((unsigned)cdiv(prep_rows, 8 * MARIUS_PREP_EDGES_PER_HALF_WAVE)), dim3(256), 0, st>>>(pa);
        else
            lp_prep_kernel<<<dim3((unsigned)cdiv(D.Bp * D.ndir, 4)), dim3(256), 0, st>>>(pa);
    }
    rc = check_launch("lp_prep");
    if (rc) return rc;
    if (l2) {
        const int64_t CN = (int64_t)D.C * D.N;
        lp_negnorm_kernel<<<dim3((unsigned)cdiv(CN * D.ndir, 4)), dim3(256), 0, st>>>(desc->emb, desc->emb_ld, desc->dst_neg,
                                                                                     desc->src_neg, CN, D.d, D.ndir, y2);
        rc = check_launch("lp_negnorm");
        if (rc) return rc;
    }

    if (L->flash) {  // training-only path: operand records + SoftmaxCE row statistics, no score tensor (lp_flash.hip)
        MARIUS_REQUIRE(kernel_level() == 2 && flash_applicable(desc, D), "lp_forward: the layout was planned for the flash path but the descriptor / environment no longer selects it");
        float* S = ((flash_store_scores(desc) || flash_chunked(D.d)) && L->neg[0]) ? (float*)(ws + L->neg[0]) : nullptr;
        const int64_t CNf = (int64_t)D.C * D.N;
        const int64_t occ_off[2] = {2 * D.B + (desc->src_neg ? CNf : 0), 2 * D.B};  // gocc rows of the dst / src negatives (map_tensors order)
        float* dadj0 = (float*)(ws + L->dadj[0]);
        return flash_forward(desc, D, pa.adj, ws + L->adjrec, ws + L->negrec, (float2*)(ws + L->fpart), S, flash_fused_prep, (float*)(ws + L->gocc), occ_off,
                             (flash_fused_prep || flash_fused()) ? nullptr : dadj0, pa.pos, dadj0, dadj0 + (size_t)D.ndir * D.Bp * D.d_ld, st);
    }

    ScoreArgs sa;
    sa.adj = pa.adj;
    sa.emb = desc->emb;
    sa.emb_ld = desc->emb_ld;
    sa.emb_vec = row_vec_width(desc->emb, desc->emb_ld, 4);
    sa.negmap[0] = desc->dst_neg;
    sa.negmap[1] = desc->src_neg;
    sa.S = (float*)(ws + L->neg[0]);
    sa.x2 = x2;
    sa.y2 = y2;
    sa.dk = (int)D.d_ld;
    sa.nkc = (sa.dk + 55) / 56;  // KC <= 56 keeps (128 + 128) * KS * 4 B under the 64 KiB dynamic-LDS default
    sa.KC = ((sa.dk + sa.nkc - 1) / sa.nkc + 3) / 4 * 4;
    sa.KS = ((sa.KC / 2) & 1) ? sa.KC : sa.KC + 2;  // stride/2 odd -> conflict-free ds_read_b64 across 32 rows
    sa.D = D;
    sa.ablate = 0;
    sa.lse_part = lse_fused(desc, D) ? (float*)(ws + L->lsepart) : nullptr;
    sa.dbg = kernel_env().timeline_grads ? nullptr : g_dbg_timeline;
    dim3 grid((unsigned)cdiv(D.N, F_TN), (unsigned)cdiv(D.Bc, F_TM), (unsigned)(D.C * D.ndir));
    size_t lds = (size_t)(F_TM + F_TN) * sa.KS * sizeof(float);
    {
        ProfScope ps(PROF_LP_SCORES, st);
        const int lvl = kernel_level();
        if (!((lvl == 2 && scores_variant(desc, D) == 'p' && launch_scores_ap(sa, l2, st)) ||
              (lvl >= 2 && launch_scores_res(sa, l2, st)) || (lvl >= 1 && launch_scores_fast(sa, l2, st)))) {
            if (l2)
                lp_scores_kernel<true><<<grid, dim3(256), lds, st>>>(sa);
            else
                lp_scores_kernel<false><<<grid, dim3(256), lds, st>>>(sa);
        }
    }
    rc = check_launch("lp_scores");
    if (rc) return rc;

    const int64_t* filt[2] = {desc->dst_filter, desc->src_filter};
    const int64_t nfilt[2] = {desc->n_dst_filter, desc->n_src_filter};
    for (int dir = 0; dir < D.ndir; ++dir) {
        if (filt[dir] && nfilt[dir] > 0) {
            int64_t blocks = cdiv(nfilt[dir], 256);
            if (blocks > 1024) blocks = 1024;
            lp_filter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((float*)(ws + L->neg[dir]), D.n_ld, D.Bp, D.N, filt[dir],
                                                                          nfilt[dir]);
            rc = check_launch("lp_filter");
            if (rc) return rc;
        }
    }
    return MARIUS_OK;
}

extern "C" int marius_lp_loss(const marius_lp_desc* desc, const marius_lp_layout* L, void* workspace, marius_stream_t stream) {
    LpDims D;
    int rc = fill_dims(desc, D);
    if (rc) return rc;
    MARIUS_REQUIRE(L && workspace, "lp_loss: null pointer");
    MARIUS_REQUIRE_FLASH_CFG(L, D, "lp_loss");
    hipStream_t st = as_stream(stream);
    char* ws = (char*)workspace;
    const int64_t rows = D.Bp * D.ndir;
    {
        ProfScope ps(PROF_LP_LSE, st);
        float* dpos = (float*)(ws + L->dpos[0]);
        if (L->flash) {
            const int64_t bpd = cdiv(D.Bp, 256);
            float* blocksum = (float*)(ws + L->aux);
            rc = flash_merge(D, (const float2*)(ws + L->fpart), (const float*)(ws + L->pos[0]), (float*)(ws + L->lse[0]), (float*)(ws + L->rowloss[0]),
                             dpos, blocksum, ws + L->adjrec, desc->absmax != nullptr && !kernel_env().flash_f16_off, st);
            if (rc) return rc;
            lp_loss_reduce_blocks_kernel<<<dim3(1), dim3(256), 0, st>>>(blocksum, bpd, D.ndir, D.gscale, (float*)(ws + L->loss));
            return check_launch("lp_loss_reduce");
        } else if (D.loss != MARIUS_LOSS_SOFTMAX_CE) {
            lp_loss_terms_kernel<<<dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st>>>((const float*)(ws + L->neg[0]), D.n_ld,
                                                                                     (const float*)(ws + L->pos[0]), rows, D.N, D.loss, D.margin,
                                                                                     D.gscale, (float*)(ws + L->lse[0]),
                                                                                     (float*)(ws + L->rowloss[0]), dpos,
                                                                                     (vlog_path(D) && L->vlog) ? (float*)(ws + L->vlog) : nullptr);
        } else if (lse_fused(desc, D)) {
            const int ng = lse_fused_groups(desc, D);
            const int64_t bpd = cdiv(D.Bp, 256);
            float* blocksum = (float*)(ws + L->aux);  // aux holds row norms only for the L2 comparator's forward; free again by now
            lp_lse_merge_kernel<<<dim3((unsigned)(bpd * D.ndir)), dim3(256), 0, st>>>((const float*)(ws + L->lsepart), ng,
                                                                                     (const float*)(ws + L->pos[0]), rows,
                                                                                     (float*)(ws + L->lse[0]), (float*)(ws + L->rowloss[0]), dpos,
                                                                                     D.gscale, D.Bp, blocksum);
            rc = check_launch("lp_lse_merge");
            if (rc) return rc;
            lp_loss_reduce_blocks_kernel<<<dim3(1), dim3(256), 0, st>>>(blocksum, bpd, D.ndir, D.gscale, (float*)(ws + L->loss));
            return check_launch("lp_loss_reduce");
        } else {
            lp_lse_kernel<<<dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st>>>((const float*)(ws + L->neg[0]), D.n_ld,
                                                                              (const float*)(ws + L->pos[0]), rows, D.N,
                                                                              (float*)(ws + L->lse[0]), (float*)(ws + L->rowloss[0]), dpos, D.gscale);
        }
    }
    rc = check_launch("lp_lse");
    if (rc) return rc;
    lp_loss_reduce_kernel<<<dim3(1), dim3(1024), 0, st>>>((const float*)(ws + L->rowloss[0]), D.Bp, D.ndir, D.gscale, (float*)(ws + L->loss));
    return check_launch("lp_loss_reduce");
}

extern "C" int marius_lp_backward(const marius_lp_desc* desc, const marius_lp_layout* L, void* workspace, marius_stream_t stream) {
    LpDims D;
    int rc = fill_dims(desc, D);
    if (rc) return rc;
    MARIUS_REQUIRE(L && workspace, "lp_backward: null pointer");
    MARIUS_REQUIRE_FLASH_CFG(L, D, "lp_backward");
    hipStream_t st = as_stream(stream);
    char* ws = (char*)workspace;
    const bool l2 = (D.cmp == MARIUS_CMP_L2);
    const int64_t CN = (int64_t)D.C * D.N;

    GradArgs ga;
    ga.S = (const float*)(ws + L->neg[0]);
    ga.lse = (const float*)(ws + L->lse[0]);
    ga.adj = (const float*)(ws + L->adj[0]);
    ga.emb = desc->emb;
    ga.emb_ld = desc->emb_ld;
    ga.emb_vec = row_vec_width(desc->emb, desc->emb_ld, 4);
    ga.negmap[0] = desc->dst_neg;
    ga.negmap[1] = desc->src_neg;
    ga.dadj = (float*)(ws + L->dadj[0]);
    ga.gocc = (float*)(ws + L->gocc);
    const bool has_src_neg = desc->src_neg != nullptr;
    ga.negocc_off[0] = 2 * D.B + (has_src_neg ? CN : 0);  // dst negatives come last in map_tensors order
    ga.negocc_off[1] = 2 * D.B;                            // src negatives
    ga.ncols = l2 ? G_TN - 1 : G_TN;
    ga.D = D;
    const bool vlog = vlog_path(D) && L->vlog;
    if (vlog) {  // SoftmaxCE form on the log-gradient buffer (see vlog_path)
        ga.S = (const float*)(ws + L->vlog);
        ga.D.loss = MARIUS_LOSS_SOFTMAX_CE;
    }
    ga.ablate = 0;
    ga.dbg = kernel_env().timeline_grads ? g_dbg_timeline : nullptr;
    const unsigned nblk = (unsigned)cdiv(D.d, ga.ncols);
    dim3 ga_grid(nblk, (unsigned)cdiv(D.Bc, G_TM), (unsigned)(D.C * D.ndir));
    dim3 gn_grid(nblk, (unsigned)cdiv(D.N, G_TM), (unsigned)(D.C * D.ndir));
    {
        // the tuned contraction kernels hard-code V = exp(S - lse); every other loss runs the generic kernels (dscore_any)
        const int lvl = (D.loss == MARIUS_LOSS_SOFTMAX_CE || vlog) ? kernel_level() : 0;
        const bool split = false;  // (the merged dAdj + dNeg launch; the separate launches below serve shapes it does not take)
        bool done = false;
        if (L->flash) {
            const bool filtered = (desc->dst_filter && desc->n_dst_filter > 0) || (D.ndir == 2 && desc->src_filter && desc->n_src_filter > 0);
            rc = flash_backward(desc, D, ws + L->adjrec, ws + L->negrec, ga.dadj, ga.gocc, ga.negocc_off, (const float2*)(ws + L->fpart), filtered,
                                flash_chunked(D.d) ? (float*)(ws + L->neg[0]) : nullptr, st);
            if (rc) return rc;
            done = true;
        }
        if (!done && lvl >= 2 && !split) {
            ProfScope ps(PROF_LP_GRAD_ADJ, st);  // merged launch is accounted under lp_grad_adj (both contractions)
            done = launch_grad16(ga, l2, 0, st);
        }
        if (!done) {
            {
                ProfScope ps(PROF_LP_GRAD_ADJ, st);
                if (!((lvl >= 2 && launch_grad16(ga, l2, 1, st)) || (lvl >= 1 && launch_grad_adj_fast(ga, l2, st)))) {
                    if (l2)
                        lp_grad_adj_kernel<true><<<ga_grid, dim3(256), 0, st>>>(ga);
                    else
                        lp_grad_adj_kernel<false><<<ga_grid, dim3(256), 0, st>>>(ga);
                }
            }
            {
                ProfScope ps(PROF_LP_GRAD_NEG, st);
                if (!((lvl >= 2 && launch_grad16(ga, l2, 2, st)) || (lvl >= 1 && launch_grad_neg_fast(ga, l2, st)))) {
                    if (l2)
                        lp_grad_neg_kernel<true><<<gn_grid, dim3(256), 0, st>>>(ga);
                    else
                        lp_grad_neg_kernel<false><<<gn_grid, dim3(256), 0, st>>>(ga);
                }
            }
        }
    }
    rc = check_launch("lp_grad");
    if (rc) return rc;
    if (has_src_neg && D.ndir == 1) {  // src negatives take part in the unique map but receive no gradient
        const int64_t n = CN * D.d_ld;
        int64_t blocks = cdiv(n, 256);
        if (blocks > 4096) blocks = 4096;
        lp_zero_rows_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(ga.gocc + 2 * D.B * D.d_ld, n);
        rc = check_launch("lp_zero_rows");
        if (rc) return rc;
    }

    EdgeBwdArgs ea;
    ea.emb = desc->emb;
    ea.emb_ld = desc->emb_ld;
    ea.edges = desc->edges;
    ea.rel[0] = desc->rel;
    ea.rel[1] = desc->inv_rel;
    ea.rel_ld = desc->rel_ld;
    ea.adj = flash_adj_elided(desc, D, L) ? nullptr : ga.adj;
    ea.pos = (const float*)(ws + L->pos[0]);
    ea.dpos = (const float*)(ws + L->dpos[0]);
    ea.dadj = ga.dadj;
    ea.dadj2 = nullptr;
    ea.part = nullptr;
    ea.pw = nullptr;
    ea.lse = nullptr;
    ea.gscale = D.gscale;
    ea.keep_dadj = 0;
    if (L->flash && flash_fused() && !flash_chunked(D.d)) {  // (rows wider than 128 take the stored-score launches: dadj is final there)
        ea.dadj2 = ga.dadj + (size_t)D.ndir * D.Bp * D.d_ld;
        ea.part = (const float2*)(ws + L->fpart);
        ea.pw = flash_part_weights(D, ea.part);
        ea.lse = (const float*)(ws + L->lse[0]);
        ea.keep_dadj = (desc->flags & MARIUS_LP_KEEP_DADJ) ? 1 : 0;
    }
    ea.gocc = ga.gocc;
    ea.occ_single = nullptr;
    ea.upd_table = ea.upd_state = ea.upd_absmax = nullptr;
    ea.upd_lr = ea.upd_eps = 0.f;
    if (marius_lp_fuses_endpoint_update(desc)) {
        ea.occ_single = desc->upd_occ_single;
        ea.upd_table = const_cast<float*>(desc->emb);  // documented in marius_hip.h: with upd_* set, emb is the caller's (mutable) node table
        ea.upd_state = desc->upd_state;
        ea.upd_absmax = desc->upd_absmax;
        ea.upd_lr = desc->upd_lr;
        ea.upd_eps = desc->upd_eps;
    }
    ea.grel[0] = (float*)(ws + L->grel[0]);
    ea.grel[1] = D.ndir == 2 ? (float*)(ws + L->grel[1]) : nullptr;
    ea.D = D;
    {
        ProfScope ps(PROF_LP_EDGE_BWD, st);
        const bool vec_ok = lp_vec_ok(desc, D);
        if (vec_ok)
            lp_edge_bwd2_kernel<<<dim3((unsigned)cdiv(D.B, 8)), dim3(256), 0, st>>>(ea);
        else
            lp_edge_bwd_kernel<<<dim3((unsigned)cdiv(D.B, 4)), dim3(256), 0, st>>>(ea);
    }
    return check_launch("lp_edge_bwd");
}

extern "C" int marius_loss_scores(int32_t loss_type, float margin, const float* pos, const float* neg, int64_t rows, int32_t N, int64_t neg_ld,
                                  int32_t reduction, float* scratch, float* loss, marius_stream_t stream) {
    MARIUS_REQUIRE(rows >= 0 && N >= 0 && neg_ld >= N, "loss_scores: bad sizes");
    MARIUS_REQUIRE(loss_type >= MARIUS_LOSS_SOFTMAX_CE && loss_type <= MARIUS_LOSS_SOFTPLUS, "loss_scores: unknown loss type %d", loss_type);
    MARIUS_REQUIRE(loss && (rows == 0 || (pos && neg && scratch)), "loss_scores: null pointer");
    hipStream_t st = as_stream(stream);
    const int lt = loss_type == MARIUS_LOSS_CROSS_ENTROPY ? MARIUS_LOSS_SOFTMAX_CE : loss_type;
    MARIUS_REQUIRE(neg_ld % 4 == 0, "loss_scores: the row pitch of neg must be a multiple of 4 floats (16-B row accesses)");
    double terms = (double)rows;
    if (lt == MARIUS_LOSS_RANKING) terms *= (double)N;
    else if (lt != MARIUS_LOSS_SOFTMAX_CE) terms *= (double)(N + 1);
    const float scale = (reduction == MARIUS_REDUCE_MEAN && terms > 0) ? (float)(1.0 / terms) : 1.f;
    float* rowloss = scratch;
    float* rowval = scratch + rows;
    if (rows > 0) {
        if (lt == MARIUS_LOSS_SOFTMAX_CE)
            lp_lse_kernel<<<dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st>>>(neg, neg_ld, pos, rows, N, rowval, rowloss, nullptr, scale);
        else
            lp_loss_terms_kernel<<<dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st>>>(neg, neg_ld, pos, rows, N, lt, margin, scale, rowval, rowloss, nullptr, nullptr);
        int rc = check_launch("loss_scores");
        if (rc) return rc;
    }
    lp_loss_reduce_kernel<<<dim3(1), dim3(1024), 0, st>>>(rowloss, rows, 1, scale, loss);
    return check_launch("loss_scores_reduce");
}

extern "C" int marius_compute_ranks(const float* pos, const float* neg, int64_t rows, int32_t N, int64_t neg_ld, int64_t* ranks,
                                    marius_stream_t stream) {
    MARIUS_REQUIRE(rows >= 0 && N >= 0 && neg_ld >= N, "compute_ranks: bad sizes");
    if (rows == 0) return MARIUS_OK;
    MARIUS_REQUIRE(pos && neg && ranks, "compute_ranks: null pointer");
    lp_ranks_kernel<<<dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, as_stream(stream)>>>(pos, neg, rows, N, neg_ld, ranks);
    return check_launch("compute_ranks");
}

// SoftmaxCrossEntropy on materialised scores (loss.cpp:50-67): lse[i] = log(e^pos_i + sum_j e^neg_ij), rowloss = lse - pos,
// loss[0] = sum (SUM) or mean (MEAN) of rowloss.  Scratch: lse[rows], rowloss[rows], loss[4].
extern "C" int marius_softmax_ce(const float* pos, const float* neg, int64_t rows, int32_t N, int64_t neg_ld, int32_t reduction, float* lse,
                                 float* rowloss, float* loss, marius_stream_t stream) {
    MARIUS_REQUIRE(rows > 0 && N > 0 && neg_ld >= N && (neg_ld % 4 == 0 || N < 4), "softmax_ce: bad sizes (neg_ld must be a multiple of 4)");
    MARIUS_REQUIRE(pos && neg && lse && rowloss && loss, "softmax_ce: null pointer");
    hipStream_t st = as_stream(stream);
    const int n_eff = (neg_ld % 4 == 0) ? N : 0;
    (void)n_eff;
    lp_lse_kernel<<<dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st>>>(neg, neg_ld, pos, rows, N, lse, rowloss, nullptr, 1.f);
    lp_loss_reduce_kernel<<<dim3(1), dim3(1024), 0, st>>>(rowloss, rows, 1, reduction == MARIUS_REDUCE_MEAN ? 1.f / (float)rows : 1.f, loss);
    return check_launch("softmax_ce");
}

// debug only: device buffer (>= 256 * 2 * 64 u64) receiving per-phase cycle stamps of the score kernel; NULL disables
extern "C" int marius_debug_set_timeline(unsigned long long* buf) {
    g_dbg_timeline = buf;
    return MARIUS_OK;
}
