// "Resident-operand" variants of the three FP32-MFMA contractions (the ones the Freebase86m d=100 workload runs).
//
// Same preconditions as lp_fast.hip (d % 4 == 0, 16-B aligned embedding rows) plus d <= 128 for the score kernel.
//   * lp_scores_res_kernel: a workgroup keeps its 64-row adj tile resident in LDS for the whole K = d and streams 64-row
//     negative tiles through a double-buffered LDS ring: one barrier per 64x64 output tile, the global loads of tile
//     t+2 (and the negative-row indices of tile t+3) are in flight while tile t is multiplied.  v_mfma_f32_32x32x2_f32.
//   * lp_grad_adj16_kernel / lp_grad_neg16_kernel: output tile 64 rows x (16-column MFMA tiles covering d), i.e. d=100 is
//     padded to 112 instead of 128 (v_mfma_f32_16x16x4_f32), K streamed in chunks of 32 through a double-buffered LDS
//     ring (one barrier per chunk), V = dL/dS recomputed from S while staging.
// All three use the XCD-aware block decoding of lp_fast.hip.  No atomics; every output element has one owner.
#include "lp_common.h"

namespace marius {

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ bool decode_block2(int lin, int units_per_cd, int ncd, int& cd, int& unit) {
    const int xcd = lin & 7, slot = lin >> 3;
    cd = (slot / units_per_cd) * 8 + xcd;
    unit = slot - (slot / units_per_cd) * units_per_cd;
    return cd < ncd;
}
static inline unsigned xcd_grid2(int units_per_cd, int ncd) { return (unsigned)(((ncd + 7) / 8) * 8 * units_per_cd); }

__device__ __forceinline__ void lds_store4x(float* p, const float4& v) {  // 8-B aligned destination
    *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y);
    *reinterpret_cast<float2*>(p + 2) = make_float2(v.z, v.w);
}
__device__ __forceinline__ float4 mul4(const float4& v, float m) { return make_float4(v.x * m, v.y * m, v.z * m, v.w * m); }

// =========================================================================================== scores, adj tile resident
constexpr int R_T = 64;  // output tile 64 x 64 per iteration; 4 waves as 2 x 2, one 32x32 MFMA tile each

// NQ = d / 4 as a compile-time constant (fully unrolled, software-pipelined MFMA chain) or 0 for a runtime K loop
template <bool L2, int NQ>
__global__ __launch_bounds__(256, 2) void lp_scores_res_kernel(ScoreArgs a, int ngroups, int nt_per_group, int units_per_cd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LpDims& D = a.D;
    int cd, unit;
    if (!decode_block2(blockIdx.x, units_per_cd, D.C * D.ndir, cd, unit)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int dir = cd / D.C, c = cd - dir * D.C;
    const int mt = unit / ngroups, ng = unit - mt * ngroups;
    const int m0 = mt * R_T;
    const int ntiles = (D.N + R_T - 1) / R_T;
    const int nt0 = ng * nt_per_group;
    const int T = min(nt_per_group, ntiles - nt0);
    if (T <= 0) return;
    const int KS = a.KS;  // d + 2
    float* As = smem;
    float* Bs0 = smem + R_T * KS;
    float* Bs1 = Bs0 + R_T * KS;
    const float* adj = a.adj + ((int64_t)dir * D.Bp + (int64_t)c * D.Bc) * D.d_ld;
    const int64_t* negmap = a.negmap[dir] + (int64_t)c * D.N;

    const int piece = tid & 31, row = tid >> 5;  // 32 threads x 16 B cover up to 128 columns; 8 rows per pass, 8 passes
    const bool col_ok = 4 * piece < D.d;
    const int colc = col_ok ? 4 * piece : 0;

    float4 vb[8];
    int64_t ids[8];
    auto load_ids = [&](int t) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int n = (nt0 + t) * R_T + row + 8 * it;
            ids[it] = negmap[n < D.N ? n : 0];
        }
    };
    auto issue_b = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it) vb[it] = *reinterpret_cast<const float4*>(a.emb + ids[it] * a.emb_ld + colc);
    };
    auto write_b = [&](float* buf, int t) {
        if (col_ok) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int n = (nt0 + t) * R_T + row + 8 * it;
                lds_store4x(buf + (row + 8 * it) * KS + 4 * piece, mul4(vb[it], n < D.N ? 1.f : 0.f));
            }
        }
    };

    // ---- prologue: adj tile + negative tile 0 into LDS, tile 1 in flight
    load_ids(0);
    {
        float4 va[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m0 + row + 8 * it;
            va[it] = *reinterpret_cast<const float4*>(adj + (int64_t)(m < D.Bc ? m : 0) * D.d_ld + colc);
        }
        issue_b();
        if (col_ok) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = m0 + row + 8 * it;
                lds_store4x(As + (row + 8 * it) * KS + 4 * piece, mul4(va[it], m < D.Bc ? 1.f : 0.f));
            }
        }
    }
    write_b(Bs0, 0);
    if (T > 1) {
        load_ids(1);
        issue_b();
    }
    if (T > 2) load_ids(2);
    __syncthreads();

    float* S = a.S + ((int64_t)dir * D.Bp + (int64_t)c * D.Bc) * D.n_ld;
    const float* ap = As + (wm * 32 + l31) * KS + 2 * h;
    const int nq = D.d >> 2;
    float run_m = -3.0e38f, run_l = 0.f;  // running (max, sum exp) of this lane's row over the unit's columns
    for (int t = 0; t < T; ++t) {
        const float* bp = ((t & 1) ? Bs1 : Bs0) + (wn * 32 + l31) * KS + 2 * h;
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (NQ > 0) {
            // operands of step q+1 are read from LDS before the two MFMAs of step q issue (sched_barrier pins that order: left to
            // itself the compiler re-uses one register pair and waits lgkmcnt(0) before every MFMA pair, exposing the LDS latency)
            float2 av[2], bv[2];
            av[0] = *reinterpret_cast<const float2*>(ap);
            bv[0] = *reinterpret_cast<const float2*>(bp);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
                    av[(q + 1) & 1] = *reinterpret_cast<const float2*>(ap + 4 * (q + 1));
                    bv[(q + 1) & 1] = *reinterpret_cast<const float2*>(bp + 4 * (q + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma32(bv[q & 1].x, av[q & 1].x, acc);  // swapped: D[n][m], a lane owns ONE row m and 16 columns n
                acc = mfma32(bv[q & 1].y, av[q & 1].y, acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int q = 0; q < nq; ++q) {
                const float2 a2 = *reinterpret_cast<const float2*>(ap + 4 * q);
                const float2 b2 = *reinterpret_cast<const float2*>(bp + 4 * q);
                acc = mfma32(b2.x, a2.x, acc);
                acc = mfma32(b2.y, a2.y, acc);
            }
        }
        // tile t+1 (in registers since the previous iteration) -> the other LDS buffer; then put tile t+2 in flight
        if (t + 1 < T) write_b((t & 1) ? Bs0 : Bs1, t + 1);
        if (t + 2 < T) {
            issue_b();  // uses ids of tile t+2
            if (t + 3 < T) load_ids(t + 3);
        }
        // epilogue of tile t: lane (m = l31, h) holds S[m][nb + 8q + 4h + e] in acc[4q + e]: four 16-B stores per lane, and the
        // row-wise (max, sum exp) of the SoftmaxCE is lane-local (no cross-lane traffic until the end of the unit)
        {
            const int m = m0 + wm * 32 + l31;
            const int nb = (nt0 + t) * R_T + wn * 32 + 4 * h;
            float xx = 0.f;
            if (L2 && m < D.Bc) xx = a.x2[(int64_t)dir * D.Bp + (int64_t)c * D.Bc + m];
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[r];
                if (L2) {
#pragma clang fp contract(off)
                    const int n = nb + 8 * (r >> 2) + (r & 3);
                    const float yy = (n < D.N) ? a.y2[(int64_t)dir * D.C * D.N + (int64_t)c * D.N + n] : 0.f;
                    const float tt = (xx + yy) - 2.f * v[r];
                    v[r] = sqrtf(fmaxf(tt, 1e-8f));
                }
            }
            if (m < D.Bc) {
                float* srow = S + (int64_t)m * D.n_ld;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + 8 * q;
                    if (n + 3 < D.N) {
                        *reinterpret_cast<float4*>(srow + n) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < D.N) srow[n + e] = v[4 * q + e];
                    }
                }
            }
            if (a.lse_part) {
                float tmax = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nb + 8 * (r >> 2) + (r & 3);
                    if (n < D.N) tmax = fmaxf(tmax, v[r]);
                }
                const float mnew = fmaxf(run_m, tmax);
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nb + 8 * (r >> 2) + (r & 3);
                    if (n < D.N) sum += __expf(v[r] - mnew);
                }
                run_l = run_l * __expf(run_m - mnew) + sum;
                run_m = mnew;
            }
        }
        __syncthreads();
    }
    if (a.lse_part) {
        // combine the two half-waves (columns +0..3 / +4..7 of every 8), then the two waves that cover the 64-column tile
        const float m2 = __shfl_xor(run_m, 32, 64), l2 = __shfl_xor(run_l, 32, 64);
        const float mm = fmaxf(run_m, m2);
        const float ll = run_l * __expf(run_m - mm) + l2 * __expf(m2 - mm);
        float* red = smem;  // LDS is free after the last barrier of the tile loop
        if (h == 0) {
            red[((wm * 2 + wn) * 32 + l31) * 2] = mm;
            red[((wm * 2 + wn) * 32 + l31) * 2 + 1] = ll;
        }
        __syncthreads();
        const int m = m0 + wm * 32 + l31;
        if (wn == 0 && h == 0 && m < D.Bc) {
            const float ma = red[((wm * 2) * 32 + l31) * 2], la = red[((wm * 2) * 32 + l31) * 2 + 1];
            const float mb = red[((wm * 2 + 1) * 32 + l31) * 2], lb = red[((wm * 2 + 1) * 32 + l31) * 2 + 1];
            const float mx = fmaxf(ma, mb);
            float* out = a.lse_part + ((((int64_t)dir * D.Bp + (int64_t)c * D.Bc + m) * ngroups) + ng) * 2;
            out[0] = mx;
            out[1] = la * __expf(ma - mx) + lb * __expf(mb - mx);
        }
    }
}

// =========================================================================================== backward contractions, 16x16x4 tiles
constexpr int H_TM = 64;          // output rows per workgroup (4 waves x 16)
constexpr int H_KC = 32;          // K chunk
constexpr int H_NT = 8;           // up to 8 x 16 = 128 output columns per n-block
constexpr int H_QS_MK = H_KC + 4; // V staged [m][k] (grad_adj): 16-B aligned rows
constexpr int H_QS_KM = H_TM + 4; // V staged [k][m] (grad_neg)
constexpr int H_BS = 128 + 4;     // B staged [k][n]
constexpr int H_QSZ = (H_TM * H_QS_MK > H_KC * H_QS_KM) ? H_TM * H_QS_MK : H_KC * H_QS_KM;
constexpr int H_MAXIDS = 2048;    // negative-row indices of one chunk kept in LDS as int32 (N <= 2048 on this path)
// dynamic LDS layout (floats): Qs[2][H_QSZ] | Bs[2][H_KC * H_BS] | sums[H_TM] | ids[H_MAXIDS] (grad_adj only, sized by N)
static inline size_t grad16_lds_bytes(int N) { return (size_t)(2 * H_QSZ + 2 * H_KC * H_BS + H_TM + ((N + 3) / 4) * 4) * sizeof(float); }

// Both backward contractions stream K in chunks of 32 through a double-buffered LDS ring with ONE barrier per chunk and
// a two-chunk-deep register prefetch (register sets 0/1 alternate, so every global load has two chunk periods to land).

// dAdj_c[m, n] = sum_j V[m, j] * Neg_c[j, n]
template <bool L2, int NT>
__device__ __forceinline__ void grad_adj16_body(const GradArgs& a, int cd, int unit, int tiles_m, float* smem) {
    float(*Qs)[H_QSZ] = reinterpret_cast<float(*)[H_QSZ]>(smem);
    float(*Bs)[H_KC * H_BS] = reinterpret_cast<float(*)[H_KC * H_BS]>(smem + 2 * H_QSZ);
    float* rsum = smem + 2 * H_QSZ + 2 * H_KC * H_BS;
    int* idl = reinterpret_cast<int*>(rsum + H_TM);
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int dir = cd / D.C, c = cd - dir * D.C;
    const int nb = unit / tiles_m;
    const int m0 = (unit - nb * tiles_m) * H_TM, n0 = nb * 128;
    const int64_t rowbase = (int64_t)dir * D.Bp + (int64_t)c * D.Bc;
    const float* S = a.S + rowbase * D.n_ld;
    const int64_t* negmap = a.negmap[dir] + (int64_t)c * D.N;

    for (int j = tid; j < D.N; j += 256) idl[j] = (int)negmap[j];  // batch-local row ids of this chunk's negatives

    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

    // staging roles: V: 8 threads x 16 B = 32 j per row, 32 rows per pass, 2 passes; B: 32 threads x 16 B, 8 rows per pass, 4 passes
    const int qpiece = tid & 7, qrow = tid >> 3;
    const int bpiece = tid & 31, brow = tid >> 5;
    const float* srow[2];
    float lse_r[2], qmask[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int m = m0 + qrow + 32 * it;
        const int mc = m < D.Bc ? m : 0;
        srow[it] = S + (int64_t)mc * D.n_ld;
        lse_r[it] = a.lse[rowbase + mc];
        qmask[it] = m < D.Bc ? 1.f : 0.f;
    }
    const int ncol = n0 + 4 * bpiece;
    const bool col_ok = ncol + 3 < D.d;
    const int ncol_c = col_ok ? ncol : 0;
    const int ones_col = L2 ? D.d - n0 : -1;  // local column of the ones column (L2, single n-block)
    const int nchunks = (D.N + H_KC - 1) / H_KC;
    __syncthreads();  // idl visible

    auto issue = [&](int ch, float4(&vs)[2], float4(&vb)[4]) {
        if (ch >= nchunks) return;
        const int j0 = ch * H_KC;
        const int j = j0 + 4 * qpiece;
#pragma unroll
        for (int it = 0; it < 2; ++it) vs[it] = *reinterpret_cast<const float4*>(srow[it] + (j < D.N ? j : 0));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int jj = j0 + brow + 8 * it;
            const int id = idl[jj < D.N ? jj : 0];
            vb[it] = *reinterpret_cast<const float4*>(a.emb + (int64_t)id * a.emb_ld + ncol_c);
        }
    };
    auto write = [&](int buf, int ch, const float4(&vs)[2], const float4(&vb)[4]) {
        if (ch >= nchunks) return;
        const int j0 = ch * H_KC;
        const int j = j0 + 4 * qpiece;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float4 v;
            v.x = (j < D.N) ? dscore<L2>(vs[it].x, lse_r[it], D.gscale) * qmask[it] : 0.f;
            v.y = (j + 1 < D.N) ? dscore<L2>(vs[it].y, lse_r[it], D.gscale) * qmask[it] : 0.f;
            v.z = (j + 2 < D.N) ? dscore<L2>(vs[it].z, lse_r[it], D.gscale) * qmask[it] : 0.f;
            v.w = (j + 3 < D.N) ? dscore<L2>(vs[it].w, lse_r[it], D.gscale) * qmask[it] : 0.f;
            *reinterpret_cast<float4*>(&Qs[buf][(qrow + 32 * it) * H_QS_MK + 4 * qpiece]) = v;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = brow + 8 * it;
            float4 v = vb[it];
            if (!col_ok || (j0 + r) >= D.N) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (L2 && (j0 + r) < D.N && (ones_col >> 2) == bpiece) {
                const int e = ones_col & 3;
                if (e == 0) v.x = 1.f; else if (e == 1) v.y = 1.f; else if (e == 2) v.z = 1.f; else v.w = 1.f;
            }
            *reinterpret_cast<float4*>(&Bs[buf][r * H_BS + 4 * bpiece]) = v;
        }
    };
    auto compute = [&](int buf) {
        const float* qp = &Qs[buf][(wave * 16 + l15) * H_QS_MK + 4 * kq];
        const float* bp = &Bs[buf][(4 * kq) * H_BS + l15];
        // 8 steps of NT MFMAs; the NT B values (and the A value) of step i+1 are read from LDS before the MFMAs of step i issue
        float bc[NT], bn[NT];
        float4 a4 = *reinterpret_cast<const float4*>(qp);
#pragma unroll
        for (int t = 0; t < NT; ++t) bc[t] = bp[16 * t];
#pragma unroll
        for (int i = 0; i < H_KC / 4; ++i) {
            const int s_ = i >> 2, e = i & 3;
            const float av = (e == 0) ? a4.x : (e == 1) ? a4.y : (e == 2) ? a4.z : a4.w;
            float4 a4n = a4;
            if (i + 1 < H_KC / 4) {
                const int sn = (i + 1) >> 2, en = (i + 1) & 3;
                const float* brow_p = bp + (16 * sn + en) * H_BS;
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = brow_p[16 * t];
                if (en == 0) a4n = *reinterpret_cast<const float4*>(qp + 16 * sn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma16(av, bc[t], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) bc[t] = bn[t];
            a4 = a4n;
            (void)s_;
        }
    };

    float4 vs0[2], vb0[4], vs1[2], vb1[4];
    issue(0, vs0, vb0);
    issue(1, vs1, vb1);
    write(0, 0, vs0, vb0);
    issue(2, vs0, vb0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
        compute(0);                       // chunk ch
        write(1, ch + 1, vs1, vb1);
        issue(ch + 3, vs1, vb1);
        __syncthreads();
        if (ch + 1 < nchunks) compute(1);  // chunk ch + 1
        write(0, ch + 2, vs0, vb0);
        issue(ch + 4, vs0, vb0);
        __syncthreads();
    }

    // lane holds D[m = 4 * kq + r][n = l15] of each 16x16 tile
    if (L2) {
        const int t1 = ones_col >> 4;
        if (l15 == (ones_col & 15)) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t == t1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rsum[wave * 16 + 4 * kq + r] = acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = a.dadj + rowbase * D.d_ld;
    const float* adj = a.adj + rowbase * D.d_ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 16 * t + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ml = wave * 16 + 4 * kq + r;
            const int m = m0 + ml;
            if (m < D.Bc && n < D.d) {
                float v = acc[t][r];
                if (L2) v -= adj[(int64_t)m * D.d_ld + n] * rsum[ml];
                out[(int64_t)m * D.d_ld + n] = v;
            }
        }
    }
}

// dNeg_c[m, n] = sum_i V[i, m] * adj_c[i, n]
template <bool L2, int NT>
__device__ __forceinline__ void grad_neg16_body(const GradArgs& a, int cd, int unit, int tiles_m, float* smem) {
    float(*Qs)[H_QSZ] = reinterpret_cast<float(*)[H_QSZ]>(smem);
    float(*Bs)[H_KC * H_BS] = reinterpret_cast<float(*)[H_KC * H_BS]>(smem + 2 * H_QSZ);
    float* csum = smem + 2 * H_QSZ + 2 * H_KC * H_BS;
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int dir = cd / D.C, c = cd - dir * D.C;
    const int nb = unit / tiles_m;
    const int m0 = (unit - nb * tiles_m) * H_TM, n0 = nb * 128;
    const int64_t rowbase = (int64_t)dir * D.Bp + (int64_t)c * D.Bc;
    const float* S = a.S + rowbase * D.n_ld;
    const float* adj = a.adj + rowbase * D.d_ld;
    const float* lse = a.lse + rowbase;

    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

    // V: rows = i (K), 16 threads x 16 B = 64 j (M), 16 rows per pass, 2 passes; B (adj rows): 32 thr x 16 B, 8 rows per pass, 4 passes
    const int qpiece = tid & 15, qrow = tid >> 4;
    const int bpiece = tid & 31, brow = tid >> 5;
    const int jq = m0 + 4 * qpiece;
    const int jq_c = jq < D.N ? jq : 0;
    const int ncol = n0 + 4 * bpiece;
    const bool col_ok = ncol + 3 < D.d_ld;
    const int ncol_c = col_ok ? ncol : 0;
    const int ones_col = L2 ? D.d - n0 : -1;
    const int nchunks = (D.Bc + H_KC - 1) / H_KC;

    auto issue = [&](int ch, float4(&vs)[2], float(&lv)[2], float4(&vb)[4]) {
        if (ch >= nchunks) return;
        const int i0 = ch * H_KC;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = i0 + qrow + 16 * it;
            const int ic = i < D.Bc ? i : 0;
            vs[it] = *reinterpret_cast<const float4*>(S + (int64_t)ic * D.n_ld + jq_c);
            lv[it] = lse[ic];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = i0 + brow + 8 * it;
            vb[it] = *reinterpret_cast<const float4*>(adj + (int64_t)(i < D.Bc ? i : 0) * D.d_ld + ncol_c);
        }
    };
    auto write = [&](int buf, int ch, const float4(&vs)[2], const float(&lv)[2], const float4(&vb)[4]) {
        if (ch >= nchunks) return;
        const int i0 = ch * H_KC;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = qrow + 16 * it;
            const bool ok = (i0 + r) < D.Bc;
            float4 v;
            v.x = (ok && jq < D.N) ? dscore<L2>(vs[it].x, lv[it], D.gscale) : 0.f;
            v.y = (ok && jq + 1 < D.N) ? dscore<L2>(vs[it].y, lv[it], D.gscale) : 0.f;
            v.z = (ok && jq + 2 < D.N) ? dscore<L2>(vs[it].z, lv[it], D.gscale) : 0.f;
            v.w = (ok && jq + 3 < D.N) ? dscore<L2>(vs[it].w, lv[it], D.gscale) : 0.f;
            *reinterpret_cast<float4*>(&Qs[buf][r * H_QS_KM + 4 * qpiece]) = v;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = brow + 8 * it;
            const bool ok = (i0 + r) < D.Bc;
            float4 v = vb[it];
            if (!col_ok || !ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (L2 && ok && (ones_col >> 2) == bpiece) {
                const int e = ones_col & 3;
                if (e == 0) v.x = 1.f; else if (e == 1) v.y = 1.f; else if (e == 2) v.z = 1.f; else v.w = 1.f;
            }
            *reinterpret_cast<float4*>(&Bs[buf][r * H_BS + 4 * bpiece]) = v;
        }
    };
    auto compute = [&](int buf) {
        const float* qp = &Qs[buf][(4 * kq) * H_QS_KM + wave * 16 + l15];
        const float* bp = &Bs[buf][(4 * kq) * H_BS + l15];
        float bc[NT], bn[NT];
        float ac = qp[0], an = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) bc[t] = bp[16 * t];
#pragma unroll
        for (int i = 0; i < H_KC / 4; ++i) {
            if (i + 1 < H_KC / 4) {
                const int sn = (i + 1) >> 2, en = (i + 1) & 3;
                const float* brow_p = bp + (16 * sn + en) * H_BS;
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = brow_p[16 * t];
                an = qp[(16 * sn + en) * H_QS_KM];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma16(ac, bc[t], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) bc[t] = bn[t];
            ac = an;
        }
    };

    float4 vs0[2], vb0[4], vs1[2], vb1[4];
    float lv0[2], lv1[2];
    issue(0, vs0, lv0, vb0);
    issue(1, vs1, lv1, vb1);
    write(0, 0, vs0, lv0, vb0);
    issue(2, vs0, lv0, vb0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
        compute(0);
        write(1, ch + 1, vs1, lv1, vb1);
        issue(ch + 3, vs1, lv1, vb1);
        __syncthreads();
        if (ch + 1 < nchunks) compute(1);
        write(0, ch + 2, vs0, lv0, vb0);
        issue(ch + 4, vs0, lv0, vb0);
        __syncthreads();
    }

    if (L2) {
        const int t1 = ones_col >> 4;
        if (l15 == (ones_col & 15)) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t == t1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) csum[wave * 16 + 4 * kq + r] = acc[t][r];
                }
        }
        __syncthreads();
    }
    const int64_t* negmap = a.negmap[dir] + (int64_t)c * D.N;
    float* out = a.gocc + (a.negocc_off[dir] + (int64_t)c * D.N) * D.d_ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 16 * t + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ml = wave * 16 + 4 * kq + r;
            const int m = m0 + ml;
            if (m < D.N && n < D.d) {
                float v = acc[t][r];
                if (L2) v -= a.emb[negmap[m] * a.emb_ld + n] * csum[ml];
                out[(int64_t)m * D.d_ld + n] = v;
            }
        }
    }
}

// One launch for both backward contractions: the unit list of a (chunk, dir) is [adj row-tiles..., neg row-tiles...].
// (Two separate launches of 1600 workgroups each over 768 resident slots run 3 rounds at 69 % fill; 3200 units run 5 at 83 %,
// and the two kinds of units have different load/MFMA phase patterns, which desynchronises the workgroups sharing a CU.)
// which: 0 = both, 1 = adj only, 2 = neg only.
template <bool L2, int NT>
__global__ __launch_bounds__(256, 3) void lp_grad16_kernel(GradArgs a, int tiles_adj, int units_adj, int tiles_neg, int units_neg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int cd, unit;
    if (!decode_block2(blockIdx.x, units_adj + units_neg, a.D.C * a.D.ndir, cd, unit)) return;
    if (unit < units_adj)
        grad_adj16_body<L2, NT>(a, cd, unit, tiles_adj, smem);
    else
        grad_neg16_body<L2, NT>(a, cd, unit - units_adj, tiles_neg, smem);
}

// =========================================================================================== launchers
static bool res_ok(const float* emb, int64_t emb_ld, int d) {
    return (d % 4 == 0) && (emb_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(emb) & 15) == 0);
}

bool scores_res_applicable(const float* emb, int64_t emb_ld, int d) { return res_ok(emb, emb_ld, d) && d <= 128; }

bool launch_scores_res(const ScoreArgs& a_in, bool l2, hipStream_t st) {
    if (!scores_res_applicable(a_in.emb, a_in.emb_ld, a_in.D.d)) return false;
    ScoreArgs a = a_in;
    a.KS = a.D.d + 2;  // (d + 2) / 2 odd for d % 4 == 0: conflict-free ds_read_b64 across 32 rows
    const int mtiles = (int)cdiv(a.D.Bc, R_T);
    int nt_per_group, ngroups;  // 4 negative tiles per workgroup: fine-grained enough to balance 256 CUs
    scores_res_geometry(a.D.N, nt_per_group, ngroups);
    const int units = mtiles * ngroups;
    const size_t lds = (size_t)3 * R_T * a.KS * sizeof(float);
    dim3 grid(xcd_grid2(units, a.D.C * a.D.ndir));
#define SCORES_RES_LAUNCH(L2V, NQV)                                                                                                     \
    do {                                                                                                                                  \
        if (lds > 65536) hipFuncSetAttribute((const void*)lp_scores_res_kernel<L2V, NQV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        lp_scores_res_kernel<L2V, NQV><<<grid, dim3(256), lds, st>>>(a, ngroups, nt_per_group, units);                                      \
    } while (0)
#define SCORES_RES_DISPATCH(L2V)                          \
    do {                                                  \
        switch (a.D.d / 4) {                              \
            case 8: SCORES_RES_LAUNCH(L2V, 8); break;     \
            case 16: SCORES_RES_LAUNCH(L2V, 16); break;   \
            case 25: SCORES_RES_LAUNCH(L2V, 25); break;   \
            case 32: SCORES_RES_LAUNCH(L2V, 32); break;   \
            default: SCORES_RES_LAUNCH(L2V, 0); break;    \
        }                                                 \
    } while (0)
    if (l2)
        SCORES_RES_DISPATCH(true);
    else
        SCORES_RES_DISPATCH(false);
#undef SCORES_RES_DISPATCH
#undef SCORES_RES_LAUNCH
    return true;
}

static bool grad16_shape(const GradArgs& a, bool l2, int& nblk, int& nt) {
    if (!res_ok(a.emb, a.emb_ld, a.D.d) || a.D.N > H_MAXIDS) return false;
    const int cols = a.D.d + (l2 ? 1 : 0);
    if (l2 && cols > 128) return false;  // ones column must sit in the single n-block
    nblk = (int)cdiv(a.D.d, 128);
    nt = nblk == 1 ? (int)cdiv(cols, 16) : H_NT;
    return true;
}

// NT (16-column MFMA tiles per workgroup) is a template parameter; shapes round up to the next instantiated value
#define GRAD16_DISPATCH(L2V)                                                                                         \
    do {                                                                                                               \
        if (nt <= 1) lp_grad16_kernel<L2V, 1><<<grid, dim3(256), lds, st>>>(a, tiles_adj, units_adj, tiles_neg, units_neg);      \
        else if (nt <= 2) lp_grad16_kernel<L2V, 2><<<grid, dim3(256), lds, st>>>(a, tiles_adj, units_adj, tiles_neg, units_neg); \
        else if (nt <= 4) lp_grad16_kernel<L2V, 4><<<grid, dim3(256), lds, st>>>(a, tiles_adj, units_adj, tiles_neg, units_neg); \
        else if (nt <= 7) lp_grad16_kernel<L2V, 7><<<grid, dim3(256), lds, st>>>(a, tiles_adj, units_adj, tiles_neg, units_neg); \
        else lp_grad16_kernel<L2V, 8><<<grid, dim3(256), lds, st>>>(a, tiles_adj, units_adj, tiles_neg, units_neg);              \
    } while (0)

// which: 0 = both contractions in one launch, 1 = dAdj only, 2 = dNeg only
bool launch_grad16(const GradArgs& a, bool l2, int which, hipStream_t st) {
    int nblk, nt;
    if (!grad16_shape(a, l2, nblk, nt)) return false;
    const int tiles_adj = (int)cdiv(a.D.Bc, H_TM), tiles_neg = (int)cdiv(a.D.N, H_TM);
    const int units_adj = (which == 2) ? 0 : tiles_adj * nblk;
    const int units_neg = (which == 1) ? 0 : tiles_neg * nblk;
    const size_t lds = grad16_lds_bytes(a.D.N);
    dim3 grid(xcd_grid2(units_adj + units_neg, a.D.C * a.D.ndir));
    if (l2)
        GRAD16_DISPATCH(true);
    else
        GRAD16_DISPATCH(false);
    return true;
}
#undef GRAD16_DISPATCH

}  // namespace marius

// =========================================================================================== scores, ping-pong persistent
// 512-thread workgroup = two 4-wave sets that run the SAME program half a period apart: while one set issues the 50 MFMAs of
// its 64x64 tile, the other stores its previous tile and stages its next negative tile — each SIMD hosts one wave of each
// set, so its matrix pipe always has a computing wave.  (With two independent 256-thread workgroups per CU the two ran in
// lockstep: both staged, both multiplied, both stored together and the phases added up instead of overlapping — measured by
// ablation: full 0.30 ms = skeleton 0.07 + MFMA 0.16 + stores 0.05 + staging 0.04.)  Workgroups are persistent: each set
// walks a static list of (chunk, dir, row-tile, negative-tile-group) units that belong to its XCD, so the prologue latency is
// paid once per workgroup and the loads of the next step (also across unit seams) are always in flight behind the MFMAs.
namespace marius {

struct PPDesc {
    bool valid, new_unit;
    int dir, c, m0, ntile;
};

template <bool L2>
__global__ __launch_bounds__(512) void lp_scores_pp_kernel(ScoreArgs a, int ngroups, int ntpg, int units_per_cd, int wg_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LpDims& D = a.D;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;  // within the set
    const int set = threadIdx.x >> 8;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int KS = a.KS;
    float* As = smem + set * (2 * R_T * KS);
    float* Bs = As + R_T * KS;
    const int xcd = blockIdx.x & 7, wgx = blockIdx.x >> 3;
    const int W = 2 * wg_per_xcd;
    const int ncd = D.C * D.ndir;
    const int ncd_x = (ncd - xcd + 7) / 8;
    const int total_units = ncd_x * units_per_cd;
    auto units_of = [&](int w) { return w < total_units ? (total_units - w + W - 1) / W : 0; };
    const int w = wgx * 2 + set;
    const int my_units = units_of(w);
    const int K = max(units_of(wgx * 2), units_of(wgx * 2 + 1)) * ntpg;  // steps (both sets run the same number of phases)
    const int ntiles = (D.N + R_T - 1) / R_T;

    auto desc = [&](int k) {
        PPDesc d;
        const int ui = k / ntpg, ti = k - ui * ntpg;
        d.valid = (k >= 0) && (ui < my_units);
        const int g = w + ui * W;
        const int cdi = g / units_per_cd, unit = g - cdi * units_per_cd;
        const int cd = xcd + 8 * cdi;
        d.dir = cd / D.C;
        d.c = cd - d.dir * D.C;
        const int mt = unit / ngroups, ng = unit - mt * ngroups;
        d.m0 = mt * R_T;
        d.ntile = ng * ntpg + ti;
        d.valid = d.valid && (d.ntile < ntiles);
        d.new_unit = (ti == 0);
        return d;
    };

    const int piece = tid & 31, row = tid >> 5;
    const bool col_ok = 4 * piece < D.d;
    const int colc = col_ok ? 4 * piece : 0;
    float4 va[8], vb[8];
    int64_t ids[8];
    auto load_ids = [&](const PPDesc& d) {
        const int64_t* negmap = a.negmap[d.dir] + (int64_t)d.c * D.N;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int n = d.ntile * R_T + row + 8 * it;
            ids[it] = negmap[n < D.N ? n : 0];
        }
    };
    auto issue_b = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it) vb[it] = *reinterpret_cast<const float4*>(a.emb + ids[it] * a.emb_ld + colc);
    };
    auto issue_a = [&](const PPDesc& d) {
        const float* adj = a.adj + ((int64_t)d.dir * D.Bp + (int64_t)d.c * D.Bc) * D.d_ld;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = d.m0 + row + 8 * it;
            va[it] = *reinterpret_cast<const float4*>(adj + (int64_t)(m < D.Bc ? m : 0) * D.d_ld + colc);
        }
    };
    auto write_b = [&](const PPDesc& d) {
        if (col_ok) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int n = d.ntile * R_T + row + 8 * it;
                lds_store4x(Bs + (row + 8 * it) * KS + 4 * piece, mul4(vb[it], n < D.N ? 1.f : 0.f));
            }
        }
    };
    auto write_a = [&](const PPDesc& d) {
        if (col_ok) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = d.m0 + row + 8 * it;
                lds_store4x(As + (row + 8 * it) * KS + 4 * piece, mul4(va[it], m < D.Bc ? 1.f : 0.f));
            }
        }
    };

    // ---- prologue: step 0 staged, ids of step 1 loaded
    {
        const PPDesc d0 = desc(0), d1 = desc(1);
        if (d0.valid) {
            load_ids(d0);
            issue_a(d0);
            issue_b();
            write_a(d0);
            write_b(d0);
        }
        if (d1.valid) load_ids(d1);
    }
    __syncthreads();

    v16f acc;
    const float* ap = As + (wm * 32 + l31) * KS + 2 * h;
    const float* bp = Bs + (wn * 32 + l31) * KS + 2 * h;
    const int nq = D.d >> 2;
    const int P = 2 * K + 1;
    for (int p = 0; p < P; ++p) {
        if ((p & 1) == set) {
            // ---------------- compute phase of step k; first put step k+1's rows (and step k+2's ids) in flight
            const int k = (p - set) >> 1;
            const PPDesc dk = desc(k), dn = desc(k + 1), d2 = desc(k + 2);
            if (dn.valid) {
                issue_b();
                if (dn.new_unit) issue_a(dn);
            }
            if (d2.valid) load_ids(d2);
            if (dk.valid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 5
                for (int q = 0; q < nq; ++q) {
                    const float2 a2 = *reinterpret_cast<const float2*>(ap + 4 * q);
                    const float2 b2 = *reinterpret_cast<const float2*>(bp + 4 * q);
                    acc = mfma32(a2.x, b2.x, acc);
                    acc = mfma32(a2.y, b2.y, acc);
                }
            }
        } else if (p - 1 - set >= 0) {
            // ---------------- memory phase after step kk: store its tile, stage step kk+1 into this set's LDS
            const int kk = (p - 1 - set) >> 1;
            const PPDesc dk = desc(kk), dn = desc(kk + 1);
            if (dk.valid) {
                float* S = a.S + ((int64_t)dk.dir * D.Bp + (int64_t)dk.c * D.Bc) * D.n_ld;
                const int n = dk.ntile * R_T + wn * 32 + l31;
                float yy = 0.f;
                if (L2 && n < D.N) yy = a.y2[(int64_t)dk.dir * D.C * D.N + (int64_t)dk.c * D.N + n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = dk.m0 + wm * 32 + acc_row(r, h);
                    if (m < D.Bc && n < D.N) {
                        float v = acc[r];
                        if (L2) {
#pragma clang fp contract(off)
                            const float xx = a.x2[(int64_t)dk.dir * D.Bp + (int64_t)dk.c * D.Bc + m];
                            const float tt = (xx + yy) - 2.f * v;
                            v = sqrtf(fmaxf(tt, 1e-8f));
                        }
                        S[(int64_t)m * D.n_ld + n] = v;
                    }
                }
            }
            if (dn.valid) {
                write_b(dn);
                if (dn.new_unit) write_a(dn);
            }
        }
        __syncthreads();
    }
}

bool launch_scores_pp(const ScoreArgs& a_in, bool l2, hipStream_t st) {
    if (!res_ok(a_in.emb, a_in.emb_ld, a_in.D.d) || a_in.D.d > 128) return false;
    ScoreArgs a = a_in;
    a.KS = a.D.d + 2;
    const int mtiles = (int)cdiv(a.D.Bc, R_T), ntiles = (int)cdiv(a.D.N, R_T);
    const int ntpg = ntiles >= 8 ? 4 : ntiles;
    const int ngroups = (int)cdiv(ntiles, ntpg);
    const int units = mtiles * ngroups;
    const int ncd = a.D.C * a.D.ndir;
    // one persistent 512-thread workgroup per CU (32 per XCD) unless the problem is smaller than that
    const int units_x = (int)cdiv(ncd, 8) * units;
    int wg_per_xcd = 32;
    if (units_x < 64) wg_per_xcd = (int)cdiv(units_x, 2);
    const size_t lds = (size_t)4 * R_T * a.KS * sizeof(float);
    dim3 grid((unsigned)(8 * wg_per_xcd));
    if (l2) {
        hipFuncSetAttribute((const void*)lp_scores_pp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lp_scores_pp_kernel<true><<<grid, dim3(512), lds, st>>>(a, ngroups, ntpg, units, wg_per_xcd);
    } else {
        hipFuncSetAttribute((const void*)lp_scores_pp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lp_scores_pp_kernel<false><<<grid, dim3(512), lds, st>>>(a, ngroups, ntpg, units, wg_per_xcd);
    }
    return true;
}

}  // namespace marius
