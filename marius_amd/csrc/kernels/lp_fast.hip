// Fast variants of the three FP32-MFMA contractions of the decoder (see lp_decoder.hip for the generic ones).
//
// Preconditions (checked by the launchers, else the generic kernels run): d % 4 == 0 and 16-B aligned embedding rows.
// What is different from the generic kernels:
//   * staging is branch-free: every global load of a K-chunk is issued back to back (clamped addresses + zero masks),
//     one wait, then the LDS writes — the generic kernels' guarded loads serialise into one HBM/L2 round trip per row;
//   * register prefetch: the loads of chunk k+1 (and the negative-row indices of chunk k+2) are in flight while the
//     MFMAs of chunk k run;
//   * XCD-aware launch: a 1-D grid is decoded so that all workgroups of one (chunk, direction) run on the same XCD
//     (block b -> XCD b % 8), i.e. the chunk's adj / negative rows (2 x 400 kB at d=100) are fetched into ONE 4-MiB L2
//     instead of eight.  Placement is a speed matter only: every output element still has exactly one owner.
#include "lp_common.h"

namespace marius {

// (chunk, dir) index and tile index of linear block `lin`; false for the padding blocks of the last round of 8
__device__ __forceinline__ bool decode_block(int lin, int tiles_per_cd, int ncd, int& cd, int& tile) {
    const int xcd = lin & 7, slot = lin >> 3;
    cd = (slot / tiles_per_cd) * 8 + xcd;
    tile = slot - (slot / tiles_per_cd) * tiles_per_cd;
    return cd < ncd;
}
static inline unsigned xcd_grid(int tiles_per_cd, int ncd) { return (unsigned)(((ncd + 7) / 8) * 8 * tiles_per_cd); }

__device__ __forceinline__ void lds_store4(float* p, const float4& v) {  // 8-B aligned destination
    *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y);
    *reinterpret_cast<float2*>(p + 2) = make_float2(v.z, v.w);
}

// =========================================================================================== scores
template <bool L2>
__global__ __launch_bounds__(256) void lp_scores_fast_kernel(ScoreArgs a, int tiles_n, int tiles_per_cd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LpDims& D = a.D;
    int cd, tile;
    if (!decode_block(blockIdx.x, tiles_per_cd, D.C * D.ndir, cd, tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int dir = cd / D.C, c = cd - dir * D.C;
    const int m0 = (tile / tiles_n) * F_TM, n0 = (tile - (tile / tiles_n) * tiles_n) * F_TN;
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

    const int piece = tid & 15, rbase = tid >> 4;  // 16 threads x 16 B per row, 16 rows per pass, 8 passes
    const float* arow[8];
    const float* brow[8];
    float amask[8], bmask[8];
    {
        int64_t ids[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int n = n0 + rbase + 16 * it;
            ids[it] = negmap[n < D.N ? n : 0];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m0 + rbase + 16 * it, n = n0 + rbase + 16 * it;
            arow[it] = adj + (int64_t)(m < D.Bc ? m : 0) * D.d_ld;
            amask[it] = m < D.Bc ? 1.f : 0.f;
            brow[it] = a.emb + ids[it] * a.emb_ld;
            bmask[it] = n < D.N ? 1.f : 0.f;
        }
    }
    float4 va[8], vb[8];
    auto issue = [&](int kc) {
        const int k0 = kc * a.KC;
        const int kcur = min(a.KC, a.dk - k0);
        const int k = k0 + (4 * piece < kcur ? 4 * piece : 0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            va[it] = *reinterpret_cast<const float4*>(arow[it] + k);
            vb[it] = *reinterpret_cast<const float4*>(brow[it] + k);
        }
    };
    issue(0);
    for (int kc = 0; kc < a.nkc; ++kc) {
        const int kcur = min(a.KC, a.dk - kc * a.KC);
        if (4 * piece < kcur) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + 16 * it;
                const float ma = amask[it], mb = bmask[it];
                lds_store4(As + row * KS + 4 * piece, make_float4(va[it].x * ma, va[it].y * ma, va[it].z * ma, va[it].w * ma));
                lds_store4(Bs + row * KS + 4 * piece, make_float4(vb[it].x * mb, vb[it].y * mb, vb[it].z * mb, vb[it].w * mb));
            }
        }
        __syncthreads();
        if (kc + 1 < a.nkc) issue(kc + 1);
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
                        const float t = (xx + yy) - 2.f * v;
                        v = sqrtf(fmaxf(t, 1e-8f));
                    }
                    S[(int64_t)m * D.n_ld + n] = v;
                }
            }
        }
}

// =========================================================================================== dAdj = V . Neg
template <bool L2>
__global__ __launch_bounds__(256) void lp_grad_adj_fast_kernel(GradArgs a, int tiles_m, int tiles_per_cd) {
    __shared__ __attribute__((aligned(16))) float Qs[G_TM * G_KSA];
    __shared__ __attribute__((aligned(16))) float Bs[G_KC * G_TNS];
    __shared__ float rsum[G_TM];
    const LpDims& D = a.D;
    int cd, tile;
    if (!decode_block(blockIdx.x, tiles_per_cd, D.C * D.ndir, cd, tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int dir = cd / D.C, c = cd - dir * D.C;
    const int nb = tile / tiles_m;
    const int m0 = (tile - nb * tiles_m) * G_TM, n0 = nb * a.ncols;
    const int64_t rowbase = (int64_t)dir * D.Bp + (int64_t)c * D.Bc;
    const float* S = a.S + rowbase * D.n_ld;
    const int64_t* negmap = a.negmap[dir] + (int64_t)c * D.N;

    v16f acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int qpiece = tid & 15, qrow = tid >> 4;  // Q: 16 thr x 16 B = 64 j per row, 16 rows per pass, 4 passes
    const int bpiece = tid & 31, brow = tid >> 5;  // B: 32 thr x 16 B = 128 cols per row, 8 rows per pass, 8 passes
    const float* srow[4];
    float lse_r[4], qmask[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int m = m0 + qrow + 16 * it;
        const int mc = m < D.Bc ? m : 0;
        srow[it] = S + (int64_t)mc * D.n_ld;
        lse_r[it] = a.lse[rowbase + mc];
        qmask[it] = m < D.Bc ? 1.f : 0.f;
    }
    // B columns this thread stages (fixed over the K loop)
    const int nl = 4 * bpiece;
    const int ncol = n0 + nl;
    const bool col_ok = (nl < a.ncols) && (ncol + 3 < D.d);  // d % 4 == 0 and n0 % 4 == 0: a float4 is all-in or all-out
    const int ncol_c = col_ok ? ncol : 0;

    float4 vs[4], vbv[8];
    int64_t ids_next[8];
    auto load_ids = [&](int j0) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int j = j0 + brow + 8 * it;
            ids_next[it] = negmap[j < D.N ? j : 0];
        }
    };
    auto issue = [&](int j0) {  // uses ids_next (ids of this chunk)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int j = j0 + 4 * qpiece;
            vs[it] = *reinterpret_cast<const float4*>(srow[it] + (j < D.N ? j : 0));
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) vbv[it] = *reinterpret_cast<const float4*>(a.emb + ids_next[it] * a.emb_ld + ncol_c);
    };
    load_ids(0);
    issue(0);
    load_ids(G_KC);
    for (int j0 = 0; j0 < D.N; j0 += G_KC) {
        {
            const int j = j0 + 4 * qpiece;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float4 v;
                v.x = (j < D.N) ? dscore<L2>(vs[it].x, lse_r[it], D.gscale) * qmask[it] : 0.f;
                v.y = (j + 1 < D.N) ? dscore<L2>(vs[it].y, lse_r[it], D.gscale) * qmask[it] : 0.f;
                v.z = (j + 2 < D.N) ? dscore<L2>(vs[it].z, lse_r[it], D.gscale) * qmask[it] : 0.f;
                v.w = (j + 3 < D.N) ? dscore<L2>(vs[it].w, lse_r[it], D.gscale) * qmask[it] : 0.f;
                lds_store4(Qs + (qrow + 16 * it) * G_KSA + 4 * qpiece, v);
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = brow + 8 * it;
                float4 v = vbv[it];
                if (!col_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j0 + row >= D.N) v = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (L2 && nl == G_TN - 4) v.w = 1.f;  // ones column -> row sums of V
                *reinterpret_cast<float4*>(Bs + row * G_TNS + nl) = v;
            }
        }
        __syncthreads();
        if (j0 + G_KC < D.N) {
            issue(j0 + G_KC);
            load_ids(j0 + 2 * G_KC);
        }
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

    if (L2) {
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
        const int nloc = wn * 64 + tn * 32 + l31;
        const int n = n0 + nloc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 + acc_row(r, h);
            const int m = m0 + ml;
            if (m < D.Bc && nloc < a.ncols && n < D.d) {
                float v = acc[tn][r];
                if (L2) v -= adj[(int64_t)m * D.d_ld + n] * rsum[ml];
                out[(int64_t)m * D.d_ld + n] = v;
            }
        }
    }
}

// =========================================================================================== dNeg = V^T . adj
template <bool L2>
__global__ __launch_bounds__(256) void lp_grad_neg_fast_kernel(GradArgs a, int tiles_m, int tiles_per_cd) {
    __shared__ __attribute__((aligned(16))) float Qs[G_KC * G_TMS];
    __shared__ __attribute__((aligned(16))) float Bs[G_KC * G_TNS];
    __shared__ float csum[G_TM];
    const LpDims& D = a.D;
    int cd, tile;
    if (!decode_block(blockIdx.x, tiles_per_cd, D.C * D.ndir, cd, tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int dir = cd / D.C, c = cd - dir * D.C;
    const int nb = tile / tiles_m;
    const int m0 = (tile - nb * tiles_m) * G_TM, n0 = nb * a.ncols;
    const int64_t rowbase = (int64_t)dir * D.Bp + (int64_t)c * D.Bc;
    const float* S = a.S + rowbase * D.n_ld;
    const float* adj = a.adj + rowbase * D.d_ld;
    const float* lse = a.lse + rowbase;

    v16f acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int qpiece = tid & 15, qrow = tid >> 4;  // Q rows = i (K), 16 thr x 16 B = 64 j (M)
    const int bpiece = tid & 31, brow = tid >> 5;
    const int jq = m0 + 4 * qpiece;
    const int jq_c = jq < D.N ? jq : 0;
    const int nl = 4 * bpiece;
    const int ncol = n0 + nl;
    const bool col_ok = (nl < a.ncols) && (ncol + 3 < D.d_ld);  // adj rows are zero padded to d_ld
    const int ncol_c = col_ok ? ncol : 0;

    float4 vs[4], vbv[8];
    float lv[4];
    auto issue = [&](int i0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = i0 + qrow + 16 * it;
            const int ic = i < D.Bc ? i : 0;
            vs[it] = *reinterpret_cast<const float4*>(S + (int64_t)ic * D.n_ld + jq_c);
            lv[it] = lse[ic];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = i0 + brow + 8 * it;
            vbv[it] = *reinterpret_cast<const float4*>(adj + (int64_t)(i < D.Bc ? i : 0) * D.d_ld + ncol_c);
        }
    };
    issue(0);
    for (int i0 = 0; i0 < D.Bc; i0 += G_KC) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = qrow + 16 * it;
            const bool ok = (i0 + row) < D.Bc;
            float4 v;
            v.x = (ok && jq < D.N) ? dscore<L2>(vs[it].x, lv[it], D.gscale) : 0.f;
            v.y = (ok && jq + 1 < D.N) ? dscore<L2>(vs[it].y, lv[it], D.gscale) : 0.f;
            v.z = (ok && jq + 2 < D.N) ? dscore<L2>(vs[it].z, lv[it], D.gscale) : 0.f;
            v.w = (ok && jq + 3 < D.N) ? dscore<L2>(vs[it].w, lv[it], D.gscale) : 0.f;
            *reinterpret_cast<float4*>(Qs + row * G_TMS + 4 * qpiece) = v;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = brow + 8 * it;
            float4 v = vbv[it];
            if (!col_ok || (i0 + row) >= D.Bc) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (L2 && nl == G_TN - 4 && (i0 + row) < D.Bc) v.w = 1.f;
            *reinterpret_cast<float4*>(Bs + row * G_TNS + nl) = v;
        }
        __syncthreads();
        if (i0 + G_KC < D.Bc) issue(i0 + G_KC);
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
        const int nloc = wn * 64 + tn * 32 + l31;
        const int n = n0 + nloc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 + acc_row(r, h);
            const int m = m0 + ml;
            if (m < D.N && nloc < a.ncols && n < D.d) {
                float v = acc[tn][r];
                if (L2) v -= a.emb[negmap[m] * a.emb_ld + n] * csum[ml];
                out[(int64_t)m * D.d_ld + n] = v;
            }
        }
    }
}

// =========================================================================================== launchers
static bool fast_ok(const float* emb, int64_t emb_ld, int d) {
    return (d % 4 == 0) && (emb_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(emb) & 15) == 0);
}

bool launch_scores_fast(const ScoreArgs& a, bool l2, hipStream_t st) {
    if (!fast_ok(a.emb, a.emb_ld, a.D.d)) return false;
    const int tiles_n = (int)cdiv(a.D.N, F_TN), tiles_m = (int)cdiv(a.D.Bc, F_TM);
    const int tiles = tiles_n * tiles_m;
    const size_t lds = (size_t)(F_TM + F_TN) * a.KS * sizeof(float);
    dim3 grid(xcd_grid(tiles, a.D.C * a.D.ndir));
    if (l2)
        lp_scores_fast_kernel<true><<<grid, dim3(256), lds, st>>>(a, tiles_n, tiles);
    else
        lp_scores_fast_kernel<false><<<grid, dim3(256), lds, st>>>(a, tiles_n, tiles);
    return true;
}

bool launch_grad_adj_fast(const GradArgs& a, bool l2, hipStream_t st) {
    if (!fast_ok(a.emb, a.emb_ld, a.D.d) || (l2 && a.D.d > a.ncols)) return false;  // multi n-block L2 (d >= 128): generic path
    const int tiles_m = (int)cdiv(a.D.Bc, G_TM), nblk = (int)cdiv(a.D.d, a.ncols);
    const int tiles = tiles_m * nblk;
    dim3 grid(xcd_grid(tiles, a.D.C * a.D.ndir));
    if (l2)
        lp_grad_adj_fast_kernel<true><<<grid, dim3(256), 0, st>>>(a, tiles_m, tiles);
    else
        lp_grad_adj_fast_kernel<false><<<grid, dim3(256), 0, st>>>(a, tiles_m, tiles);
    return true;
}

bool launch_grad_neg_fast(const GradArgs& a, bool l2, hipStream_t st) {
    if (!fast_ok(a.emb, a.emb_ld, a.D.d) || (l2 && a.D.d > a.ncols)) return false;
    const int tiles_m = (int)cdiv(a.D.N, G_TM), nblk = (int)cdiv(a.D.d, a.ncols);
    const int tiles = tiles_m * nblk;
    dim3 grid(xcd_grid(tiles, a.D.C * a.D.ndir));
    if (l2)
        lp_grad_neg_fast_kernel<true><<<grid, dim3(256), 0, st>>>(a, tiles_m, tiles);
    else
        lp_grad_neg_fast_kernel<false><<<grid, dim3(256), 0, st>>>(a, tiles_m, tiles);
    return true;
}

}  // namespace marius
