// Backward contractions on the BF16 matrix pipe with exact 3-way operand splitting (companion of lp_split.hip; Dot comparator).
//
//   dAdj_c[m][n] = sum_j V[m][j] * Neg_c[j][n]          dNeg_c[j][n] = sum_m V[m][j] * adj_c[m][n]          V = dL/dS = gscale * exp(S - lse)
//
// v_mfma_f32_16x16x32_bf16 wants, per lane, eight CONSECUTIVE contraction indices of one row (A) / one column (B):
//   * the V operand never goes through LDS: every wave owns 16 output rows, so a lane reads exactly its own S values from global
//     memory (dAdj: two 16-B loads along a row of S; dNeg: eight 4-B loads down a column, 64-B segments per 16 lanes), applies
//     exp2 and splits into the three bf16 planes in registers;
//   * the other operand must be contraction-major: lp_transpose_planes_kernel builds, once per step, negT[cd][plane][n][j] (the chunk's
//     negative rows, gathered and transposed from the emb planes) and adjT[cd][plane][n][m] (from the adj planes); tiles of them are
//     staged through LDS (shared by the four waves) with plain 16-B copies.
// Six MFMAs per 16x16x32 block (hh, hm, mh, hl, lh, mm) reproduce the fp32 product up to the dropped 2^-24 terms; 42 MFMAs x 16
// cycles per wave and K chunk, underneath the VALU work (BF16 MFMAs and VALU overlap; FP32 MFMAs and VALU do not).
#include "lp_common.h"

namespace marius {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // first-class vectors: structs of HIP's float4 carried through the loop end up in scratch

constexpr int GB_TM = 64;            // output rows per workgroup (4 waves x 16)
constexpr int GB_KC = 32;            // K chunk = one MFMA
constexpr int GB_RS = GB_KC + 8;     // bf16 per LDS row of the B tile (80 B): 16-B slots of 8 consecutive rows on disjoint banks

__device__ __forceinline__ v4f mfma16_bf16(const v8bf& a, const v8bf& b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void split8v(const float (&x)[8], v8bf& H, v8bf& M, v8bf& L) {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        const float r1 = x[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        H[j] = h;
        M[j] = m;
        L[j] = (__bf16)r2;  // exact: at most 8 significant bits are left
    }
}

// ---------------------------------------------------------------------------------------------------------------- transposed planes
// out[cd][plane][n][k], k < kld (zero padded): k-th row of chunk-direction cd is planes_in[plane][row_of(cd, k)][n].
// rows: ids != null -> ids[cd * K + k] (gather), else cd * K + k.  One block = 32 source rows x all n of the three planes.
__global__ __launch_bounds__(256) void lp_transpose_planes_kernel(const __bf16* __restrict__ in, int64_t in_plane, int kp, const int64_t* __restrict__ ids0,
                                                                  const int64_t* __restrict__ ids1, int cd_per_dir, int K, int kld,
                                                                  __bf16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    __bf16* tile = reinterpret_cast<__bf16*>(tsm);  // [32][3 * kp + 8]
    const int TS = 3 * kp + 8;
    const int cd = blockIdx.y, k0 = blockIdx.x * 32;
    const int tid = threadIdx.x;
    const int ppr = 3 * (kp >> 3);  // 16-B pieces per source row over the three planes
    for (int q = tid; q < 32 * ppr; q += 256) {
        const int r = q / ppr, p = q - r * ppr;
        const int plane = p / (kp >> 3), w = p - plane * (kp >> 3);
        const int k = k0 + r;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (k < K) {
            int64_t row;
            if (ids0) {
                const int dir = cd / cd_per_dir, c = cd - dir * cd_per_dir;
                row = (dir ? ids1 : ids0)[(int64_t)c * K + k];
            } else {
                row = (int64_t)cd * K + k;
            }
            v = *reinterpret_cast<const u32x4*>(in + (int64_t)plane * in_plane + row * kp + 8 * w);
        }
        *reinterpret_cast<u32x4*>(tile + r * TS + plane * kp + 8 * w) = v;
    }
    __syncthreads();
    // out rows: (plane, n) -> 32 consecutive k = 64 B = 4 pieces
    const int orows = 3 * kp;
    for (int q = tid; q < orows * 4; q += 256) {
        const int orow = q >> 2, w = q & 3;  // orow = plane * kp + n
        v8bf v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[(8 * w + e) * TS + orow];
        const int plane = orow / kp, n = orow - plane * kp;
        *reinterpret_cast<v8bf*>(out + (((int64_t)cd * 3 + plane) * kp + n) * kld + k0 + 8 * w) = v;
    }
}

int launch_transpose_planes(const void* in, int64_t in_plane, int kp, const int64_t* ids0, const int64_t* ids1, int cd_per_dir, int ncd, int K, int kld,
                            void* out, hipStream_t st) {
    const size_t lds = (size_t)32 * (3 * kp + 8) * sizeof(__bf16);
    lp_transpose_planes_kernel<<<dim3((unsigned)(kld / 32), (unsigned)ncd), dim3(256), lds, st>>>((const __bf16*)in, in_plane, kp, ids0, ids1, cd_per_dir, K, kld,
                                                                                                 (__bf16*)out);
    return check_launch("lp_transpose_planes");
}

// ---------------------------------------------------------------------------------------------------------------- the contractions
struct GradB6Args {
    const float* S;
    const float* lse;
    const __bf16* negT;  // [ncd][3][kp][nld]
    const __bf16* adjT;  // [ncd][3][kp][bld]
    float* dadj;
    float* gocc;
    int64_t negocc_off[2];
    int kp, nld, bld;
    LpDims D;
};

// common K loop: B tile [3][NT*16][GB_RS] double-buffered in LDS; `loadA(ch, aH, aM, aL)` produces the wave's V fragments of chunk ch
template <int NT, class LoadS, class MakeA>
__device__ __forceinline__ void gradb6_loop(const __bf16* __restrict__ bT, int kld, int nchunks, __bf16* lds, v4f (&acc)[NT], LoadS loadS, MakeA makeA) {
    constexpr int NR = NT * 16;                 // B rows (output columns)
    constexpr int PLANE = NR * GB_RS;           // bf16 per plane in LDS
    constexpr int BUF = 3 * PLANE;
    constexpr int NP = 3 * NR * 4;              // 16-B pieces per chunk
    constexpr int NI = (NP + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, g = lane >> 4;
    // staging: piece q = tid + 256 i  ->  (plane, n, w): global bT[plane][n][k0 + 8 w], LDS [plane][n][8 w]
    uint32_t goff[NI];
    int loff[NI];
    bool pok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = tid + 256 * i;
        pok[i] = q < NP;
        const int qc = pok[i] ? q : 0;
        const int row = qc >> 2, w = qc & 3;    // row = plane * NR + n
        const int plane = row / NR, n = row - plane * NR;
        goff[i] = (uint32_t)(((int64_t)plane * (NR) + n) * kld + 8 * w) * 2u;  // plane stride inside one cd is kp * kld with kp == NR
        loff[i] = plane * PLANE + n * GB_RS + 8 * w;
    }
    struct Tile { u32x4 v[NI]; };
    auto issueB = [&](int ch) __attribute__((always_inline)) {
        Tile t;
        const int chc = ch < nchunks ? ch : nchunks - 1;
        const char* base = reinterpret_cast<const char*>(bT) + (size_t)chc * GB_KC * 2;
#pragma unroll
        for (int i = 0; i < NI; ++i) t.v[i] = *reinterpret_cast<const u32x4*>(base + goff[i]);
        return t;
    };
    auto writeB = [&](int buf, Tile t) __attribute__((always_inline)) {
        __bf16* b = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (pok[i]) *reinterpret_cast<u32x4*>(b + loff[i]) = t.v[i];
    };
    auto compute = [&](int buf, const v8bf& aH, const v8bf& aM, const v8bf& aL) __attribute__((always_inline)) {
        const __bf16* bp = lds + buf * BUF + l15 * GB_RS + 8 * g;
        v8bf bH[2], bM[2], bL[2];
        bH[0] = *reinterpret_cast<const v8bf*>(bp);
        bM[0] = *reinterpret_cast<const v8bf*>(bp + PLANE);
        bL[0] = *reinterpret_cast<const v8bf*>(bp + 2 * PLANE);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c_ = t & 1, n_ = c_ ^ 1;
            if (t + 1 < NT) {
                bH[n_] = *reinterpret_cast<const v8bf*>(bp + (t + 1) * 16 * GB_RS);
                bM[n_] = *reinterpret_cast<const v8bf*>(bp + (t + 1) * 16 * GB_RS + PLANE);
                bL[n_] = *reinterpret_cast<const v8bf*>(bp + (t + 1) * 16 * GB_RS + 2 * PLANE);
            }
            acc[t] = mfma16_bf16(aL, bH[c_], acc[t]);
            acc[t] = mfma16_bf16(aH, bL[c_], acc[t]);
            acc[t] = mfma16_bf16(aM, bM[c_], acc[t]);
            acc[t] = mfma16_bf16(aM, bH[c_], acc[t]);
            acc[t] = mfma16_bf16(aH, bM[c_], acc[t]);
            acc[t] = mfma16_bf16(aH, bH[c_], acc[t]);
        }
    };

    // pipeline: B tile of chunk ch+1 is written to the other LDS buffer while chunk ch is multiplied; its global loads were
    // issued one chunk earlier; the S values of chunk ch+2 are in flight
    Tile tb = issueB(0);
    auto s0 = loadS(0);
    auto s1 = loadS(1);
    writeB(0, tb);
    tb = issueB(1);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
        {
            v8bf aH, aM, aL;
            makeA(ch, s0, aH, aM, aL);
            s0 = loadS(ch + 2);
            compute(0, aH, aM, aL);
            writeB(1, tb);
            tb = issueB(ch + 2);
            __syncthreads();
        }
        {   // unconditional (a phantom chunk past the end multiplies zeros: makeA masks every element): a branch around the loads
            // would make the compiler merge the vmcnt state conservatively and drain the prefetch queue every chunk
            v8bf aH, aM, aL;
            makeA(ch + 1, s1, aH, aM, aL);
            s1 = loadS(ch + 3);
            compute(1, aH, aM, aL);
        }
        writeB(0, tb);
        tb = issueB(ch + 3);
        __syncthreads();
    }
}

struct SAdj { f32x4 a, b; };                 // dAdj: S[m][j0 + 8 g .. + 7]
struct SNeg { f32x4 s0, s1, la, lb; };      // dNeg: S[m0 + 8 g + e][j] (e = 0..7), lse[m0 + 8 g .. + 7]

template <int NT>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void lp_grad_b6_kernel(GradB6Args a, int tiles_adj, int tiles_neg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    __bf16* lds = reinterpret_cast<__bf16*>(gsm);
    const LpDims& D = a.D;
    constexpr float LOG2E = 1.4426950408889634f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // block -> (cd, unit): XCD-aware as in the FP32 kernels (block b runs on XCD b % 8; a chunk's tiles stay on one XCD)
    const int units = tiles_adj + tiles_neg;
    const int ncd = D.C * D.ndir;
    const int lin = blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int cd = (idx / units) * 8 + xcd, unit = idx % units;
    if (cd >= ncd) return;
    const int dir = cd / D.C, c = cd - dir * D.C;
    const int64_t rowbase = (int64_t)dir * D.Bp + (int64_t)c * D.Bc;
    const float* S = a.S + rowbase * D.n_ld;
    const float lg = __log2f(D.gscale);

    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

    if (unit < tiles_adj) {
        // ---------------- dAdj tile: rows m0 .. m0 + 63, K = negatives j
        const int m0 = unit * GB_TM;
        const int m = m0 + wave * 16 + l15;
        const bool m_ok = m < D.Bc;
        const float* srow = S + (int64_t)(m_ok ? m : 0) * D.n_ld;
        const float cexp = m_ok ? lg - a.lse[rowbase + m] * LOG2E : -INFINITY;  // -inf switches a padding row off
        const int nchunks = (D.N + GB_KC - 1) / GB_KC;
        const __bf16* bT = a.negT + (int64_t)cd * 3 * a.kp * a.nld;
        auto loadS = [&](int ch) __attribute__((always_inline)) {
            SAdj s;
            const int chc = ch < nchunks ? ch : nchunks - 1;
            const int j = chc * GB_KC + 8 * g;
            // n_ld is N rounded up to 4: each 16-B half is either entirely inside the row or entirely masked (then: any valid address)
            s.a = *reinterpret_cast<const f32x4*>(srow + ((j + 3 < (int)D.n_ld) ? j : 0));
            s.b = *reinterpret_cast<const f32x4*>(srow + ((j + 7 < (int)D.n_ld) ? j + 4 : 0));
            return s;
        };
        auto makeA = [&](int ch, const SAdj& s, v8bf& aH, v8bf& aM, v8bf& aL) __attribute__((always_inline)) {
            const float sv[8] = {s.a[0], s.a[1], s.a[2], s.a[3], s.b[0], s.b[1], s.b[2], s.b[3]};
            float x[8];
            const int j = ch * GB_KC + 8 * g;
            if (ch * GB_KC + GB_KC <= D.N) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = __builtin_amdgcn_exp2f(fmaf(sv[e], LOG2E, cexp));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (j + e < D.N) ? __builtin_amdgcn_exp2f(fmaf(sv[e], LOG2E, cexp)) : 0.f;
            }
            split8v(x, aH, aM, aL);
        };
        gradb6_loop<NT>(bT, a.nld, nchunks, lds, acc, loadS, makeA);
        // lane holds D[m = 4 g + r][n = l15] of each 16x16 tile
        float* out = a.dadj + rowbase * D.d_ld;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * t + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = m0 + wave * 16 + 4 * g + r;
                if (mm < D.Bc && n < D.d) out[(int64_t)mm * D.d_ld + n] = acc[t][r];
            }
        }
    } else {
        // ---------------- dNeg tile: rows j0 .. j0 + 63 (negatives), K = batch rows m
        const int j0 = (unit - tiles_adj) * GB_TM;
        const int j = j0 + wave * 16 + l15;
        const int jc = j < D.N ? j : 0;  // rows past N are computed on valid data and never stored
        const float* scol = S + jc;
        const float* lse = a.lse + rowbase;
        const int nchunks = (D.Bc + GB_KC - 1) / GB_KC;
        const __bf16* bT = a.adjT + (int64_t)cd * 3 * a.kp * a.bld;
        auto loadS = [&](int ch) __attribute__((always_inline)) {
            SNeg s;
            const int chc = ch < nchunks ? ch : nchunks - 1;
            const int mb = chc * GB_KC + 8 * g;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int mm = mb + e;
                t[e] = scol[(int64_t)(mm < D.Bc ? mm : 0) * D.n_ld];
            }
            s.s0 = (f32x4){t[0], t[1], t[2], t[3]};
            s.s1 = (f32x4){t[4], t[5], t[6], t[7]};
            // lse is [ndir][Bp] inside the workspace, followed by rowloss: the (masked) entries past this chunk's Bc rows — at the very
            // end of the array up to 31 floats past it — are readable memory
            s.la = *reinterpret_cast<const f32x4*>(lse + mb);
            s.lb = *reinterpret_cast<const f32x4*>(lse + mb + 4);
            return s;
        };
        auto makeA = [&](int ch, const SNeg& s, v8bf& aH, v8bf& aM, v8bf& aL) __attribute__((always_inline)) {
            const float lv[8] = {s.la[0], s.la[1], s.la[2], s.la[3], s.lb[0], s.lb[1], s.lb[2], s.lb[3]};
            const float sv[8] = {s.s0[0], s.s0[1], s.s0[2], s.s0[3], s.s1[0], s.s1[1], s.s1[2], s.s1[3]};
            float x[8];
            const int mb = ch * GB_KC + 8 * g;
            if (ch * GB_KC + GB_KC <= D.Bc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = __builtin_amdgcn_exp2f(fmaf(sv[e], LOG2E, fmaf(-lv[e], LOG2E, lg)));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (mb + e < D.Bc) ? __builtin_amdgcn_exp2f(fmaf(sv[e], LOG2E, fmaf(-lv[e], LOG2E, lg))) : 0.f;
            }
            split8v(x, aH, aM, aL);
        };
        gradb6_loop<NT>(bT, a.bld, nchunks, lds, acc, loadS, makeA);
        float* out = a.gocc + (a.negocc_off[dir] + (int64_t)c * D.N) * D.d_ld;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * t + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = j0 + wave * 16 + 4 * g + r;
                if (jj < D.N && n < D.d) out[(int64_t)jj * D.d_ld + n] = acc[t][r];
            }
        }
    }
}

// planes must have been produced by marius_lp_forward of the same step (variant 'b'); negT / adjT are built here
bool launch_grad_b6(const GradArgs& ga, const void* embp, int64_t embp_plane, const void* adjp, int64_t adjp_plane, int kp, void* negT, void* adjT,
                    hipStream_t st) {
    const LpDims& D = ga.D;
    if (D.cmp == MARIUS_CMP_L2) return false;
    const int nt = kp / 16;
    if (!(nt == 2 || nt == 4 || nt == 7 || nt == 8)) return false;
    const int ncd = D.C * D.ndir;
    const int nld = (D.N + 31) / 32 * 32, bld = (D.Bc + 31) / 32 * 32;
    if ((int64_t)3 * kp * (nld > bld ? nld : bld) * 2 >= ((int64_t)1 << 31)) return false;
    if (launch_transpose_planes(embp, embp_plane, kp, ga.negmap[0], ga.negmap[1], D.C, ncd, D.N, nld, negT, st)) return false;
    if (launch_transpose_planes(adjp, adjp_plane, kp, nullptr, nullptr, D.C, ncd, D.Bc, bld, adjT, st)) return false;
    GradB6Args a;
    a.S = ga.S;
    a.lse = ga.lse;
    a.negT = (const __bf16*)negT;
    a.adjT = (const __bf16*)adjT;
    a.dadj = ga.dadj;
    a.gocc = ga.gocc;
    a.negocc_off[0] = ga.negocc_off[0];
    a.negocc_off[1] = ga.negocc_off[1];
    a.kp = kp;
    a.nld = nld;
    a.bld = bld;
    a.D = D;
    const int tiles_adj = (int)cdiv(D.Bc, GB_TM), tiles_neg = (int)cdiv(D.N, GB_TM);
    const int units = tiles_adj + tiles_neg;
    const unsigned grid = (unsigned)(((ncd + 7) / 8) * 8 * units);
    const size_t lds = (size_t)2 * 3 * kp * GB_RS * sizeof(__bf16);
#define GB_LAUNCH(NTV)                                                                                                              \
    do {                                                                                                                            \
        static bool attr_set = false;                                                                                               \
        if (!attr_set && lds > 65536) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)lp_grad_b6_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
            attr_set = true;                                                                                                        \
        }                                                                                                                           \
        lp_grad_b6_kernel<NTV><<<dim3(grid), dim3(256), lds, st>>>(a, tiles_adj, tiles_neg);                                        \
    } while (0)
    switch (nt) {
        case 2: GB_LAUNCH(2); break;
        case 4: GB_LAUNCH(4); break;
        case 7: GB_LAUNCH(7); break;
        default: GB_LAUNCH(8); break;
    }
#undef GB_LAUNCH
    return true;
}

}  // namespace marius
