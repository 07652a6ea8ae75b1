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

#ifndef GB6_ABLATE
#define GB6_ABLATE 0  // experiment builds only: 2 = no MFMAs, 4 = no B staging, 8 = no S loads, 16 = no exp/split (constant V)
#endif

namespace marius {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // first-class vectors: structs of HIP's float4 carried through the loop end up in scratch

constexpr int GB_TM = 128;           // output rows per workgroup (4 waves x 32)
constexpr int GB_KC = 32;            // K chunk = two 32x32x16 MFMA blocks
constexpr int GB_RS = GB_KC;         // bf16 per LDS row of the B tile (64 B, no padding); the 16-B slot w of row n lives at slot
                                     // w ^ ((n >> 2) & 3), which keeps 8 consecutive rows of one fragment read on disjoint bank groups

typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16f mfma32_bf16(const v8bf& a, const v8bf& b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

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
    unsigned long long* dbg;  // debug only: cycle stamps (marius_debug_set_timeline + MARIUS_TIMELINE_GRADS)
    LpDims D;
};

// Why 32 rows per wave and 32x32x16 MFMAs: with 16-row wave tiles (16x16x32) the kernel was bound by LDS reads — every wave re-reads
// the whole B tile for 16 rows of output (ablation: 0.14 ms of pure fragment traffic, 8.8 GB at 69 TB/s) — and by the L2 -> LDS copies of
// the B tile (0.12 ms).  A 32-row wave tile halves the fragment bytes per flop, a 128-row workgroup tile halves the copies.
//
// common K loop: B tile [3][NTB*32][GB_RS] double-buffered in LDS; loadS(ch) fetches the wave's S values of chunk ch (two chunks ahead),
// makeA(ch, s, kb, ...) turns them into the V fragments of K block kb
template <int NTB, class LoadS, class MakeA>
__device__ __forceinline__ void gradb6_loop(const __bf16* __restrict__ bT, int kp, int kld, int nchunks, __bf16* lds, v16f (&acc)[NTB], LoadS loadS, MakeA makeA) {
    constexpr int NR = NTB * 32;                // B rows held in LDS (output columns, zero rows past kp)
    constexpr int PLANE = NR * GB_RS;           // bf16 per plane in LDS
    constexpr int BUF = 3 * PLANE;
    constexpr int NI = (3 * NR * 4 + 255) / 256;  // upper bound of 16-B pieces per thread and chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const int np = 3 * kp * 4;                  // 16-B pieces per chunk: (plane, n < kp, w)
    // staging: piece q = tid + 256 i  ->  (plane, n, w): global bT[plane][n][k0 + 8 w], LDS [plane][n][8 (w ^ swz)]
    uint32_t goff[NI];
    int loff[NI];
    bool pok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = tid + 256 * i;
        pok[i] = q < np;
        const int qc = pok[i] ? q : 0;
        const int row = qc >> 2, w = qc & 3;    // row = plane * kp + n
        const int plane = row / kp, n = row - plane * kp;
        goff[i] = (uint32_t)(((int64_t)plane * kp + n) * kld + 8 * w) * 2u;
        loff[i] = plane * PLANE + n * GB_RS + 8 * (w ^ ((n >> 2) & 3));
    }
    for (int i = tid; i < 2 * 3 * (NR - kp) * 4; i += 256) {  // rows kp .. NR of every plane of both buffers: zero, never staged
        const int b_ = i / (3 * (NR - kp) * 4), r = i % (3 * (NR - kp) * 4);
        const int plane = r / ((NR - kp) * 4), rr = r % ((NR - kp) * 4);
        *reinterpret_cast<u32x4*>(lds + b_ * BUF + plane * PLANE + (kp + (rr >> 2)) * GB_RS + 8 * (rr & 3)) = (u32x4){0u, 0u, 0u, 0u};
    }
    struct Tile { u32x4 v[NI]; };
    // buffer loads: per-lane byte offset in a VGPR (constant over the K loop) + the chunk offset in an SGPR: no per-load address
    // arithmetic on the VALU, and reads past the end of the buffer return zero instead of faulting
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(bT), 0, 3 * kp * kld * 2, 0x00020000);
    auto issueB = [&](int ch) __attribute__((always_inline)) {
        Tile t;
        const int chc = ch < nchunks ? ch : nchunks - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) t.v[i] = __builtin_amdgcn_raw_buffer_load_b128(brs, (int)goff[i], chc * GB_KC * 2, 0);
        return t;
    };
    auto writeB = [&](int buf, Tile t) __attribute__((always_inline)) {
        __bf16* b = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (pok[i]) *reinterpret_cast<u32x4*>(b + loff[i]) = t.v[i];
    };
    auto compute = [&](int buf, const v8bf (&aH)[2], const v8bf (&aM)[2], const v8bf (&aL)[2]) __attribute__((always_inline)) {
        // lane reads row n = 32 t + l31, K block kb: slot (2 kb + h) ^ ((n >> 2) & 3), and (n >> 2) & 3 == (l31 >> 2) & 3 for every t.
        // Two column tiles are multiplied in lock step: a v_mfma_f32_32x32x16_bf16 occupies the pipe for 32 cycles but its result is
        // only available to a dependent MFMA after ~64 (timeline: a 48-long chain on one accumulator ran at 66 cycles per MFMA), so
        // consecutive MFMAs must alternate between two accumulators.
        const int swz = (l31 >> 2) & 3;
        const __bf16* bp = lds + buf * BUF + l31 * GB_RS;
        constexpr int TP = NTB >= 2 ? 2 : 1;          // tiles per step
        constexpr int NS = (NTB / TP) * 2;            // steps: (tile pair, K block)
        v8bf bH[2][TP], bM[2][TP], bL[2][TP];
        auto rd = [&](int s_, int slot) __attribute__((always_inline)) {
            const int kb2 = s_ & 1, pr = s_ >> 1;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const __bf16* q = bp + (pr * TP + j) * 32 * GB_RS + 8 * ((2 * kb2 + h) ^ swz);
                bH[slot][j] = *reinterpret_cast<const v8bf*>(q);
                bM[slot][j] = *reinterpret_cast<const v8bf*>(q + PLANE);
                bL[slot][j] = *reinterpret_cast<const v8bf*>(q + 2 * PLANE);
            }
        };
        rd(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int kb = s_ & 1, pr = s_ >> 1, c_ = s_ & 1;
            if (s_ + 1 < NS) rd(s_ + 1, c_ ^ 1);
            if (GB6_ABLATE & 2) continue;
#define GB6_STEP(AX, BX)                                                                  \
    _Pragma("unroll") for (int j = 0; j < TP; ++j) acc[pr * TP + j] = mfma32_bf16(AX[kb], BX[c_][j], acc[pr * TP + j]);
            GB6_STEP(aL, bH)
            GB6_STEP(aH, bL)
            GB6_STEP(aM, bM)
            GB6_STEP(aM, bH)
            GB6_STEP(aH, bM)
            GB6_STEP(aH, bH)
#undef GB6_STEP
        }
    };

    // pipeline: B tile of chunk ch+1 is written to the other LDS buffer while chunk ch is multiplied; its global loads were
    // issued one chunk earlier; the S values of chunk ch+2 are in flight.  The V fragments of chunk ch+1 are produced (exp2 + split,
    // ~150 VALU instructions) in the same scheduling region as the 48 MFMAs of chunk ch, so the scheduler is free to interleave them
    // (forcing a 1 MFMA : 4 VALU pattern with sched_group_barrier was measured slower: 0.467 vs 0.444 ms).
    Tile tb = issueB(0);
    auto s0 = loadS(0);
    auto s1 = loadS(1);
    writeB(0, tb);
    tb = issueB(1);
    v8bf cH[2], cM[2], cL[2];
    makeA(0, s0, cH, cM, cL);
    s0 = loadS(2);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
        v8bf nH[2], nM[2], nL[2];
        {
            makeA(ch + 1, s1, nH, nM, nL);   // masks every element of a phantom chunk past the end: branch-free on purpose
            compute(0, cH, cM, cL);
            if (!(GB6_ABLATE & 8)) s1 = loadS(ch + 3);
            if (!(GB6_ABLATE & 4)) {
                writeB(1, tb);
                tb = issueB(ch + 2);
            }
            __syncthreads();
        }
        {
            makeA(ch + 2, s0, cH, cM, cL);
            compute(1, nH, nM, nL);
            if (!(GB6_ABLATE & 8)) s0 = loadS(ch + 4);
        }
        if (!(GB6_ABLATE & 4)) {
            writeB(0, tb);
            tb = issueB(ch + 3);
        }
        __syncthreads();
    }
}

// ---- ping-pong form: an 8-wave workgroup = two groups of four waves that run the same K loop half a phase apart.  In every phase one
// group issues the 48 MFMAs of a chunk while the other produces its next V fragments (exp2 + split), moves the next B tile to LDS
// (group B only) and issues its loads; a workgroup barrier ends the phase.  Each SIMD hosts one wave of each group, so its matrix pipe
// always has a computing wave and its VALU a preparing one — BF16 MFMAs and VALU overlap on gfx950 (FP32 MFMAs do not: the same
// structure was a loss for the FP32 score kernel).  Three LDS buffers: tile c is read by group A in phase 2c and by group B in
// phase 2c + 1 while group B writes tile c + 1 in phase 2c.
template <int NTB, bool GROUPB, class LoadS, class MakeA>
__device__ __forceinline__ void gradb6_pp_loop(const __bf16* __restrict__ bT, int kp, int kld, int nchunks, __bf16* lds, v16f (&acc)[NTB], LoadS loadS, MakeA makeA,
                                               unsigned long long* dbg_base) {
    unsigned long long* dbg = (dbg_base && blockIdx.x < 2048 && (blockIdx.x & 7) == 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 4))
                                  ? dbg_base + ((size_t)(blockIdx.x >> 3) * 2 + ((threadIdx.x >> 6) ? 1 : 0)) * 64 : nullptr;
    int dbi = 0;
#define PSTAMP() do { if (dbg && dbi < 64) dbg[dbi++] = __builtin_readcyclecounter(); } while (0)
    PSTAMP();
    constexpr int NR = NTB * 32;
    constexpr int PLANE = NR * GB_RS;
    constexpr int BUF = 3 * PLANE;
    constexpr int NI = (3 * NR * 4 + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int st = tid & 255;                   // staging thread index inside group B
    const int l31 = lane & 31, h = lane >> 5;
    const int np = 3 * kp * 4;
    uint32_t goff[NI];
    int loff[NI];
    bool pok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = st + 256 * i;
        pok[i] = q < np;
        const int qc = pok[i] ? q : 0;
        const int row = qc >> 2, w = qc & 3;
        const int plane = row / kp, n = row - plane * kp;
        goff[i] = (uint32_t)(((int64_t)plane * kp + n) * kld + 8 * w) * 2u;
        loff[i] = plane * PLANE + n * GB_RS + 8 * (w ^ ((n >> 2) & 3));
    }
    struct Tile { u32x4 v[NI]; };
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(bT), 0, 3 * kp * kld * 2, 0x00020000);
    auto issueB = [&](int ch) __attribute__((always_inline)) {
        Tile t;
        const int chc = ch < nchunks ? ch : nchunks - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) t.v[i] = __builtin_amdgcn_raw_buffer_load_b128(brs, (int)goff[i], chc * GB_KC * 2, 0);
        return t;
    };
    auto writeB = [&](int buf, Tile t) __attribute__((always_inline)) {
        __bf16* b = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (pok[i]) *reinterpret_cast<u32x4*>(b + loff[i]) = t.v[i];
    };
    auto compute = [&](int buf, const v8bf (&aH)[2], const v8bf (&aM)[2], const v8bf (&aL)[2]) __attribute__((always_inline)) {
        // lane reads row n = 32 t + l31, K block kb: slot (2 kb + h) ^ ((n >> 2) & 3), and (n >> 2) & 3 == (l31 >> 2) & 3 for every t.
        // Two column tiles are multiplied in lock step: a v_mfma_f32_32x32x16_bf16 occupies the pipe for 32 cycles but its result is
        // only available to a dependent MFMA after ~64 (timeline: a 48-long chain on one accumulator ran at 66 cycles per MFMA), so
        // consecutive MFMAs must alternate between two accumulators.
        const int swz = (l31 >> 2) & 3;
        const __bf16* bp = lds + buf * BUF + l31 * GB_RS;
        constexpr int TP = NTB >= 2 ? 2 : 1;          // tiles per step
        constexpr int NS = (NTB / TP) * 2;            // steps: (tile pair, K block)
        v8bf bH[2][TP], bM[2][TP], bL[2][TP];
        auto rd = [&](int s_, int slot) __attribute__((always_inline)) {
            const int kb2 = s_ & 1, pr = s_ >> 1;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const __bf16* q = bp + (pr * TP + j) * 32 * GB_RS + 8 * ((2 * kb2 + h) ^ swz);
                bH[slot][j] = *reinterpret_cast<const v8bf*>(q);
                bM[slot][j] = *reinterpret_cast<const v8bf*>(q + PLANE);
                bL[slot][j] = *reinterpret_cast<const v8bf*>(q + 2 * PLANE);
            }
        };
        rd(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int kb = s_ & 1, pr = s_ >> 1, c_ = s_ & 1;
            if (s_ + 1 < NS) rd(s_ + 1, c_ ^ 1);
            __builtin_amdgcn_sched_barrier(0);  // pin the fragment prefetch above the MFMAs of this step (hipcc otherwise sinks each read to its use)
            if (GB6_ABLATE & 2) continue;
#define GB6_STEP(AX, BX)                                                                  \
    _Pragma("unroll") for (int j = 0; j < TP; ++j) acc[pr * TP + j] = mfma32_bf16(AX[kb], BX[c_][j], acc[pr * TP + j]);
            GB6_STEP(aL, bH)
            GB6_STEP(aH, bL)
            GB6_STEP(aM, bM)
            GB6_STEP(aM, bH)
            GB6_STEP(aH, bM)
            GB6_STEP(aH, bH)
#undef GB6_STEP
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int npair = (nchunks + 1) / 2;   // the loop is unrolled by two chunks (static register sets); a phantom last chunk multiplies zeros
    v8bf cH[2], cM[2], cL[2];
    auto s0 = loadS(0);
    auto s1 = loadS(1);
    if (!GROUPB) {
        // ------------------------------------------------ group A: computes chunk c in phase 2c, prepares chunk c + 1 in phase 2c + 1
        makeA(0, s0, cH, cM, cL);
        s0 = loadS(2);
        __syncthreads();
        PSTAMP();
        int buf = 0;
        for (int p = 0; p < npair; ++p) {
            const int c = 2 * p;
            __builtin_amdgcn_s_setprio(0);
            compute(buf, cH, cM, cL);                       // phase 2c
            if (p < 7) PSTAMP();
            __syncthreads();
            if (p < 7) PSTAMP();
            __builtin_amdgcn_s_setprio(3);
            makeA(c + 1, s1, cH, cM, cL);                   // phase 2c + 1
            s1 = loadS(c + 3);
            if (p < 7) PSTAMP();
            __syncthreads();
            if (p < 7) PSTAMP();
            buf = buf == 2 ? 0 : buf + 1;
            __builtin_amdgcn_s_setprio(0);
            compute(buf, cH, cM, cL);                       // phase 2c + 2
            __syncthreads();
            __builtin_amdgcn_s_setprio(3);
            makeA(c + 2, s0, cH, cM, cL);                   // phase 2c + 3
            s0 = loadS(c + 4);
            __syncthreads();
            buf = buf == 2 ? 0 : buf + 1;
        }
    } else {
        // ------------------------------------------------ group B: stages the B tiles; prepares chunk c in phase 2c, computes it in phase 2c + 1
        for (int i = st; i < 3 * 3 * (NR - kp) * 4; i += 256) {  // rows kp .. NR of every plane of the three buffers: zero, never staged
            const int b_ = i / (3 * (NR - kp) * 4), r = i % (3 * (NR - kp) * 4);
            const int plane = r / ((NR - kp) * 4), rr = r % ((NR - kp) * 4);
            *reinterpret_cast<u32x4*>(lds + b_ * BUF + plane * PLANE + (kp + (rr >> 2)) * GB_RS + 8 * (rr & 3)) = (u32x4){0u, 0u, 0u, 0u};
        }
        Tile tb = issueB(0);
        writeB(0, tb);
        tb = issueB(1);
        __syncthreads();
        PSTAMP();
        int buf = 0;
        for (int p = 0; p < npair; ++p) {
            const int c = 2 * p;
            __builtin_amdgcn_s_setprio(3);
            makeA(c, s0, cH, cM, cL);                       // phase 2c
            s0 = loadS(c + 2);
            writeB(buf == 2 ? 0 : buf + 1, tb);             // tile c + 1
            tb = issueB(c + 2);
            if (p < 7) PSTAMP();
            __syncthreads();
            if (p < 7) PSTAMP();
            __builtin_amdgcn_s_setprio(0);
            compute(buf, cH, cM, cL);                       // phase 2c + 1
            if (p < 7) PSTAMP();
            __syncthreads();
            if (p < 7) PSTAMP();
            buf = buf == 2 ? 0 : buf + 1;
            __builtin_amdgcn_s_setprio(3);
            makeA(c + 1, s1, cH, cM, cL);                   // phase 2c + 2
            s1 = loadS(c + 3);
            writeB(buf == 2 ? 0 : buf + 1, tb);             // tile c + 2
            tb = issueB(c + 3);
            __syncthreads();
            __builtin_amdgcn_s_setprio(0);
            compute(buf, cH, cM, cL);                       // phase 2c + 3
            __syncthreads();
            buf = buf == 2 ? 0 : buf + 1;
        }
    }
    __builtin_amdgcn_s_setprio(0);
    PSTAMP();
#undef PSTAMP
}

struct SAdj { f32x4 v[4]; };                 // dAdj: S[m][j0 + 16 kb + 8 h .. + 7], kb = 0, 1
struct SNeg { f32x4 s[4]; f32x4 l[4]; };     // dNeg: S[m0 + 16 kb + 8 h + e][j] (e = 0..7) and lse[m0 + 16 kb + 8 h .. + 7], kb = 0, 1

template <int NTB, bool PP>
__global__ __launch_bounds__(PP ? 512 : 256, PP ? 1 : 2) void lp_grad_b6_kernel(GradB6Args a, int tiles_adj, int tiles_neg) {
    constexpr int TM = PP ? 256 : GB_TM;  // rows per workgroup: 32 per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    __bf16* lds = reinterpret_cast<__bf16*>(gsm);
    const LpDims& D = a.D;
    constexpr float LOG2E = 1.4426950408889634f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
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
    // S of this chunk-direction as a buffer: rows past Bc / columns past the row end are other rows of S or (past the end) zero; all masked
    const int64_t sbytes = ((int64_t)D.ndir * D.Bp - rowbase) * D.n_ld * 4;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, (int)(sbytes < 0x7fffffff ? sbytes : 0x7fffffff), 0x00020000);

    v16f acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (unit < tiles_adj) {
        // ---------------- dAdj tile: rows m0 .. m0 + 127, K = negatives j
        const int m0 = unit * TM;
        const int m = m0 + wave * 32 + l31;
        const bool m_ok = m < D.Bc;
        const float cexp = m_ok ? lg - a.lse[rowbase + m] * LOG2E : -INFINITY;  // -inf switches a padding row off
        const int nchunks = (D.N + GB_KC - 1) / GB_KC;
        const __bf16* bT = a.negT + (int64_t)cd * 3 * a.kp * a.nld;
        const int svoff = (int)(((int64_t)m * D.n_ld + 8 * h) * 4);
        auto loadS = [&](int ch) __attribute__((always_inline)) {
            SAdj s;
            const int chc = ch < nchunks ? ch : nchunks - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)  // i = 2 kb + half
                s.v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, svoff + (i >> 1) * 64 + (i & 1) * 16, chc * GB_KC * 4, 0));
            return s;
        };
        auto makeA = [&](int ch, const SAdj& s, v8bf (&aH)[2], v8bf (&aM)[2], v8bf (&aL)[2]) __attribute__((always_inline)) {
            const bool full = ch * GB_KC + GB_KC <= D.N;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float sv[8] = {s.v[2 * kb][0], s.v[2 * kb][1], s.v[2 * kb][2], s.v[2 * kb][3],
                                     s.v[2 * kb + 1][0], s.v[2 * kb + 1][1], s.v[2 * kb + 1][2], s.v[2 * kb + 1][3]};
                float x[8];
                const int j = ch * GB_KC + 16 * kb + 8 * h;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    x[e] = __builtin_amdgcn_exp2f(fmaf(sv[e], LOG2E, cexp));
                    if (!full && j + e >= D.N) x[e] = 0.f;
                }
                if (GB6_ABLATE & 16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) aH[kb][e] = aM[kb][e] = aL[kb][e] = (__bf16)sv[e];
                } else {
                    split8v(x, aH[kb], aM[kb], aL[kb]);
                }
            }
        };
        if (!PP) gradb6_loop<NTB>(bT, a.kp, a.nld, nchunks, lds, acc, loadS, makeA);
        else if (wave < 4) gradb6_pp_loop<NTB, false>(bT, a.kp, a.nld, nchunks, lds, acc, loadS, makeA, a.dbg);
        else gradb6_pp_loop<NTB, true>(bT, a.kp, a.nld, nchunks, lds, acc, loadS, makeA, a.dbg);
        // lane holds D[row = (r & 3) + 8 (r >> 2) + 4 h][col = l31] of each 32x32 tile
        float* out = a.dadj + rowbase * D.d_ld;
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            const int n = 32 * t + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (mm < D.Bc && n < D.d) out[(int64_t)mm * D.d_ld + n] = acc[t][r];
            }
        }
    } else {
        // ---------------- dNeg tile: rows j0 .. j0 + 127 (negatives), K = batch rows m
        const int j0 = (unit - tiles_adj) * TM;
        const int j = j0 + wave * 32 + l31;
        const int jc = j < D.N ? j : 0;  // rows past N are computed on valid data and never stored
        const int nchunks = (D.Bc + GB_KC - 1) / GB_KC;
        const __bf16* bT = a.adjT + (int64_t)cd * 3 * a.kp * a.bld;
        // lse of this chunk-direction; the (masked) entries past its Bc rows are the next chunk's, or zero past the end of the array
        const __amdgpu_buffer_rsrc_t lrs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lse + rowbase), 0, (int)(((int64_t)D.ndir * D.Bp - rowbase) * 4), 0x00020000);
        const int svoff = (int)(((int64_t)8 * h * D.n_ld + jc) * 4);  // row 8 h of the K block, column j; + (32 ch + 16 kb + e) rows in the SGPR offset
        const int rstride = (int)D.n_ld * 4;
        auto loadS = [&](int ch) __attribute__((always_inline)) {
            SNeg s;
            const int chc = ch < nchunks ? ch : nchunks - 1;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int so = (chc * GB_KC + 16 * kb) * rstride;
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, svoff, so + e * rstride, 0));
                s.s[2 * kb] = (f32x4){t[0], t[1], t[2], t[3]};
                s.s[2 * kb + 1] = (f32x4){t[4], t[5], t[6], t[7]};
                s.l[2 * kb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, 32 * h, (chc * GB_KC + 16 * kb) * 4, 0));
                s.l[2 * kb + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, 32 * h + 16, (chc * GB_KC + 16 * kb) * 4, 0));
            }
            return s;
        };
        auto makeA = [&](int ch, const SNeg& s, v8bf (&aH)[2], v8bf (&aM)[2], v8bf (&aL)[2]) __attribute__((always_inline)) {
            const bool full = ch * GB_KC + GB_KC <= D.Bc;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float sv[8] = {s.s[2 * kb][0], s.s[2 * kb][1], s.s[2 * kb][2], s.s[2 * kb][3],
                                     s.s[2 * kb + 1][0], s.s[2 * kb + 1][1], s.s[2 * kb + 1][2], s.s[2 * kb + 1][3]};
                const float lv[8] = {s.l[2 * kb][0], s.l[2 * kb][1], s.l[2 * kb][2], s.l[2 * kb][3],
                                     s.l[2 * kb + 1][0], s.l[2 * kb + 1][1], s.l[2 * kb + 1][2], s.l[2 * kb + 1][3]};
                float x[8];
                const int mb = ch * GB_KC + 16 * kb + 8 * h;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    x[e] = __builtin_amdgcn_exp2f(fmaf(sv[e], LOG2E, fmaf(-lv[e], LOG2E, lg)));
                    if (!full && mb + e >= D.Bc) x[e] = 0.f;
                }
                if (GB6_ABLATE & 16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) aH[kb][e] = aM[kb][e] = aL[kb][e] = (__bf16)sv[e];
                } else {
                    split8v(x, aH[kb], aM[kb], aL[kb]);
                }
            }
        };
        if (!PP) gradb6_loop<NTB>(bT, a.kp, a.bld, nchunks, lds, acc, loadS, makeA);
        else if (wave < 4) gradb6_pp_loop<NTB, false>(bT, a.kp, a.bld, nchunks, lds, acc, loadS, makeA, nullptr);
        else gradb6_pp_loop<NTB, true>(bT, a.kp, a.bld, nchunks, lds, acc, loadS, makeA, nullptr);
        float* out = a.gocc + (a.negocc_off[dir] + (int64_t)c * D.N) * D.d_ld;
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            const int n = 32 * t + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = j0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
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
    a.dbg = ga.dbg;
    a.D = D;
    const char* ppe = getenv("MARIUS_GB6");
    // MARIUS_GB6=p: the ping-pong kernel (8 waves, 256-row tiles, three LDS buffers).  Measured no faster than the single-group kernel
    // (0.47 vs 0.44 ms incl. transposes): its phase timeline shows the MFMA phase at ~60 cycles per v_mfma_f32_32x32x16_bf16 instead of 32
    // (tools/timeline_gb6.py; tools/micro/mfma_lds_operands.hip reproduces ~46 with alternating operands) — the matrix pipe, not the
    // overlap structure, is what falls short.
    const bool pp = ppe && ppe[0] == 'p';
    const int tm = pp ? 256 : GB_TM;
    const int tiles_adj = (int)cdiv(D.Bc, tm), tiles_neg = (int)cdiv(D.N, tm);
    const int units = tiles_adj + tiles_neg;
    const unsigned grid = (unsigned)(((ncd + 7) / 8) * 8 * units);
    const int ntb = (kp + 31) / 32;
    const size_t lds = (size_t)(pp ? 3 : 2) * 3 * ntb * 32 * GB_RS * sizeof(__bf16);  // 48 kB / 72 kB at kp = 112
#define GB_LAUNCH(NTV)                                                                                                              \
    do {                                                                                                                            \
        if (pp) {                                                                                                                   \
            static bool attr_set = false;                                                                                           \
            if (!attr_set && lds > 65536) {                                                                                         \
                (void)hipFuncSetAttribute((const void*)lp_grad_b6_kernel<NTV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                attr_set = true;                                                                                                    \
            }                                                                                                                       \
            lp_grad_b6_kernel<NTV, true><<<dim3(grid), dim3(512), lds, st>>>(a, tiles_adj, tiles_neg);                              \
        } else {                                                                                                                    \
            lp_grad_b6_kernel<NTV, false><<<dim3(grid), dim3(256), lds, st>>>(a, tiles_adj, tiles_neg);                             \
        }                                                                                                                           \
    } while (0)
    switch (ntb) {
        case 1: GB_LAUNCH(1); break;
        case 2: GB_LAUNCH(2); break;
        default: GB_LAUNCH(4); break;
    }
#undef GB_LAUNCH
    return true;
}

}  // namespace marius
