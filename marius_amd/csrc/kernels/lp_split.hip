// Negative-score contraction on the BF16 matrix pipe with fp32-faithful 3-way operand splitting ("bf16x6").
//
// Why: measured on gfx950 (tools/micro/mfma_valu_overlap*.hip) an FP32 MFMA and VALU work of ANOTHER wave on the same SIMD do not
// overlap at all (time = sum, with or without s_setprio), while a BF16 MFMA and VALU work do (time = max).  The FP32-MFMA score
// kernel therefore pays MFMA cycles + every staging / epilogue / SoftmaxCE VALU cycle, which caps it near 50 % of the FP32 matrix
// peak.  Here every fp32 operand x is split exactly into three bf16 values, x = h + m + l (8 + 8 + 8 significand bits: the two
// remainders are exact in fp32, and the last one has at most 8 significant bits left, so it converts to bf16 without error), and
//   x * y  =  h.h' + (h.m' + m.h') + (h.l' + l.h' + m.m')  +  O(2^-24 |x y|)        (the three dropped products m.l', l.m', l.l')
// is accumulated in fp32 by six v_mfma_f32_32x32x16_bf16 per 16-wide K block: 42 x 32 = 1344 matrix-pipe cycles per 32x32x112 tile
// instead of 50 x 64 = 3200 on the FP32 path, and those cycles now run underneath the VALU work instead of next to it.  Every bf16
// product is exact in fp32 (8 x 8 significand bits), so the result differs from an fp32 FMA chain only by the dropped 2^-24 terms
// and by summation order: the same error class as the FP32-MFMA kernels (parity tests: rtol 1e-4 against the fp32 oracle, and a
// direct comparison with the FP32-MFMA kernel).
//
// Structure = lp_scores_ap_kernel (persistent, adj operand in registers, negative tiles through an LDS ring):
//   * 512 workgroups (2 per CU: 84 VGPRs of adj planes per lane), each walking a contiguous, XCD-local range of
//     (chunk-direction, 128-row tile, 64-column pair) units;
//   * every batch row and every adj row is split ONCE per step by lp_split_rows_kernel into three bf16 planes (embp / adjp in the
//     workspace), so staging a negative tile is pure data movement: three planes [32][120] in LDS; 240-B rows keep the 16-B
//     fragment reads of 8 consecutive rows on disjoint bank groups;
//   * D[n][m] orientation: a lane owns one score row m and 16 columns, so S is stored with 16-B stores and the SoftmaxCE partial
//     (running max / sum exp) is reduced in registers.
#include "lp_common.h"

namespace marius {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 16-B piece as a first-class vector (HIP's uint4 is a struct: copies of arrays of it stay in memory)

constexpr int B6_WAVES = 8, B6_TM = 32 * B6_WAVES, B6_TN = 32, B6_SLOTS = 3, B6_NT = 64 * B6_WAVES;
constexpr int B6_TS = 36;  // row stride (floats) of the per-wave 32x32 transpose buffer: 16-B slots of 8 consecutive rows on disjoint banks

struct Split3 {
    float h, m, l;  // each exactly representable in bf16
};
__device__ __forceinline__ Split3 split3(float x) {
#pragma clang fp contract(off)
    Split3 s;
    const __bf16 h = (__bf16)x;
    s.h = (float)h;
    const float r1 = x - s.h;
    const __bf16 m = (__bf16)r1;
    s.m = (float)m;
    const float r2 = r1 - s.m;
    s.l = r2;  // <= 8 significant bits: the bf16 conversion below is exact
    return s;
}

// eight consecutive floats -> three bf16x8 fragments
__device__ __forceinline__ void split8(const float (&x)[8], v8bf& H, v8bf& M, v8bf& L) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const Split3 s = split3(x[j]);
        H[j] = (__bf16)s.h;
        M[j] = (__bf16)s.m;
        L[j] = (__bf16)s.l;
    }
}

__device__ __forceinline__ v16f mfma_bf16(const v8bf& a, const v8bf& b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// ---- fp32 rows -> three bf16 planes.  One thread per 4 floats: planes[p][row][4 piece .. +3]; columns d .. kp are zero.
__global__ __launch_bounds__(256) void lp_split_rows_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int d, int kp,
                                                            __bf16* __restrict__ planes, int64_t plane_elems) {
    const int ppr = kp >> 2;  // pieces per row
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * ppr) return;
    const int64_t row = idx / ppr;
    const int piece = (int)(idx - row * ppr);
    typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
    v4bf H, M, L;
    if (4 * piece < d) {
        const float4 v = *reinterpret_cast<const float4*>(src + row * ld + 4 * piece);
        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Split3 s3 = split3(x[j]);
            H[j] = (__bf16)s3.h;
            M[j] = (__bf16)s3.m;
            L[j] = (__bf16)s3.l;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) H[j] = M[j] = L[j] = (__bf16)0.f;
    }
    __bf16* o = planes + row * kp + 4 * piece;
    *reinterpret_cast<v4bf*>(o) = H;
    *reinterpret_cast<v4bf*>(o + plane_elems) = M;
    *reinterpret_cast<v4bf*>(o + 2 * plane_elems) = L;
}

int launch_split_rows(const float* src, int64_t ld, int64_t rows, int d, int kp, void* planes, int64_t plane_elems, hipStream_t st) {
    if (rows <= 0) return MARIUS_OK;
    const int64_t n = rows * (kp / 4);
    lp_split_rows_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st>>>(src, ld, rows, d, kp, (__bf16*)planes, plane_elems);
    return check_launch("lp_split_rows");
}

// ---- the contraction.  Operands arrive pre-split (embp / adjp planes), so staging is pure data movement:
// a negative tile = 32 rows x 3 planes x KB*32 B; thread (r = tid / 16, sub = tid % 16) moves the 16-B pieces sub, sub + 16, ... of row r.
// Workgroup = 8 waves x 32 adj rows = 256 rows (one workgroup per CU): every negative tile is fetched from L2 by 4 row tiles per
// chunk instead of 8 — the staging traffic, not the matrix pipe, is what this kernel is bound by (ablation: MFMAs are free).
// S leaves through a per-wave LDS transpose so that every store instruction writes 8 full 128-B lines: the D[n][m] accumulator layout
// gives each lane one row, and 64 lanes x 16 B to 64 different rows cost ~4 cycles per touched line in the store path (ablation:
// 0.13 ms of 0.29).
template <bool L2, int KB>
__global__ __launch_bounds__(B6_NT, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void lp_scores_b6_kernel(ScoreArgs a, int npairs, int mtiles, int total_units, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KP = KB * 16;
    constexpr int RS = KP + 8;               // bf16 elements per LDS row (240 B for KB = 7): 16-B slots of 8 consecutive rows fall on disjoint banks
    constexpr int PLANE = B6_TN * RS;        // bf16 elements per plane
    constexpr int SLOT = 3 * PLANE;          // three planes per ring slot
    constexpr int PPR = 3 * KB * 2;          // 16-B pieces per row over the three planes
    constexpr int TPR = B6_NT / B6_TN;       // threads per staged row (16)
    constexpr int NI = (PPR + TPR - 1) / TPR;  // pieces per thread
    __bf16* lds = reinterpret_cast<__bf16*>(smem_raw);
    float* tbuf = reinterpret_cast<float*>(smem_raw + (size_t)B6_SLOTS * SLOT * sizeof(__bf16));  // [waves][32][B6_TS]
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int per_xcd = nwg >> 3;
    const int wlin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int u0 = (int)((int64_t)wlin * total_units / nwg), u1 = (int)((int64_t)(wlin + 1) * total_units / nwg);
    if (u0 >= u1) return;
    const int ntiles = (D.N + B6_TN - 1) / B6_TN;
    const int ncd = D.C * D.ndir;

    struct Cur { int cd, mt, g; };
    auto advance = [&](Cur& c) {
        if (++c.g == npairs) {
            c.g = 0;
            if (++c.mt == mtiles) { c.mt = 0; ++c.cd; }
        }
    };
    Cur cur;
    {
        const int per_cd = mtiles * npairs;
        cur.cd = u0 / per_cd;
        const int r = u0 - cur.cd * per_cd;
        cur.mt = r / npairs;
        cur.g = r - cur.mt * npairs;
    }
    Cur pre = cur;

    // staging roles
    const int srow_i = tid / TPR, sub = tid % TPR;
    const char* embp = reinterpret_cast<const char*>(a.embp);
    uint32_t goff[NI];   // byte offset of piece i inside (plane, row)
    int loff[NI];        // LDS element offset of piece i
    bool pok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = sub + TPR * i;
        pok[i] = p < PPR;
        const int pc = pok[i] ? p : 0;
        const int plane = pc / (2 * KB), w = pc - plane * (2 * KB);
        goff[i] = (uint32_t)((int64_t)plane * a.embp_plane * 2 + w * 16);
        loff[i] = plane * PLANE + srow_i * RS + w * 8;
    }
    // LDS row tails (KP .. RS) are never read: fragment reads stop at KP.

    struct Tile { u32x4 v[NI]; };  // passed and returned BY VALUE: references to two alternating local arrays end up as a runtime-selected
                                   // pointer (common-code sinking) and push the arrays into scratch
    int64_t id0, id1;
    Tile vb0, vb1;
    auto load_id = [&](const Cur& c, int s_) -> int64_t {
        const int cdc = c.cd < ncd ? c.cd : ncd - 1;  // prefetch past the end of the list re-reads valid rows
        const int dir = cdc / D.C, cc = cdc - dir * D.C;
        const int64_t* negmap = (dir ? a.negmap[1] : a.negmap[0]) + (int64_t)cc * D.N;
        const int n = (2 * c.g + s_) * B6_TN + srow_i;
        return negmap[n < D.N ? n : 0];
    };
    auto issue = [&](int64_t id) __attribute__((always_inline)) {
        Tile vb;
        const char* rp = embp + id * (KP * 2);
#pragma unroll
        for (int i = 0; i < NI; ++i) vb.v[i] = *reinterpret_cast<const u32x4*>(rp + goff[i]);
        return vb;
    };
    auto write = [&](int slot, Tile vb) __attribute__((always_inline)) {  // columns past N hold a valid row; their scores are never stored nor summed
        __bf16* base = lds + (size_t)slot * SLOT;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (pok[i]) *reinterpret_cast<u32x4*>(base + loff[i]) = vb.v[i];
    };

    // ---- prologue: steps 0,1 -> LDS, steps 2,3 in flight (sets 0,1), ids of steps 4,5 loaded
    id0 = load_id(pre, 0);
    id1 = load_id(pre, 1);
    vb0 = issue(id0);
    vb1 = issue(id1);
    advance(pre);
    id0 = load_id(pre, 0);
    id1 = load_id(pre, 1);
    write(0, vb0);
    write(1, vb1);
    vb0 = issue(id0);
    vb1 = issue(id1);
    advance(pre);
    id0 = load_id(pre, 0);
    id1 = load_id(pre, 1);
    advance(pre);  // pre = u0 + 3: the next ids to fetch

    // per row-tile state: adj fragments of this lane, A[m_row][16 blk + 8 h .. +7], three planes
    v8bf aH[KB], aM[KB], aL[KB];
    float* srow = nullptr;
    float* swave = nullptr;
    int rows_ok = 0;
    float xx = 0.f;
    bool m_ok = false;
    int64_t rowbase = 0;
    int m_row = 0, dirc = 0, cc = 0;
    auto load_tile = [&](const Cur& c) {
        dirc = c.cd / D.C;
        cc = c.cd - dirc * D.C;
        rowbase = (int64_t)dirc * D.Bp + (int64_t)cc * D.Bc;
        m_row = c.mt * B6_TM + wave * 32 + l31;
        m_ok = m_row < D.Bc;
        const __bf16* ap = reinterpret_cast<const __bf16*>(a.adjp) + (rowbase + (m_ok ? m_row : 0)) * KP + 8 * h;
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
            aH[blk] = *reinterpret_cast<const v8bf*>(ap + 16 * blk);
            aM[blk] = *reinterpret_cast<const v8bf*>(ap + 16 * blk + a.adjp_plane);
            aL[blk] = *reinterpret_cast<const v8bf*>(ap + 16 * blk + 2 * a.adjp_plane);
        }
        srow = a.S + (rowbase + (m_ok ? m_row : 0)) * D.n_ld;
        rows_ok = D.Bc - (c.mt * B6_TM + wave * 32);  // valid rows of this wave's 32-row block (may be <= 0 or >= 32)
        swave = a.S + (rowbase + c.mt * B6_TM + wave * 32 + (lane >> 3)) * D.n_ld + 4 * (lane & 7);
        if (L2) xx = m_ok ? a.x2[rowbase + m_row] : 0.f;
        // Land the fragments HERE.  Otherwise their first use — the MFMAs at the top of the next step — carries the wait, and as
        // these loads are the youngest in the in-order vmcnt queue that wait is vmcnt(0) on EVERY step (the state is merged
        // conservatively at the loop header): it would drain the negative-tile prefetch each step.
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) asm volatile("" : "+v"(aH[blk]), "+v"(aM[blk]), "+v"(aL[blk]));
    };
    load_tile(cur);
    __syncthreads();

    float run_m = -3.0e38f, run_l = 0.f;
    int slot = 0;  // LDS slot of the current step; the step two ahead goes to (slot + 2) % 3
    constexpr float LOG2E = 1.4426950408889634f;

    auto step = [&](int t, int64_t id, Tile vb) __attribute__((always_inline)) {
        // on entry: LDS holds this step and the next; `vb` holds the step two ahead; `id` the row id of the step four ahead
        const __bf16* bp = lds + (size_t)slot * SLOT + l31 * RS + 8 * h;
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        v8bf bH[2], bM[2], bL[2];
        bH[0] = *reinterpret_cast<const v8bf*>(bp);
        bM[0] = *reinterpret_cast<const v8bf*>(bp + PLANE);
        bL[0] = *reinterpret_cast<const v8bf*>(bp + 2 * PLANE);
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
            const int c_ = blk & 1, n_ = c_ ^ 1;
            if (blk + 1 < KB) {
                bH[n_] = *reinterpret_cast<const v8bf*>(bp + 16 * (blk + 1));
                bM[n_] = *reinterpret_cast<const v8bf*>(bp + 16 * (blk + 1) + PLANE);
                bL[n_] = *reinterpret_cast<const v8bf*>(bp + 16 * (blk + 1) + 2 * PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (a.ablate & 2) continue;  // debug only (MARIUS_ABLATE)
            // D[n][m] += neg[n][k] * adj[m][k]; smallest products first
            acc = mfma_bf16(bL[c_], aH[blk], acc);
            acc = mfma_bf16(bH[c_], aL[blk], acc);
            acc = mfma_bf16(bM[c_], aM[blk], acc);
            acc = mfma_bf16(bM[c_], aH[blk], acc);
            acc = mfma_bf16(bH[c_], aM[blk], acc);
            acc = mfma_bf16(bH[c_], aH[blk], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int wslot = slot >= 1 ? slot - 1 : 2;  // (slot + 2) % 3, last read one step ago
        if (!(a.ablate & 4)) {
            write(wslot, vb);
            vb = issue(id);
        }
        slot = slot == 2 ? 0 : slot + 1;
        if (t < ntiles) {
            const int nb = t * B6_TN + 4 * h;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[r];
                if (L2) {
#pragma clang fp contract(off)
                    const int n = nb + 8 * (r >> 2) + (r & 3);
                    const float yy = (n < D.N) ? a.y2[(int64_t)dirc * D.C * D.N + (int64_t)cc * D.N + n] : 0.f;
                    const float tt = (xx + yy) - 2.f * v[r];
                    v[r] = sqrtf(fmaxf(tt, 1e-8f));
                }
            }
            if ((t + 1) * B6_TN <= D.N) {  // every column of this tile is a real negative: no per-element guards
                if (!(a.ablate & 1)) {
                    float* tb = tbuf + wave * (32 * B6_TS);
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(tb + l31 * B6_TS + 8 * q + 4 * h) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    // same wave writes and reads: LDS executes a wave's instructions in order, no barrier needed
                    float* sw = swave + (int64_t)t * B6_TN;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 x = *reinterpret_cast<const float4*>(tb + (8 * i + (lane >> 3)) * B6_TS + 4 * (lane & 7));
                        if (8 * i + (lane >> 3) < rows_ok) *reinterpret_cast<float4*>(sw + (int64_t)(8 * i) * D.n_ld) = x;
                    }
                }
                if (a.lse_part && !(a.ablate & 8)) {
                    float tmax = v[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, v[r]);
                    const float mnew = fmaxf(run_m, tmax);
                    const float cs = -mnew * LOG2E;
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(fmaf(v[r], LOG2E, cs));
                    run_l = run_l * __builtin_amdgcn_exp2f(fmaf(run_m, LOG2E, cs)) + sum;
                    run_m = mnew;
                }
            } else {
                if (m_ok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = nb + 8 * (r >> 2) + (r & 3);
                        if (n < D.N) srow[n] = v[r];
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
        }
        __syncthreads();
        return vb;
    };

    for (int u = u0; u < u1; ++u) {
        vb0 = step(2 * cur.g, id0, vb0);
        id0 = load_id(pre, 0);
        vb1 = step(2 * cur.g + 1, id1, vb1);
        id1 = load_id(pre, 1);
        advance(pre);
        if (a.lse_part) {
            const float m2 = __shfl_xor(run_m, 32, 64), l2 = __shfl_xor(run_l, 32, 64);
            const float mm = fmaxf(run_m, m2);
            const float ll = run_l * __expf(run_m - mm) + l2 * __expf(m2 - mm);
            if (h == 0 && m_ok) {
                float* out = a.lse_part + ((int64_t)cur.g * D.ndir * D.Bp + rowbase + m_row) * 2;
                out[0] = mm;
                out[1] = ll;
            }
            run_m = -3.0e38f;
            run_l = 0.f;
        }
        advance(cur);
        if (cur.g == 0 && u + 1 < u1) load_tile(cur);  // row-tile seam: new adj fragments (the negative-tile pipeline keeps running)
    }
}

static bool split_ok(const float* emb, int64_t emb_ld, int d) {
    const bool aligned = (d % 4 == 0) && (emb_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(emb) & 15) == 0);
    return aligned && (d == 32 || d == 64 || d == 100 || d == 128);
}

bool scores_b6_applicable(const float* emb, int64_t emb_ld, int d) { return split_ok(emb, emb_ld, d); }

bool launch_scores_b6(const ScoreArgs& a, bool l2, hipStream_t st) {
    if (!split_ok(a.emb, a.emb_ld, a.D.d) || !a.embp || !a.adjp) return false;
    if (a.embp_plane * 6 >= (int64_t)1 << 31) return false;  // 32-bit piece offsets inside the plane buffer
    const int npairs = scores_ap_groups(a.D.N);
    const int mtiles = (int)cdiv(a.D.Bc, B6_TM);
    const int total = a.D.C * a.D.ndir * mtiles * npairs;
    const int nwg = 256;  // one 8-wave workgroup per CU, all resident
    const int kb = (a.D.d + 15) / 16;
    const size_t lds = (size_t)B6_SLOTS * 3 * B6_TN * (kb * 16 + 8) * sizeof(__bf16) + (size_t)B6_WAVES * 32 * B6_TS * sizeof(float);
#define B6_LAUNCH(L2V, KBV)                                                                                                         \
    do {                                                                                                                            \
        static bool attr_set = false;                                                                                               \
        if (!attr_set && lds > 65536) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)lp_scores_b6_kernel<L2V, KBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                                        \
        }                                                                                                                           \
        lp_scores_b6_kernel<L2V, KBV><<<dim3(nwg), dim3(B6_NT), lds, st>>>(a, npairs, mtiles, total, nwg);                            \
    } while (0)
#define B6_DISPATCH(L2V)                     \
    do {                                     \
        switch (kb) {                        \
            case 2: B6_LAUNCH(L2V, 2); break; \
            case 4: B6_LAUNCH(L2V, 4); break; \
            case 7: B6_LAUNCH(L2V, 7); break; \
            default: B6_LAUNCH(L2V, 8); break; \
        }                                    \
    } while (0)
    if (l2)
        B6_DISPATCH(true);
    else
        B6_DISPATCH(false);
#undef B6_DISPATCH
#undef B6_LAUNCH
    return true;
}

}  // namespace marius
