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
#include <cstdlib>

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
// =========================================================================================== scores, persistent workgroups
// Timeline stamps of lp_scores_res_kernel (tools/timeline_scores.py) showed that a workgroup spends ~8.3k cycles per 64x64 tile
// (3.7k of them in the MFMA chain) but ~24k cycles per 4-tile unit in launch + prologue (negative ids -> rows -> LDS is two
// dependent HBM round trips) — 40 % of the kernel.  Here 2 workgroups per CU stay resident and walk a static list of units
// that belong to their XCD; the tile stream is continuous across unit seams (the first negative tile and the adj tile of the
// next unit are prefetched like any other tile), so the prologue is paid once per workgroup instead of once per unit.
struct PSDesc {
    bool valid, new_unit, last_of_unit;
    int dir, c, m0, ntile, ng;
};

template <bool L2, int NQ>
__global__ __launch_bounds__(256, 2) void lp_scores_ps_kernel(ScoreArgs a, int ngroups, int ntpg, int units_per_cd, int wg_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int KS = a.KS;
    float* As = smem;
    float* Bs0 = smem + R_T * KS;
    float* Bs1 = Bs0 + R_T * KS;
    float* red = Bs1 + R_T * KS;  // 256 floats for the SoftmaxCE partial exchange
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, W = wg_per_xcd;
    const int ncd = D.C * D.ndir;
    const int ncd_x = (ncd - xcd + 7) / 8;
    const int total_units = ncd_x * units_per_cd;
    const int my_units = w < total_units ? (total_units - w + W - 1) / W : 0;
    const int K = my_units * ntpg;
    if (K == 0) return;
    const int ntiles = (D.N + R_T - 1) / R_T;

    // step descriptors are advanced incrementally (the divisions run once per unit, not four times per tile)
    auto unit_fields = [&](PSDesc& d, int ui) {
        const int g = w + ui * W;
        const int cdi = g / units_per_cd, unit = g - cdi * units_per_cd;
        const int cd = xcd + 8 * cdi;
        d.dir = cd / D.C;
        d.c = cd - d.dir * D.C;
        const int mt = unit / ngroups;
        d.ng = unit - mt * ngroups;
        d.m0 = mt * R_T;
    };
    int adv_ui = 0, adv_ti = -1;  // position of the most recently produced descriptor
    PSDesc adv_d;
    adv_d.valid = adv_d.new_unit = adv_d.last_of_unit = false;
    adv_d.dir = adv_d.c = adv_d.m0 = adv_d.ntile = adv_d.ng = 0;
    auto next_desc = [&]() {
        ++adv_ti;
        if (adv_ti == ntpg) {
            adv_ti = 0;
            ++adv_ui;
        }
        const bool in_range = adv_ui < my_units;
        if (adv_ti == 0 && in_range) unit_fields(adv_d, adv_ui);
        adv_d.ntile = adv_d.ng * ntpg + adv_ti;
        adv_d.valid = in_range && (adv_d.ntile < ntiles);
        adv_d.new_unit = in_range && (adv_ti == 0);
        adv_d.last_of_unit = in_range && (adv_ti == ntpg - 1);
        return adv_d;
    };

    const int piece = tid & 31, row = tid >> 5;
    const bool col_ok = 4 * piece < D.d;
    const int colc = col_ok ? 4 * piece : 0;
    float4 va[8], vb[8];
    int64_t ids[8];
    auto load_ids = [&](const PSDesc& d) {
        const int64_t* negmap = a.negmap[d.dir] + (int64_t)d.c * D.N;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int n = d.ntile * R_T + row + 8 * it;
            ids[it] = negmap[(d.valid && n < D.N) ? n : 0];
        }
    };
    auto issue_b = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it) vb[it] = *reinterpret_cast<const float4*>(a.emb + ids[it] * a.emb_ld + colc);
    };
    auto issue_a = [&](const PSDesc& d) {
        const float* adj = a.adj + ((int64_t)d.dir * D.Bp + (int64_t)d.c * D.Bc) * D.d_ld;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = d.m0 + row + 8 * it;
            va[it] = *reinterpret_cast<const float4*>(adj + (int64_t)(m < D.Bc ? m : 0) * D.d_ld + colc);
        }
    };
    auto write_b = [&](float* buf, const PSDesc& d) {
        if (col_ok) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int n = d.ntile * R_T + row + 8 * it;
                lds_store4x(buf + (row + 8 * it) * KS + 4 * piece, mul4(vb[it], n < D.N ? 1.f : 0.f));
            }
        }
    };
    auto write_a = [&](const PSDesc& d) {
        if (col_ok) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = d.m0 + row + 8 * it;
                lds_store4x(As + (row + 8 * it) * KS + 4 * piece, mul4(va[it], m < D.Bc ? 1.f : 0.f));
            }
        }
    };

    // ---- prologue: step 0 in LDS, step 1 in flight, ids of step 2 loaded
    PSDesc q0 = next_desc(), q1 = next_desc(), q2 = next_desc(), q3 = next_desc();  // steps k, k+1, k+2, k+3
    {
        const PSDesc &d0 = q0, &d1 = q1, &d2 = q2;
        load_ids(d0);
        issue_a(d0);
        issue_b();
        write_a(d0);
        write_b(Bs0, d0);
        load_ids(d1);
        issue_b();
        if (d1.new_unit) issue_a(d1);
        load_ids(d2);
    }
    __syncthreads();

    const float* ap = As + (wm * 32 + l31) * KS + 2 * h;
    const int nq = D.d >> 2;
    float run_m = -3.0e38f, run_l = 0.f;
    for (int k = 0; k < K; ++k) {
        const PSDesc dk = q0, dn = q1, d2 = q2, d3 = q3;
        q0 = q1;
        q1 = q2;
        q2 = q3;
        q3 = next_desc();
        const float* bp = ((k & 1) ? Bs1 : Bs0) + (wn * 32 + l31) * KS + 2 * h;
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (dk.valid) {
            if constexpr (NQ > 0) {
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
                    acc = mfma32(bv[q & 1].x, av[q & 1].x, acc);
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
        }
        // step k+1's negative tile (in vb since the previous step) -> the other LDS buffer; step k+2's rows into flight
        if (dn.valid) write_b((k & 1) ? Bs0 : Bs1, dn);
        // (va may still hold step k+1's adj tile: it is written to LDS after the barrier below, before any new issue_a)
        float4 va_keep[8];
        const bool seam = dn.new_unit;
        if (seam) {
#pragma unroll
            for (int it = 0; it < 8; ++it) va_keep[it] = va[it];
        }
        if (d2.valid) {
            issue_b();
            if (d2.new_unit) issue_a(d2);
        }
        load_ids(d3);
        // epilogue of step k
        if (dk.valid) {
            const int m = dk.m0 + wm * 32 + l31;
            const int nb = dk.ntile * R_T + wn * 32 + 4 * h;
            float* S = a.S + ((int64_t)dk.dir * D.Bp + (int64_t)dk.c * D.Bc) * D.n_ld;
            float xx = 0.f;
            if (L2 && m < D.Bc) xx = a.x2[(int64_t)dk.dir * D.Bp + (int64_t)dk.c * D.Bc + m];
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[r];
                if (L2) {
#pragma clang fp contract(off)
                    const int n = nb + 8 * (r >> 2) + (r & 3);
                    const float yy = (n < D.N) ? a.y2[(int64_t)dk.dir * D.C * D.N + (int64_t)dk.c * D.N + n] : 0.f;
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
        // end of a unit: publish the unit's SoftmaxCE partial (two half-waves, then the two waves that share the rows)
        const bool unit_end = dk.last_of_unit;
        if (a.lse_part && unit_end) {
            const float m2 = __shfl_xor(run_m, 32, 64), l2 = __shfl_xor(run_l, 32, 64);
            const float mm = fmaxf(run_m, m2);
            const float ll = run_l * __expf(run_m - mm) + l2 * __expf(m2 - mm);
            if (h == 0) {
                red[((wm * 2 + wn) * 32 + l31) * 2] = mm;
                red[((wm * 2 + wn) * 32 + l31) * 2 + 1] = ll;
            }
            run_m = -3.0e38f;
            run_l = 0.f;
        }
        __syncthreads();
        if (a.lse_part && unit_end) {
            const int m = dk.m0 + wm * 32 + l31;
            if (wn == 0 && h == 0 && m < D.Bc) {
                const float ma = red[((wm * 2) * 32 + l31) * 2], la = red[((wm * 2) * 32 + l31) * 2 + 1];
                const float mb = red[((wm * 2 + 1) * 32 + l31) * 2], lb = red[((wm * 2 + 1) * 32 + l31) * 2 + 1];
                const float mx = fmaxf(ma, mb);
                float* out = a.lse_part + ((int64_t)dk.ng * D.ndir * D.Bp + (int64_t)dk.dir * D.Bp + (int64_t)dk.c * D.Bc + m) * 2;
                out[0] = mx;
                out[1] = la * __expf(ma - mx) + lb * __expf(mb - mx);
            }
        }
        if (seam) {  // next step starts a new unit: its adj tile replaces As now that every wave is past the barrier
            if (col_ok) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = dn.m0 + row + 8 * it;
                    lds_store4x(As + (row + 8 * it) * KS + 4 * piece, mul4(va_keep[it], m < D.Bc ? 1.f : 0.f));
                }
            }
            __syncthreads();
        }
    }
}

// =========================================================================================== scores, adj fragments in registers
// Lessons from the timeline / ablation runs above: (1) the random 400-B row gathers need ~2-4 us under load, more than one
// tile period, (2) two co-resident workgroups with the same phase pattern fall into lockstep.  This variant therefore
//   * keeps each wave's 32 adj rows as MFMA fragments in REGISTERS for the whole unit (no adj tile in LDS, half the LDS reads),
//   * multiplies 128 rows x 32 negatives per step: one 32-row negative tile (12.8 kB) feeds all four waves, i.e. half the
//     gather traffic per MFMA of the 64x64 variant,
//   * prefetches four tiles deep: two register sets in flight (issued two steps before they are written to LDS) and a 3-slot
//     LDS ring (written two steps before it is read),
//   * fits 3 workgroups per CU (40 kB LDS, <= 168 VGPRs), which breaks the two-workgroup lockstep.
// Each wave owns its rows for every column, so the SoftmaxCE partial needs no cross-wave exchange.
constexpr int A_TM = 128, A_TN = 32, A_SLOTS = 3;

// =========================================================================================== scores, adj in registers, persistent
// lp_scores_a_kernel spends 28 % of a workgroup's life in its prologue (ids -> rows -> LDS, adj fragments) for only four 32-column
// steps, and 6400 equal workgroups over 768 resident slots quantise to 8.33 -> 9 rounds.  Here 768 workgroups (3 per CU, all
// resident) each walk a contiguous range of the global unit list in (chunk-direction, row tile, column pair) order, a unit being two
// 32-column steps: one prologue per workgroup, the negative-tile pipeline (ids six steps ahead, rows four, LDS two) runs straight
// through unit and tile seams, and the ranges are balanced to +-1 unit (12800 units / 768 = 16.67).  Only the adj fragments are
// reloaded at a row-tile seam.  Ranges are contiguous per XCD (block b runs on XCD b % 8), so a chunk's negative rows stay in one
// L2.  SoftmaxCE partials are flushed per unit: lse_part[row][npairs][2].
template <bool L2, int NQ>
__global__ __launch_bounds__(256, 3) void lp_scores_ap_kernel(ScoreArgs a, int npairs, int mtiles, int total_units, int nwg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LpDims& D = a.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int per_xcd = nwg >> 3;
    const int wlin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int u0 = (int)((int64_t)wlin * total_units / nwg), u1 = (int)((int64_t)(wlin + 1) * total_units / nwg);
    if (u0 >= u1) return;
    const int ntiles = (D.N + A_TN - 1) / A_TN;
    const int ncd = D.C * D.ndir;
    const int KS = a.KS;

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

    const int piece = tid & 31, row = tid >> 5;  // 32 thr x 16 B per row, 8 rows per pass, 4 passes = 32 rows
    const bool col_ok = 4 * piece < D.d;
    const int colc = col_ok ? 4 * piece : 0;
    int64_t ids0[4], ids1[4];
    float4 vb0[4], vb1[4];
    auto load_ids = [&](const Cur& c, int s_, int64_t(&ids)[4]) {
        const int cdc = c.cd < ncd ? c.cd : ncd - 1;  // prefetch past the end of the list re-reads valid rows
        const int dir = cdc / D.C, cc = cdc - dir * D.C;
        const int64_t* negmap = (dir ? a.negmap[1] : a.negmap[0]) + (int64_t)cc * D.N;
        const int t = 2 * c.g + s_;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int n = t * A_TN + row + 8 * it;
            ids[it] = negmap[n < D.N ? n : 0];
        }
    };
    auto issue = [&](const int64_t(&ids)[4], float4(&vb)[4]) {
#pragma unroll
        for (int it = 0; it < 4; ++it) vb[it] = *reinterpret_cast<const float4*>(a.emb + ids[it] * a.emb_ld + colc);
    };
    auto write = [&](int slot, const float4(&vb)[4]) {  // columns past N hold a valid row; their scores are never stored nor summed
        if (col_ok) {
            float* buf = smem + slot * (A_TN * KS);
#pragma unroll
            for (int it = 0; it < 4; ++it) lds_store4x(buf + (row + 8 * it) * KS + 4 * piece, vb[it]);
        }
    };

    // ---- prologue: steps 0,1 -> LDS, steps 2,3 in flight (sets 0,1), ids of steps 4,5 loaded
    load_ids(pre, 0, ids0);
    load_ids(pre, 1, ids1);
    issue(ids0, vb0);
    issue(ids1, vb1);
    advance(pre);
    load_ids(pre, 0, ids0);
    load_ids(pre, 1, ids1);
    write(0, vb0);
    write(1, vb1);
    issue(ids0, vb0);
    issue(ids1, vb1);
    advance(pre);
    load_ids(pre, 0, ids0);
    load_ids(pre, 1, ids1);
    advance(pre);  // pre = u0 + 3: the next ids to fetch

    // per row-tile state
    float2 af[NQ];
    float* srow = nullptr;
    float xx = 0.f;
    bool m_ok = false;
    int64_t rowbase = 0;
    int m_row = 0, dirc = 0, cc = 0;
    auto load_tile = [&](const Cur& c) {
        dirc = c.cd / D.C;
        cc = c.cd - dirc * D.C;
        rowbase = (int64_t)dirc * D.Bp + (int64_t)cc * D.Bc;
        m_row = c.mt * A_TM + wave * 32 + l31;
        m_ok = m_row < D.Bc;
        const float* arow = a.adj + (rowbase + (m_ok ? m_row : 0)) * D.d_ld + 2 * h;
#pragma unroll
        for (int q = 0; q < NQ; ++q) af[q] = *reinterpret_cast<const float2*>(arow + 4 * q);
        if (!m_ok) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) af[q] = make_float2(0.f, 0.f);
        }
        srow = a.S + (rowbase + (m_ok ? m_row : 0)) * D.n_ld;
        if (L2) xx = m_ok ? a.x2[rowbase + m_row] : 0.f;
        // land the fragments here: a wait at their first use inside the step loop would be vmcnt(0) on every step (see lp_split.hip)
#pragma unroll
        for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(af[q].x), "+v"(af[q].y));
    };
    load_tile(cur);
    {
        // All 768 workgroups start together, so the three that share a CU would run their MFMA / store phases in lockstep and the
        // phases would add up instead of overlapping.  Skew them by a third of a step period each, keyed on the hardware wave slot
        // of wave 0 (HW_ID[3:0]; co-resident waves of one SIMD hold different slots).
        __shared__ int phase_s;
        if (tid == 0) phase_s = (int)(__builtin_amdgcn_s_getreg(0x1804) % 3u);
        __syncthreads();
        const int ph = a.ablate & 32 ? 0 : phase_s;
        for (int k = 0; k < ph; ++k) __builtin_amdgcn_s_sleep(60);
    }
    __syncthreads();

    float run_m = -3.0e38f, run_l = 0.f;
    int slot = 0;  // LDS slot of the current step; the step two ahead goes to (slot + 2) % 3

    auto step = [&](int t, const int64_t(&ids)[4], float4(&vb)[4]) {
        // on entry: LDS holds this step and the next; `vb` holds the step two ahead; `ids` the ids of the step four ahead
        const float* bp = smem + slot * (A_TN * KS) + l31 * KS + 2 * h;
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float2 bv[2];
        bv[0] = *reinterpret_cast<const float2*>(bp);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) bv[(q + 1) & 1] = *reinterpret_cast<const float2*>(bp + 4 * (q + 1));
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32(bv[q & 1].x, af[q].x, acc);  // D[n][m]: the lane owns row m and 16 columns n
            acc = mfma32(bv[q & 1].y, af[q].y, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int wslot = slot >= 1 ? slot - 1 : 2;  // (slot + 2) % 3, last read one step ago
        write(wslot, vb);
        issue(ids, vb);
        slot = slot == 2 ? 0 : slot + 1;
        if (t < ntiles) {
            const int nb = t * A_TN + 4 * h;
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
            if ((t + 1) * A_TN <= D.N) {
                // every column of this tile is a real negative (uniform test): no per-element guards.  FP32 MFMAs and VALU
                // instructions exclude each other on a SIMD, so the epilogue is kept to ~5 instructions per score
                if (m_ok) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(srow + nb + 8 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                }
                if (a.lse_part) {
                    constexpr float LOG2E = 1.4426950408889634f;
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
    };

    for (int u = u0; u < u1; ++u) {
        step(2 * cur.g, ids0, vb0);
        load_ids(pre, 0, ids0);
        step(2 * cur.g + 1, ids1, vb1);
        load_ids(pre, 1, ids1);
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
        if (cur.g == 0 && u + 1 < u1) load_tile(cur);  // row-tile seam: new adj fragments (the B pipeline keeps running)
    }
}

// =========================================================================================== backward contractions, 16x16x4 tiles
#ifndef GRAD_ABLATE
#define GRAD_ABLATE 0  // experiment builds only (tools/ablate_grad16.sh): 2 = VALU stand-in for the MFMAs, 4 = no LDS staging writes, 8 = no global loads after the prologue
#endif
constexpr int H_TM = 64;          // output rows per workgroup (4 waves x 16)
constexpr int H_KC = 32;          // K chunk
constexpr int H_NT = 8;           // up to 8 x 16 = 128 output columns per n-block
constexpr int H_QS_MK = H_KC + 4; // V staged [m][k] (grad_adj): 16-B aligned rows
constexpr int H_QS_KM = H_TM + 4; // V staged [k][m] (grad_neg)
constexpr int H_QSZ = (H_TM * H_QS_MK > H_KC * H_QS_KM) ? H_TM * H_QS_MK : H_KC * H_QS_KM;
constexpr int H_MAXIDS = 2048;    // negative-row indices of one chunk kept in LDS as int32 (N <= 2048 on this path)
// dynamic LDS layout (floats): Qs[2][H_QSZ] | Bs[2][H_KC * (16 NT + 4)] | sums[H_TM] | ids[H_MAXIDS] (grad_adj only, sized by N)
static inline size_t grad16_lds_bytes(int N, int nt) { return (size_t)(2 * H_QSZ + 2 * H_KC * (16 * nt + 4) + H_TM + ((N + 31) / 32) * 32) * sizeof(float); }

// Both backward contractions stream K in chunks of 32 through a double-buffered LDS ring with ONE barrier per chunk and
// a two-chunk-deep register prefetch (register sets 0/1 alternate, so every global load has two chunk periods to land).

// dAdj_c[m, n] = sum_j V[m, j] * Neg_c[j, n]
template <bool L2, int NT>
__device__ __forceinline__ void grad_adj16_body(const GradArgs& a, int cd, int unit, int tiles_m, float* smem, int cbeg = 0, int cend = -1, float* part = nullptr) {
    constexpr int BS = 16 * NT + 4;  // B staged [k][n]: stride % 8 == 4 keeps the four k-rows of an MFMA B fragment on disjoint banks
    float(*Qs)[H_QSZ] = reinterpret_cast<float(*)[H_QSZ]>(smem);
    float(*Bs)[H_KC * BS] = reinterpret_cast<float(*)[H_KC * BS]>(smem + 2 * H_QSZ);
    float* rsum = smem + 2 * H_QSZ + 2 * H_KC * BS;
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

    unsigned long long* dbg = (a.dbg && blockIdx.x < 2048 && (blockIdx.x & 7) == 0 && lane == 0 && (wave == 0 || wave == 3))
                                  ? a.dbg + ((size_t)(blockIdx.x >> 3) * 2 + (wave ? 1 : 0)) * 64 : nullptr;
    int dbi = 0;
#define GSTAMP() do { if (dbg && dbi < 64) dbg[dbi++] = __builtin_readcyclecounter(); } while (0)
    GSTAMP();
    const int nchunks = (D.N + H_KC - 1) / H_KC;
    for (int j = tid; j < nchunks * H_KC; j += 256) idl[j] = j < D.N ? (int)negmap[j] : 0;  // batch-local row ids of this chunk's negatives (tail: a valid row)

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
    // Dot: dL/dS = gscale * exp(S - lse) = exp2(S * log2e + c), c = log2(gscale) - lse * log2e; c = -inf switches a padding row off.
    // FP32 MFMAs and VALU instructions exclude each other on a SIMD (tools/micro/mfma_valu_overlap*.hip), so every VALU instruction of
    // the staging code is paid in full: keep the per-element work at two instructions and the masks out of the common path.
    constexpr float LOG2E = 1.4426950408889634f;
    float cexp[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        cexp[it] = qmask[it] != 0.f ? __log2f(D.gscale) - lse_r[it] * LOG2E : -INFINITY;
        asm volatile("" : "+v"(cexp[it]), "+v"(lse_r[it]));  // keep in registers (the compiler otherwise re-loads lse every chunk)
    }
    if (!col_ok && 4 * bpiece < 16 * NT) {  // K-padding columns of the B tile: zero once, the staging of real columns never touches them
        for (int r = brow; r < 2 * H_KC; r += 8) *reinterpret_cast<float4*>(&Bs[0][r * BS + 4 * bpiece]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();  // idl visible

    // issue/write are branch-free on purpose: with a conditional early-out the compiler's waitcnt pass merges the two paths
    // conservatively and drains vmcnt to 0 in every write phase, which collapses the two-chunk prefetch distance to one
    auto issue = [&](int ch, float4(&vs)[2], float4(&vb)[4]) {
        ch = ch < nchunks ? ch : nchunks - 1;
        const int j0 = ch * H_KC;
        const int j = j0 + 4 * qpiece;
#pragma unroll
        for (int it = 0; it < 2; ++it) vs[it] = *reinterpret_cast<const float4*>(srow[it] + (j < D.N ? j : 0));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int id = idl[j0 + brow + 8 * it];
            vb[it] = *reinterpret_cast<const float4*>(a.emb + (int64_t)id * a.emb_ld + ncol_c);
        }
    };
    auto write = [&](int buf, int ch, const float4(&vs)[2], const float4(&vb)[4]) {
        const int j0 = ch * H_KC;
        const int j = j0 + 4 * qpiece;
        if (!L2 && j0 + H_KC <= D.N) {  // common case (uniform): a full chunk of real negatives, no masks
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                float4 v;
                v.x = __builtin_amdgcn_exp2f(fmaf(vs[it].x, LOG2E, cexp[it]));
                v.y = __builtin_amdgcn_exp2f(fmaf(vs[it].y, LOG2E, cexp[it]));
                v.z = __builtin_amdgcn_exp2f(fmaf(vs[it].z, LOG2E, cexp[it]));
                v.w = __builtin_amdgcn_exp2f(fmaf(vs[it].w, LOG2E, cexp[it]));
                *reinterpret_cast<float4*>(&Qs[buf][(qrow + 32 * it) * H_QS_MK + 4 * qpiece]) = v;
            }
            if (col_ok) {
#pragma unroll
                for (int it = 0; it < 4; ++it) *reinterpret_cast<float4*>(&Bs[buf][(brow + 8 * it) * BS + 4 * bpiece]) = vb[it];
            }
            return;
        }
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
            if (4 * bpiece < 16 * NT) *reinterpret_cast<float4*>(&Bs[buf][r * BS + 4 * bpiece]) = v;
        }
    };
    auto compute = [&](int buf) {
        const float* qp = &Qs[buf][(wave * 16 + l15) * H_QS_MK + 4 * kq];
        const float* bp = &Bs[buf][(4 * kq) * BS + l15];
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
                const float* brow_p = bp + (16 * sn + en) * BS;
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = brow_p[16 * t];
                if (en == 0) a4n = *reinterpret_cast<const float4*>(qp + 16 * sn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (!(GRAD_ABLATE & 2)) acc[t] = mfma16(av, bc[t], acc[t]); else acc[t][0] += av * bc[t];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) bc[t] = bn[t];
            a4 = a4n;
            (void)s_;
        }
    };

    // K range of this call: chunks [cbeg, cend) (the stream-K launch gives a workgroup a slice of a tile's K loop)
    const int cend_ = cend < 0 ? nchunks : cend;
    float4 vs0[2], vb0[4], vs1[2], vb1[4];
    issue(cbeg, vs0, vb0);
    issue(cbeg + 1, vs1, vb1);
    write(0, cbeg, vs0, vb0);
    issue(cbeg + 2, vs0, vb0);
    __syncthreads();
    GSTAMP();
    for (int ch = cbeg; ch < cend_; ch += 2) {
        compute(0);                       // chunk ch
        if (ch < 12) GSTAMP();
        if (!(GRAD_ABLATE & 4)) write(1, ch + 1, vs1, vb1);
        if (ch < 12) GSTAMP();
        if (!(GRAD_ABLATE & 8)) issue(ch + 3, vs1, vb1);
        if (ch < 12) GSTAMP();
        __syncthreads();
        if (ch < 12) GSTAMP();
        if (ch + 1 < cend_) compute(1);  // chunk ch + 1
        if (!(GRAD_ABLATE & 4)) write(0, ch + 2, vs0, vb0);
        if (!(GRAD_ABLATE & 8)) issue(ch + 4, vs0, vb0);
        __syncthreads();
        if (ch < 12) GSTAMP();
    }
    GSTAMP();
    if (part) {  // partial K range: park the accumulators, the fix-up kernel adds the other part and stores the tile
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(t * 4 + r) * 256 + tid] = acc[t][r];
        return;
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
    GSTAMP();
#undef GSTAMP
}

// dNeg_c[m, n] = sum_i V[i, m] * adj_c[i, n]
template <bool L2, int NT>
__device__ __forceinline__ void grad_neg16_body(const GradArgs& a, int cd, int unit, int tiles_m, float* smem, int cbeg = 0, int cend = -1, float* part = nullptr) {
    constexpr int BS = 16 * NT + 4;  // B staged [k][n]: stride % 8 == 4 keeps the four k-rows of an MFMA B fragment on disjoint banks
    float(*Qs)[H_QSZ] = reinterpret_cast<float(*)[H_QSZ]>(smem);
    float(*Bs)[H_KC * BS] = reinterpret_cast<float(*)[H_KC * BS]>(smem + 2 * H_QSZ);
    float* csum = smem + 2 * H_QSZ + 2 * H_KC * BS;
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
        ch = ch < nchunks ? ch : nchunks - 1;  // branch-free (see grad_adj16_body)
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
    constexpr float LOG2E = 1.4426950408889634f;
    const float lg = __log2f(D.gscale);
    if (!col_ok && 4 * bpiece < 16 * NT) {  // K-padding columns of the B tile: zero once, the staging of real columns never touches them
        for (int r = brow; r < 2 * H_KC; r += 8) *reinterpret_cast<float4*>(&Bs[0][r * BS + 4 * bpiece]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto write = [&](int buf, int ch, const float4(&vs)[2], const float(&lv)[2], const float4(&vb)[4]) {
        const int i0 = ch * H_KC;
        if (!L2 && i0 + H_KC <= D.Bc) {  // common case (uniform): a full chunk of real rows; columns j >= N feed output rows that are never stored
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const float c_ = fmaf(-lv[it], LOG2E, lg);
                float4 v;
                v.x = __builtin_amdgcn_exp2f(fmaf(vs[it].x, LOG2E, c_));
                v.y = __builtin_amdgcn_exp2f(fmaf(vs[it].y, LOG2E, c_));
                v.z = __builtin_amdgcn_exp2f(fmaf(vs[it].z, LOG2E, c_));
                v.w = __builtin_amdgcn_exp2f(fmaf(vs[it].w, LOG2E, c_));
                *reinterpret_cast<float4*>(&Qs[buf][(qrow + 16 * it) * H_QS_KM + 4 * qpiece]) = v;
            }
            if (col_ok) {
#pragma unroll
                for (int it = 0; it < 4; ++it) *reinterpret_cast<float4*>(&Bs[buf][(brow + 8 * it) * BS + 4 * bpiece]) = vb[it];
            }
            return;
        }
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
            if (4 * bpiece < 16 * NT) *reinterpret_cast<float4*>(&Bs[buf][r * BS + 4 * bpiece]) = v;
        }
    };
    auto compute = [&](int buf) {
        const float* qp = &Qs[buf][(4 * kq) * H_QS_KM + wave * 16 + l15];
        const float* bp = &Bs[buf][(4 * kq) * BS + l15];
        float bc[NT], bn[NT];
        float ac = qp[0], an = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) bc[t] = bp[16 * t];
#pragma unroll
        for (int i = 0; i < H_KC / 4; ++i) {
            if (i + 1 < H_KC / 4) {
                const int sn = (i + 1) >> 2, en = (i + 1) & 3;
                const float* brow_p = bp + (16 * sn + en) * BS;
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = brow_p[16 * t];
                an = qp[(16 * sn + en) * H_QS_KM];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (!(GRAD_ABLATE & 2)) acc[t] = mfma16(ac, bc[t], acc[t]); else acc[t][0] += ac * bc[t];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) bc[t] = bn[t];
            ac = an;
        }
    };

    const int cend_ = cend < 0 ? nchunks : cend;
    float4 vs0[2], vb0[4], vs1[2], vb1[4];
    float lv0[2], lv1[2];
    issue(cbeg, vs0, lv0, vb0);
    issue(cbeg + 1, vs1, lv1, vb1);
    write(0, cbeg, vs0, lv0, vb0);
    issue(cbeg + 2, vs0, lv0, vb0);
    __syncthreads();
    for (int ch = cbeg; ch < cend_; ch += 2) {
        compute(0);
        if (!(GRAD_ABLATE & 4)) write(1, ch + 1, vs1, lv1, vb1);
        if (!(GRAD_ABLATE & 8)) issue(ch + 3, vs1, lv1, vb1);
        __syncthreads();
        if (ch + 1 < cend_) compute(1);
        if (!(GRAD_ABLATE & 4)) write(0, ch + 2, vs0, lv0, vb0);
        if (!(GRAD_ABLATE & 8)) issue(ch + 4, vs0, lv0, vb0);
        __syncthreads();
    }
    if (part) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(t * 4 + r) * 256 + threadIdx.x] = acc[t][r];
        return;
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
// (One launch instead of two saves a launch gap, and the two kinds of units have different load/MFMA phase patterns, which
// desynchronises the workgroups sharing a CU.  The "rounds of resident slots" arithmetic is not the reason: see launch_grad16_hy.)
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

bool scores_a_applicable(const float* emb, int64_t emb_ld, int d) {
    const int nq = d / 4;
    return res_ok(emb, emb_ld, d) && (nq == 8 || nq == 16 || nq == 25 || nq == 32);
}

bool launch_scores_ap(const ScoreArgs& a_in, bool l2, hipStream_t st) {
    if (!scores_a_applicable(a_in.emb, a_in.emb_ld, a_in.D.d)) return false;
    ScoreArgs a = a_in;
    a.KS = a.D.d + 2;
    const int npairs = scores_ap_groups(a.D.N);
    const int mtiles = (int)cdiv(a.D.Bc, A_TM);
    const int total = a.D.C * a.D.ndir * mtiles * npairs;
    const int nwg = 768;  // 3 workgroups x 256 CUs, all resident
    const size_t lds = (size_t)A_SLOTS * A_TN * a.KS * sizeof(float);
#define SCORES_AP_LAUNCH(L2V, NQV) lp_scores_ap_kernel<L2V, NQV><<<dim3(nwg), dim3(256), lds, st>>>(a, npairs, mtiles, total, nwg)
#define SCORES_AP_DISPATCH(L2V)                         \
    do {                                                \
        switch (a.D.d / 4) {                            \
            case 8: SCORES_AP_LAUNCH(L2V, 8); break;    \
            case 16: SCORES_AP_LAUNCH(L2V, 16); break;  \
            case 25: SCORES_AP_LAUNCH(L2V, 25); break;  \
            default: SCORES_AP_LAUNCH(L2V, 32); break;  \
        }                                               \
    } while (0)
    if (l2)
        SCORES_AP_DISPATCH(true);
    else
        SCORES_AP_DISPATCH(false);
#undef SCORES_AP_DISPATCH
#undef SCORES_AP_LAUNCH
    return true;
}

bool launch_scores_res(const ScoreArgs& a_in, bool l2, hipStream_t st) {
    if (!scores_res_applicable(a_in.emb, a_in.emb_ld, a_in.D.d)) return false;
    ScoreArgs a = a_in;
    a.KS = a.D.d + 2;  // (d + 2) / 2 odd for d % 4 == 0: conflict-free ds_read_b64 across 32 rows
    const int mtiles = (int)cdiv(a.D.Bc, R_T);
    int nt_per_group, ngroups;  // 4 negative tiles per workgroup: fine-grained enough to balance 256 CUs
    scores_res_geometry(a.D.N, nt_per_group, ngroups);
    const int units = mtiles * ngroups;
    const size_t lds = (size_t)3 * R_T * a.KS * sizeof(float);
    // persistent workgroups.  d / 4 in {8, 16, 25, 32} never comes here by default (the adj-in-registers kernel takes those): runtime K loop only
    const int ncd = a.D.C * a.D.ndir;
    const int units_x = (int)cdiv(ncd, 8) * units;
    int wg_per_xcd = 64;  // 2 resident workgroups on each of the XCD's 32 CUs
    if (units_x < wg_per_xcd) wg_per_xcd = units_x;
    const size_t lds_ps = lds + 256 * sizeof(float);
    dim3 pgrid((unsigned)(8 * wg_per_xcd));
    if (l2) {
        if (lds_ps > 65536) (void)hipFuncSetAttribute((const void*)lp_scores_ps_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ps);
        lp_scores_ps_kernel<true, 0><<<pgrid, dim3(256), lds_ps, st>>>(a, ngroups, nt_per_group, units, wg_per_xcd);
    } else {
        if (lds_ps > 65536) (void)hipFuncSetAttribute((const void*)lp_scores_ps_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ps);
        lp_scores_ps_kernel<false, 0><<<pgrid, dim3(256), lds_ps, st>>>(a, ngroups, nt_per_group, units, wg_per_xcd);
    }
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
    const int nt_inst = nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : nt <= 7 ? 7 : 8;
    const size_t lds = grad16_lds_bytes(a.D.N, nt_inst);
    dim3 grid(xcd_grid2(units_adj + units_neg, a.D.C * a.D.ndir));
    if (l2)
        GRAD16_DISPATCH(true);
    else
        GRAD16_DISPATCH(false);
    return true;
}
#undef GRAD16_DISPATCH

}  // namespace marius
