// Flash-style CORRUPT_NODE training path on the BF16 matrix pipe: negative scores are never materialised.
//
// Reference path replaced (src/cpp/src): the negative half of nn/decoders/edge/decoder_methods.cpp:57-114 (node_corrupt_forward:
// pad_and_reshape + bmm, comparators.cpp:7-41), nn/loss.cpp:50-67 (SoftmaxCrossEntropy) and their autograd backward (nn/model.cpp:324)
// for the DotCompare decoders (DistMult, ComplEx) when the caller only trains (marius_lp_desc.flags & MARIUS_LP_TRAIN_ONLY: nobody
// reads `neg`).  The materialised-score kernels (lp_res.hip) stay the API path of forward_lp / evaluate_batch.
//
// Why: on the FP32 matrix pipe the step is bound by 6e10 flop at ~50 % of 157 TF, and the 400 MB score tensor is written once and
// read twice (1.2 GB of the step's 2.6 GB HBM traffic, profiles/r1h_pmc_traffic.json).  Here
//   * every fp32 operand x is split once per step into two bf16 values x = h + l (h = bf16(x), l = bf16(x - h): 16 significand bits),
//     and a product is taken as h.h' + (h.l' + l.h') on v_mfma_f32_32x32x16_bf16, accumulated in fp32.  Dropped: l.l' and the
//     representation residuals, each <= 2^-18 |x y| with round-to-nearest splits, so
//         |error of one contraction| <= 3 * 2^-18 * sum_k |a_k b_k|   (+ the usual fp32 accumulation error)
//     — tested against exactly this bound (tests/test_gpu_flash.py) and inside the 1e-4 score contract;
//   * forward keeps only the SoftmaxCE row statistics (running max / sum exp2, online softmax in registers);
//   * backward recomputes the score tile, forms V = dL/dS = g exp(S - lse) in registers and feeds it straight back into the
//     matrix pipe as the A operand of the gradient contraction (the accumulator layout of a 32x32 tile IS the A-operand layout of
//     the next MFMA once the streamed rows are stored in a fixed permutation): dAdj = V Neg and dNeg = V^T adj are two launches of
//     the same kernel with the roles of the two operands swapped.  (Round 2: 5 contractions x 3 products per step; since round 3 the
//     forward statistics and dAdj are ONE sweep — FLASH_FDADJ below — and a step needs 4: 2.4e11 16-bit flop at the bench shape.)
//   * round 3: fp16 halves of power-of-two-scaled rows (22 significand bits) instead of bf16 ones whenever the caller supplies magnitude
//     bounds (marius_lp_desc.absmax / absmax_rel); round 4: every training path does (a tracked table, the partition buffer's slab bound, or a
//     scan of the gathered rows), the bound on adj follows the relation operator (FlRange::adj_bound), and the arithmetic is measured against
//     the reference's fp32 evaluation in every bench run (oracle/arith_check.py).
//
// Operand records.  adj rows and negative rows are packed once per step (flash_pack_*_kernel) in OCCURRENCE order, one block of
// XR = ceil32(rows) records per (direction, chunk):
//     record = [ hi: KP bf16 | lo: KP bf16 | lsec: f32 | 12 B pad ]      KP = 16 ceil(d / 16),  pitch P = 4 KP + 16 bytes
// P = 16 mod 32, so (a) the 16-B fragment reads of 16 consecutive records hit 16 distinct bank groups and (b) four records 4 apart
// tile the 64 banks with their 64-B column windows, which is what ds_read_b64_tr_b16 (the hardware transpose read that yields the
// k-major B operand of the gradient contraction from row-major records) needs.  Inside every 16-record group, logical row y sits at
// physical slot rho(y) = 4 (y & 3) + (y >> 2) for that reason.  A streamed tile (32 records) is one contiguous 32 P byte run of
// HBM: it is DMA'd into a 4-slot LDS ring by global_load_lds_dwordx4 (no VGPRs, no VALU), three tiles in flight.
// `lsec` (adj records only) = lse log2(e) - log2(g), patched in by the merge kernel after the forward: V = exp2(S log2(e) - lsec).
//
// Work decomposition.  Items = (chunk-direction, 128-row tile of the stationary operand, 32-row block of the streamed operand),
// streamed index fastest.  nwg persistent workgroups (up to 2 per CU — fl_num_wg picks the count whose ranges align best with the tiles —
// 4 waves, a wave owns 32 stationary rows as MFMA fragments in
// registers) each take a contiguous, XCD-local range of items, balanced to +-1 block, with nwg <= number of tiles so that a tile is
// shared by at most two workgroups: the gradient of a split tile is accumulated with float atomics onto a zeroed output by exactly
// two contributors, which is order-independent (a + b == b + a), so results stay bit-reproducible; the forward statistics of a split
// tile go to two partial slots.
#include <cstdlib>

#include "lp_common.h"

namespace marius {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// ---- operand element type.  Round 2: bf16 (x = h + l carries 16 significand bits).  Round 3: fp16 when the caller supplies the magnitude
// bounds of its tables (marius_lp_desc.absmax): x 2^k = h + l with both halves fp16 carries 22 bits — the split error drops from 3 2^-18 to
// 3 2^-24 of sum|a_k b_k|, the class of the fp32 accumulation itself — at the same MFMA rate.  fp16's narrow exponent is what the bounds
// are for: every operand set is scaled by a power of two that puts its largest possible magnitude at [2^11, 2^12) (a factor 16 below
// overflow; conversions saturate anyway), so elements down to 2^-15 of the maximum keep all 22 bits and smaller ones lose low bits of an
// already negligible contribution.  Records, fragments and LDS traffic are bit patterns either way (the v8bf type below is storage only): the
// element type shows up in exactly two places, the conversion and the MFMA opcode.
// (fl_cvt16 / fl_back16 / fl_scale_of / fl_scales: lp_common.h — the prep kernel of lp_decoder.hip packs adj records with them too)
typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// FLASH_FDADJ (round 3, the default): forward statistics AND dAdj in one sweep, flash-attention style — V = exp2(S log2(e) - mref) against a
// per-row reference mref that starts at the row's positive score and is raised (accumulators rescaled) only when a block's maximum
// exceeds it by more than FL_TAU; the unnormalised sums leave the kernel as (mref, sum V) statistics plus the unnormalised dAdj partial of
// every (tile, contributor), and the consumer (lp_edge_bwd*) scales by g exp(mref - lse).  One score contraction less per step (4 instead
// of 5), no zero fill of dadj, no atomics.  FLASH_FWD remains for the score-storing parity runs; FLASH_DADJ only as the base of the stored-score mode FLASH_DADJS.
// d > 128 (round 3; cfg5's d = 400): the stationary operand of a tile no longer fits the register file, so the contraction index is cut into
// nch equal column chunks of <= 128 (400 = 4 x 100) with one operand-record set per chunk, and the scores ARE materialised (fp32 [Bp, n_ld]):
//   FLASH_FWDS   one launch per chunk: S += adj_c . neg_c^T (the first chunk stores, the last also leaves the SoftmaxCE row statistics);
//   FLASH_DADJS / FLASH_DNEGS   one launch per chunk: V = exp(S - lse) from the stored scores, dAdj[:, chunk] = V neg_c, dNeg[:, chunk] = V^T adj_c.
// Same tiles, ring and MFMA phases as the d <= 128 kernels; what replaces the FP32-MFMA kernels' 1.6 ms of matrix time is 3 x 16-bit products.
// The stored scores live in tile order and travel through a second LDS ring (DMA, two items ahead, same counted waits as the blocks): with the
// tile fetched by ordinary loads one item ahead these launches were latency-bound (4.9 us per item, cfg5 step 2.18 ms = no faster than the
// FP32 path); with the ring 1.78 ms.  Three slots + two score tiles = 80 KB at KS >= 7, so two workgroups still share a CU (four + three
// slots at one workgroup per CU: 1.91 ms).
enum { FLASH_FWD = 0, FLASH_DADJ = 1, FLASH_DNEG = 2, FLASH_FDADJ = 3, FLASH_FWDS = 4, FLASH_DADJS = 5, FLASH_DNEGS = 6 };
__host__ __device__ constexpr int fl_base(int mode) { return mode == FLASH_FWDS ? FLASH_FWD : mode == FLASH_DADJS ? FLASH_DADJ : mode == FLASH_DNEGS ? FLASH_DNEG : mode; }
constexpr float FL_TAU = 8.f;  // log2 units: V <= 2^8 between raises (bf16 / fp32 share the exponent range: no overflow, no lost precision)

// FL_WAVES_N=8 (256-row stationary tile, one workgroup per CU, six slots) halves the streamed bytes per flop; measured slower at the
// bench shape (scores 82 vs 72 us, dAdj 141 vs 134, dNeg 148 vs 138): the streamed operand is not what the kernels wait for.
#ifndef FL_WAVES_N
#define FL_WAVES_N 4
#endif
constexpr int FL_WAVES = FL_WAVES_N, FL_NT = 64 * FL_WAVES, FL_XT = 32 * FL_WAVES, FL_YB = 32;
// LDS ring slots / workgroups per CU: the forward needs 122 VGPRs, so three workgroups fit a CU if each keeps to three slots
#ifndef FL_FWD_SLOTS
#define FL_FWD_SLOTS 3
#endif
// FL_S_DEEP=1: four block slots + three score-tile slots, one workgroup per CU; 0: three + two (both two items ahead), two workgroups per CU
#ifndef FL_S_DEEP
#define FL_S_DEEP 0
#endif
// 1: the DMA pieces of the next streamed block are issued between the k-steps of the score contraction instead of in a run at the top of the item
#ifndef FL_DMA_INTERLEAVE
#define FL_DMA_INTERLEAVE 1
#endif
#ifndef FL_DMA_TRIM
#define FL_DMA_TRIM 1
#endif
#ifndef FL_DMA_KS0   // first k-step that is followed by a DMA piece, and the k-step distance between pieces
#define FL_DMA_KS0 0
#endif
#ifndef FL_DMA_STRIDE
#define FL_DMA_STRIDE 1
#endif
__host__ __device__ constexpr int fl_slots(int mode) { return FL_WAVES == 8 ? 6 : (mode == 0 ? FL_FWD_SLOTS : (mode >= 4 && !FL_S_DEEP) ? 3 : 4); }
// the stored-score modes (d > 128) keep a second ring beside the streamed blocks — the 16 KB score tile of every item in flight — and
// run one workgroup per CU for it
constexpr int FL_STILE = FL_XT * FL_YB * 4;
__host__ __device__ constexpr int fl_sslots(int mode) { return mode >= 4 ? (FL_S_DEEP ? 3 : 2) : 0; }
__host__ __device__ constexpr int fl_sring_bytes(int mode) { return fl_sslots(mode) * FL_STILE; }
__host__ __device__ constexpr int fl_wg_per_cu(int mode) { return (FL_WAVES == 8 || (mode >= 4 && FL_S_DEEP)) ? 1 : (mode == 0 ? (FL_FWD_SLOTS == 3 ? 3 : 2) : 2); }
constexpr float FL_LOG2E = 1.4426950408889634f, FL_LN2 = 0.6931471805599453f;

// Round 5: the FOLDED COLUMN TAIL.  d = 100 fills six k-steps of 16 and leaves four columns: as a seventh k-step of its own they cost three MFMAs
// in the score contraction and six in the gradient contraction for 4 of 16 (32) useful columns.  When d = 4 (mod 16) and the number of k-steps
// is odd (d = 36, 68, 100) the record keeps those four columns' hi AND lo halves side by side in ONE 16-byte piece at the end of the hi region,
//     record = [ hi: 16 (KS - 1) | T = h0 h1 h2 h3 l0 l1 l2 l3 | 8 zeros | lo: 16 (KS - 1) | lsec: f32 | 12 B pad ],   P = 64 KS - 16,
// so that (a) the last score k-step is TWO MFMAs on the streamed fragment [T | 0]: against [h'h' | 0] (h h' + l h') and against [l'0 | 0] (h l');
// (b) the last 32-column tile of the gradient contraction — which starts at T because KS - 1 is even — is TWO MFMAs per k-step, V_h and V_l against
// the hi window, whose columns 0-3 / 4-7 are the h / l parts of the four tail columns (added when the accumulators leave the registers; the
// V_l l product it also forms is the one the three-product scheme drops).  42 instead of 45 MFMAs per item; lo offset 2 KP and P = 16 (mod 32)
// are unchanged, so every bank-conflict property of the old layout holds.  MARIUS_FLASH_TAIL4=0: the seven-k-step layout (A/B runs).
__host__ __device__ constexpr int fl_pitch(int KS, bool TAIL = false) { return TAIL ? 64 * KS - 16 : 64 * KS + 16; }   // bytes per record
__host__ __device__ constexpr int fl_lsec_off(int KS, bool TAIL = false) { return fl_pitch(KS, TAIL) - 16; }        // byte offset of lsec in a record
__host__ __device__ constexpr int fl_slot_bytes(int KS, bool TAIL = false) { return (32 * fl_pitch(KS, TAIL) + FL_WAVES * 1024 - 1) / (FL_WAVES * 1024) * (FL_WAVES * 1024); }   // DMA granule: 4 waves x 1 KB
__host__ __device__ constexpr int fl_rho(int y) { return 4 * (y & 3) + ((y >> 2) & 3) + (y & ~15); }          // logical -> physical slot in a 16-group
__host__ __device__ constexpr int fl_pi(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }         // swap bits 2 and 3

struct FlashArgs {
    const char* xrec;   // stationary operand records  [ncd][XR]
    const char* yrec;   // streamed operand records    [ncd][YR]
    int XR, YR;         // records per (chunk, direction) block, multiples of 32
    int Xrows, Yrows;   // valid rows per block
    int ncd, XT, YB, nwg;
    int rotate;         // 1: rotated sweeps (see flash_kernel)
    int64_t total;      // ncd * XT * YB
    int C, Bc, N, d;
    int64_t Bp;
    // FWD: SoftmaxCE partials [2][ndir * Bp] (m, l) with contribution l e^m; S (debug / parity only) [ndir][Bp][n_ld]
    float2* part;
    float* S;
    int64_t n_ld;
    // BWD: output rows [., out_ld]; DADJ: row = dir Bp + c Bc + x; DNEG: row = negocc_off[dir] + c N + x
    float* out;
    int64_t out_ld;
    int64_t negocc_off[2];
    // FDADJ: positive scores [ndir Bp] (initial reference of the online softmax); out = partial of a tile's first contributor, out2 = of its second
    const float* pos;
    float* out2;
    int chunk_first, chunk_last;  // FWDS: this launch handles the first / last column chunk of the contraction index
    int s_tiled;                  // stored scores in tile order [chunk-direction][128-row adj tile][32-column block][128][32] (internal) instead of [Bp, n_ld]
    FlRange rg;  // operand scales (fp16 records) are derived from it by every kernel of the step
    // score filter (apply_score_filter, negative.cpp:306-311: listed (row, column) scores count as -1e9): entries bucketed by item,
    // foff[item] .. foff[item + 1] into fent; an entry = (stationary row inside the 128-row tile) << 5 | (streamed row inside the block).
    // nullptr: no filter.
    const uint32_t* foff;
    const uint16_t* fent;
};

// zero up to three float ranges (lengths are multiples of 4, bases 16-B aligned) in one launch
__global__ __launch_bounds__(256) void flash_zero_kernel(float* a, int64_t na, float* b, int64_t nb, float* c, int64_t nc) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < na; i += stride) *reinterpret_cast<float4*>(a + i) = z;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < nb; i += stride) *reinterpret_cast<float4*>(b + i) = z;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < nc; i += stride) *reinterpret_cast<float4*>(c + i) = z;
}

// ---------------------------------------------------------------------------------------------------------------- pack kernels
// fp32 row -> record.  One thread per 4 consecutive elements (8 B of hi, 8 B of lo); threads past d write the zero K padding.
// tail_piece >= 0 (folded column tail): piece tail_piece holds the last four columns — hi then lo side by side; the two pieces behind it are the
// eight zeros of T's k-step (written where the old layout had hi padding), the piece after those writes nothing
template <bool F16>
__device__ __forceinline__ void fl_write_piece(char* rec, int KP, int piece, float4 v, float scale, int tail_piece = -1) {
#pragma clang fp contract(off)
    if (tail_piece >= 0 && piece > tail_piece) {
        if (piece <= tail_piece + 2) *reinterpret_cast<uint2*>(rec + (piece + 1) * 8) = make_uint2(0u, 0u);
        return;
    }
    const float x[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};  // scale is a power of two: exact
    unsigned short H[4], L[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        H[j] = fl_cvt16<F16>(x[j]);
        L[j] = fl_cvt16<F16>(x[j] - fl_back16<F16>(H[j]));
    }
    *reinterpret_cast<uint2*>(rec + piece * 8) = make_uint2((unsigned)H[0] | ((unsigned)H[1] << 16), (unsigned)H[2] | ((unsigned)H[3] << 16));
    char* lo = piece == tail_piece ? rec + piece * 8 + 8 : rec + 2 * KP + piece * 8;
    *reinterpret_cast<uint2*>(lo) = make_uint2((unsigned)L[0] | ((unsigned)L[1] << 16), (unsigned)L[2] | ((unsigned)L[3] << 16));
}

// adj [ndir][Bp][d_ld] fp32 (written by lp_prep*) -> adj records; chunk c of direction dir holds rows c Bc .. (c + 1) Bc of that direction
// d = columns of this record set, taken from column col0 on (one set per column chunk when the rows are wider than 128)
__global__ __launch_bounds__(256) void flash_pack_adj_kernel(const float* __restrict__ adj, int64_t d_ld, int64_t Bp, int Bc, int C, int ndir, int d,
                                                             int KP, int XR, char* __restrict__ rec, FlRange rg, int col0, int tail) {
    const int ppr = KP / 4 + 1;  // pieces per record (+1: the tail)
    const int tp = tail ? (d - 4) / 4 : -1;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nrec = (int64_t)ndir * C * XR;
    if (idx >= nrec * ppr) return;
    const int64_t r = idx / ppr;
    const int piece = (int)(idx - r * ppr);
    const int64_t cd = r / XR;
    const int x = (int)(r - cd * XR);
    const int P = tail ? 4 * KP - 16 : 4 * KP + 16;
    char* o = rec + (cd * XR + fl_rho(x)) * (int64_t)P;
    if (piece == KP / 4) {  // tail: lsec of a padding record must be finite (V of a zero row is multiplied by zeros)
        *reinterpret_cast<float4*>(o + P - 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < Bc && 4 * piece < d) {
        const int64_t dir = cd / C, c = cd - dir * C;
        const float* src = adj + (dir * Bp + c * Bc + x) * d_ld + col0 + 4 * piece;
        if (4 * piece + 3 < d) v = *reinterpret_cast<const float4*>(src);
        else {
            v.x = src[0];
            if (4 * piece + 1 < d) v.y = src[1];
            if (4 * piece + 2 < d) v.z = src[2];
        }
    }
    if (rg.absmax) fl_write_piece<true>(o, KP, piece, v, fl_scales(rg).s_adj, tp);
    else fl_write_piece<false>(o, KP, piece, v, 1.f, tp);
}

// negatives: record (dir, c, j) = emb[negmap[dir][c N + j]]; rows N .. NR are zero
__global__ __launch_bounds__(256) void flash_pack_neg_kernel(const float* __restrict__ emb, int64_t emb_ld, const int64_t* __restrict__ neg0,
                                                             const int64_t* __restrict__ neg1, int N, int C, int ndir, int d, int KP, int NR,
                                                             int vec, char* __restrict__ rec, float* __restrict__ gocc, int64_t d_ld, int64_t off0,
                                                             int64_t off1, FlRange rg, int col0, int tail) {
    const int ppr = KP / 4 + 1;
    const int tp = tail ? (d - 4) / 4 : -1;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nrec = (int64_t)ndir * C * NR;
    if (idx >= nrec * ppr) return;
    const int64_t r = idx / ppr;
    const int piece = (int)(idx - r * ppr);
    const int64_t cd = r / NR;
    const int j = (int)(r - cd * NR);
    const int P = tail ? 4 * KP - 16 : 4 * KP + 16;
    char* o = rec + (cd * NR + fl_rho(j)) * (int64_t)P;
    if (piece == KP / 4) {
        *reinterpret_cast<float4*>(o + P - 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < N) {
        const int64_t dir = cd / C, c = cd - dir * C;
        if (4 * piece < d) {
            const int64_t id = (dir ? neg1 : neg0)[c * N + j];
            v = load_row4(emb + id * emb_ld + col0, 4 * piece, d, vec);
        }
        // the negative's gradient row is accumulated by at most two workgroups of the backward: it starts from zero
        if (col0 + 4 * piece < d_ld) *reinterpret_cast<float4*>(gocc + ((dir ? off1 : off0) + c * N + j) * d_ld + col0 + 4 * piece) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (rg.absmax) fl_write_piece<true>(o, KP, piece, v, fl_scales(rg).s_neg, tp);
    else fl_write_piece<false>(o, KP, piece, v, 1.f, tp);
}

// ---------------------------------------------------------------------------------------------------------------- score filter index
// One workgroup per orientation (0: adj rows stationary — the forward + dAdj sweep; 1: negatives stationary — dNeg) buckets the filter
// entries (row in [0, B), column in [0, N)) of both directions by the item that holds their score: counts, exclusive scan and scatter all in
// LDS.  The order of the entries inside an item is whatever the atomics give: masking is idempotent, results do not depend on it.
__global__ __launch_bounds__(1024) void flash_filter_index_kernel(const int64_t* __restrict__ f0, int64_t n0, const int64_t* __restrict__ f1, int64_t n1,
                                                                  int Bc, int C, int N, int ndir, int XTa, int YBa, int XTn, int YBn, int nkeys_max,
                                                                  uint32_t* __restrict__ foff, uint16_t* __restrict__ fent, int64_t ent_cap) {
    extern __shared__ uint32_t cnt[];  // [nkeys + 1]
    __shared__ uint32_t wsum[16];
    const int o = blockIdx.x, tid = threadIdx.x;
    const int XT = o == 0 ? XTa : XTn, YB = o == 0 ? YBa : YBn;
    const int nkeys = ndir * C * XT * YB;
    uint32_t* off_out = foff + (size_t)o * (nkeys_max + 1);
    uint16_t* ent_out = fent + (size_t)o * ent_cap;
    for (int k = tid; k <= nkeys; k += 1024) cnt[k] = 0u;
    __syncthreads();
    auto key_of = [&](int dir, int64_t row, int64_t col, int& ent) {
        const int c = (int)(row / Bc), x = (int)(row - (int64_t)c * Bc);
        const int cd = dir * C + c;
        if (o == 0) {
            ent = ((x & 127) << 5) | (int)(col & 31);
            return (cd * XT + (x >> 7)) * YB + (int)(col >> 5);
        }
        ent = (((int)col & 127) << 5) | (x & 31);
        return (cd * XT + (int)(col >> 7)) * YB + (x >> 5);
    };
    const int64_t n = n0 + n1;
    // rows of (-1, -1) are the slots of an uncompacted DEG filter (marius_deg_filter): ignored, like anything outside the score matrix
    auto valid = [&](const int64_t* e) { return e[0] >= 0 && e[0] < (int64_t)Bc * C && e[1] >= 0 && e[1] < N; };
    for (int64_t i = tid; i < n; i += 1024) {
        const int dir = i < n0 ? 0 : 1;
        const int64_t* e = dir == 0 ? f0 + 2 * i : f1 + 2 * (i - n0);
        int ent;
        if (valid(e)) atomicAdd(&cnt[key_of(dir, e[0], e[1], ent)], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0 .. nkeys]: thread t owns a contiguous slice
    const int per = (nkeys + 1 + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nkeys + 1);
    uint32_t s_ = 0;
    for (int k = lo; k < hi; ++k) s_ += cnt[k];
    uint32_t x_ = s_;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d_ = 1; d_ < 64; d_ <<= 1) {
        const uint32_t y_ = __shfl_up(x_, d_, 64);
        if (lane >= d_) x_ += y_;
    }
    if (lane == 63) wsum[wave] = x_;
    __syncthreads();
    uint32_t base = x_ - s_;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    for (int k = lo; k < hi; ++k) {
        const uint32_t c_ = cnt[k];
        cnt[k] = base;
        off_out[k] = base;
        base += c_;
    }
    __syncthreads();
    for (int64_t i = tid; i < n; i += 1024) {
        const int dir = i < n0 ? 0 : 1;
        const int64_t* e = dir == 0 ? f0 + 2 * i : f1 + 2 * (i - n0);
        int ent;
        if (!valid(e)) continue;
        const int k = key_of(dir, e[0], e[1], ent);
        const uint32_t slot = atomicAdd(&cnt[k], 1u);
        if ((int64_t)slot < ent_cap) ent_out[slot] = (uint16_t)ent;
    }
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
template <bool F16>
__device__ __forceinline__ v16f fl_mfma(const v8bf& a, const v8bf& b, v16f c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- the two matrix phases, shared by both kernels.  Operand fragments are read from LDS two steps ahead of the MFMAs that consume
// them (a three-entry register ring): without the explicit distance hipcc emits read -> wait -> MFMA chains and every step pays the LDS
// latency inside the matrix phase.
// TAIL: the last k-step is the folded column tail — the streamed fragment [T | 0] against xh[KS - 1] = [h' h' | 0] and xl[KS - 1] = [l' 0 | 0]
// (built by load_x): two MFMAs, no lo fragment
// `between(ks)` runs after the MFMAs of k-step ks (FL_DMA_INTERLEAVE: the next block's DMA pieces are issued there, one per k-step, into the
// shadow of the matrix instructions in flight instead of in a run of their own at the top of the item)
template <int KS, bool F16, bool TAIL, class Between>
__device__ __forceinline__ v16f fl_score_tile(const unsigned char* T, int a_off, const v8bf (&xh)[KS], const v8bf (&xl)[KS], const Between& between) {
    constexpr int KP = 16 * KS;
    v16f accM, accC;
#pragma unroll
    for (int r_ = 0; r_ < 16; ++r_) accM[r_] = accC[r_] = 0.f;
    v8bf yh[3], yl[3];
#pragma unroll
    for (int ks = 0; ks < 2 && ks < KS; ++ks) {
        yh[ks] = *reinterpret_cast<const v8bf*>(T + a_off + 32 * ks);
        if (!(TAIL && ks == KS - 1)) yl[ks] = *reinterpret_cast<const v8bf*>(T + a_off + 2 * KP + 32 * ks);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 2 < KS) {
            yh[(ks + 2) % 3] = *reinterpret_cast<const v8bf*>(T + a_off + 32 * (ks + 2));
            if (!(TAIL && ks + 2 == KS - 1)) yl[(ks + 2) % 3] = *reinterpret_cast<const v8bf*>(T + a_off + 2 * KP + 32 * (ks + 2));
        }
#ifdef FL_ONE_ACC
        accM = fl_mfma<F16>(yl[ks % 3], xh[ks], accM);
        accM = fl_mfma<F16>(yh[ks % 3], xl[ks], accM);
        accM = fl_mfma<F16>(yh[ks % 3], xh[ks], accM);
#else
        accM = fl_mfma<F16>(yh[ks % 3], xh[ks], accM);
        accC = fl_mfma<F16>(yh[ks % 3], xl[ks], accC);
        if (!(TAIL && ks == KS - 1)) accC = fl_mfma<F16>(yl[ks % 3], xh[ks], accC);
#endif
        between(ks);
    }
    v16f acc;
#pragma unroll
    for (int r_ = 0; r_ < 16; ++r_) acc[r_] = accM[r_] + accC[r_];
    return acc;
}

struct FlTrFrag {
    union { v4s p[2]; v8bf f; } h, l;
};
// TAIL: the last column tile starts at T (KS - 1 is even): its hi window is [h0..3 l0..3 | 8 zeros | 16 columns of the lo region], so V_h and V_l
// against it give the h and l parts of the four tail columns in accumulator columns 0-3 and 4-7 (folded by fl_fold_tail when the accumulators
// leave the registers); there is no lo window
template <int KS, bool TAIL>
__device__ __forceinline__ FlTrFrag fl_tr_read(const unsigned char* T, int tr_off, int g) {  // g = 2 ct + s
    constexpr int KP = 16 * KS, P = fl_pitch(KS, TAIL), NCT = (KP + 31) / 32;
    const unsigned char* q = T + tr_off + (16 * (g & 1)) * P + 64 * (g >> 1);
    FlTrFrag r;
    r.h.p[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(q));
    r.h.p[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(q + P));
    if (!(TAIL && (g >> 1) == NCT - 1)) {
        r.l.p[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(q + 2 * KP));
        r.l.p[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(q + 2 * KP + P));
    }
    return r;
}
template <int KS, int NCT, bool F16, bool TAIL>
__device__ __forceinline__ void fl_grad_tile(const unsigned char* T, int tr_off, const v8bf (&wh)[2], const v8bf (&wl)[2], v16f (&out)[NCT]) {
    constexpr int G = 2 * NCT;
    FlTrFrag b[3];
#pragma unroll
    for (int g = 0; g < 2 && g < G; ++g) b[g] = fl_tr_read<KS, TAIL>(T, tr_off, g);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 2 < G) b[(g + 2) % 3] = fl_tr_read<KS, TAIL>(T, tr_off, g + 2);
        const int ct = g >> 1, s_ = g & 1;
        out[ct] = fl_mfma<F16>(wh[s_], b[g % 3].h.f, out[ct]);
        if (!(TAIL && ct == NCT - 1)) out[ct] = fl_mfma<F16>(wh[s_], b[g % 3].l.f, out[ct]);
        out[ct] = fl_mfma<F16>(wl[s_], b[g % 3].h.f, out[ct]);
    }
}
// accumulator columns 4-7 of the tail tile hold the lo parts of columns 0-3: lane c takes lane c + 4's value (lanes of one half-wave share rows)
__device__ __forceinline__ float fl_fold_tail(float v) { return v + __shfl_down(v, 4, 64); }

// (chunks wider than 128 columns — KS > 8, stored-score modes — run one workgroup per CU: their block slots fill the LDS, and one wave per SIMD
// may then use the whole 512-entry register file)
__host__ __device__ constexpr int fl_wg_per_cu_ks(int mode, int ks) { return (mode >= FLASH_FWDS && ks > 8) ? 1 : fl_wg_per_cu(mode); }
template <int KS, int MODE, bool STORE_S, bool F16, bool TAIL>
__global__ __launch_bounds__(FL_NT, fl_wg_per_cu_ks(MODE, KS)) void flash_kernel(FlashArgs a) {
    static_assert(!TAIL || ((KS & 1) && KS >= 3 && MODE < FLASH_FWDS), "folded column tail: odd k-step counts of the d <= 128 kernels only");
    constexpr int KP = 16 * KS, P = fl_pitch(KS, TAIL), LSEC = fl_lsec_off(KS, TAIL), SLOT = fl_slot_bytes(KS, TAIL), NSLOT = fl_slots(MODE);
    // ---- scales (all powers of two).  Accumulated scores carry s_x s_y; V = exp2(...) is formed VSH binades up so that its fp16 halves keep
    // their bits (V <= 2^FL_TAU in the fused sweep, <= 1 where lse is known); the outputs are scaled back when they leave the registers.
    const FlScales sc_ = F16 ? fl_scales(a.rg) : FlScales{1.f, 1.f};
    constexpr int BASE = fl_base(MODE);                                      // which of the d <= 128 kernels this launch is shaped like
    constexpr bool SACC = (MODE == FLASH_FWDS), SLOAD = (MODE == FLASH_DADJS || MODE == FLASH_DNEGS);
    const float s_y = (BASE == FLASH_DNEG) ? sc_.s_adj : sc_.s_neg;
    const float inv_xy = 1.f / (sc_.s_adj * sc_.s_neg);
    const float c_s = (SACC || SLOAD) ? FL_LOG2E : FL_LOG2E * inv_xy;        // t[] -> score in log2 units (stored scores are in true units)
    constexpr float VSH = !F16 ? 0.f : (MODE == FLASH_FDADJ ? 6.f : (BASE == FLASH_FWD ? 0.f : 14.f));
    const float out_unscale = __builtin_amdgcn_exp2f(-VSH) / s_y;
    constexpr int NCT = (KP + 31) / 32;       // 32-column tiles of the gradient output
    constexpr int DMA_PER_WAVE = SLOT / (FL_WAVES * 1024);  // 1 KB wave-instructions per wave and tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5, t16 = lane & 15, c16 = (lane >> 4) & 1;

    // ---- item range of this workgroup (contiguous per XCD: blockIdx % 8 is the XCD the block runs on).  total < 2^31 (checked on the host)
    int wlin = blockIdx.x;
    if ((a.nwg & 7) == 0) wlin = (int)(blockIdx.x & 7) * (a.nwg >> 3) + (int)(blockIdx.x >> 3);
    const int it0 = (int)(a.total * wlin / a.nwg), it1 = (int)(a.total * (wlin + 1) / a.nwg);
    if (it0 >= it1) return;

    // ---- per-lane LDS offsets
    // S-MFMA A operand: lane supplies C-row i = l31 of the streamed tile, i.e. logical row pi(i), physical slot rho(pi(i)); k = 16 ks + 8 h + e
    const int a_off = fl_rho(fl_pi(l31)) * P + 16 * h;
    // gradient-MFMA B operand through the transpose read: 16-lane group (c16, h) reads keys 4 apart; see the file header
    const int tr_off = (4 * (t16 >> 2) + 2 * h) * P + 32 * c16 + 8 * (t16 & 3);

    // ---- DMA: wave w moves the 1 KB pieces w, w + 4, ... of a tile's 32 P bytes (the tail of the last piece over-reads into the next tile).
    // Issued from inline asm so that hipcc does not count it: with the builtin it drains vmcnt(0) before the first ds_read of every
    // iteration (it cannot tell which slot a DMA writes), which serialises the ring.  M0 = LDS byte address of the piece.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_piece = [&](const char* src, unsigned dst, int i) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src + i * (FL_WAVES * 1024)), "s"(dst + (unsigned)(i * FL_WAVES * 1024))
                     : "memory");
    };
    auto dma_src = [&](int tile, int yb) {
        const int cdp = tile / a.XT;
        return a.yrec + (((int64_t)cdp * a.YR + (int64_t)yb * FL_YB) * P) + lane * 16 + wave_u * 1024;
    };
    auto dma_dst = [&](int slot) { return lds_base + (unsigned)(slot * SLOT) + (unsigned)(wave_u * 1024); };
    // FL_DMA_TRIM: a slot is a whole number of 4 KB rounds (one 1 KB piece per wave and round) but a block is 32 P bytes — 13.5 pieces of 16 at
    // d = 100: the waves whose last piece lies wholly beyond the block do not issue it (and wait for one load less per block in flight)
    constexpr int NPIECES = (32 * P + 1023) / 1024;
    const bool short_wave = FL_DMA_TRIM && !(SACC || SLOAD) && (wave_u + FL_WAVES * (DMA_PER_WAVE - 1)) >= NPIECES;
    auto dma = [&](int tile, int yb, int slot) {
        const char* src = dma_src(tile, yb);
        const unsigned dst = dma_dst(slot);
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i)
            if (!(short_wave && i == DMA_PER_WAVE - 1)) dma_piece(src, dst, i);
    };

    // stored-score modes, tile order: the score tile of an item travels with its streamed block, straight into LDS (a register prefetch cannot
    // be three items deep without the compiler moving half-loaded registers about; one item of distance leaves HBM latency exposed: 4.9 us
    // per item at cfg5's shape).  A wave's part of a tile is one contiguous 4 KB run in both layouts: forward / dAdj — rows 32 wave .. + 31 of
    // the item's own [128][32] tile; dNeg — rows (32 yph & 127) .. + 31 of tile (adj tile yph / 4, negative block 4 xt + wave).  The wave that
    // fetches a part is the only one that reads it.
    constexpr int S_DMA = 4;  // 1 KB wave-instructions per wave and score tile
    constexpr int SSLOTS = fl_sslots(MODE) > 0 ? fl_sslots(MODE) : 1;  // score tiles in flight = the distance of the block ring (NSLOT - 1)
    auto dma_s = [&](int tile_, int yph_, int slot_) {
        const int cd_ = tile_ / a.XT, xt_ = tile_ - cd_ * a.XT;
        const float* src;
        if (BASE != FLASH_DNEG) {
            src = a.S + ((((int64_t)cd_ * a.XT + xt_) * a.YB + yph_) * FL_XT + wave * 32) * FL_YB;
        } else {
            const int nta = (a.Yrows + FL_XT - 1) / FL_XT, nnb = (a.Xrows + FL_YB - 1) / FL_YB;
            const int nb = min(xt_ * 4 + wave, nnb - 1);
            const int y0 = yph_ * FL_YB;
            src = a.S + ((((int64_t)cd_ * nta + (y0 >> 7)) * nnb + nb) * FL_XT + (y0 & 127)) * FL_YB;
        }
        const char* sp = reinterpret_cast<const char*>(src) + lane * 16;
        const unsigned dst = lds_base + (unsigned)(NSLOT * SLOT) + (unsigned)(slot_ * FL_STILE) + (unsigned)(wave_u * 4096);
#pragma unroll
        for (int i = 0; i < S_DMA; ++i) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(sp + i * 1024), "s"(dst + (unsigned)(i * 1024))
                         : "memory");
        }
    };
    auto s_take = [&](int xt_, int yph_, int slot_, float (&v)[16]) {  // this lane's sixteen scores of the item out of the LDS tile; outside the matrix: -inf
        const unsigned char* base = smem + NSLOT * SLOT + slot_ * FL_STILE + wave * 4096;
        const int xq = xt_ * FL_XT + wave * 32 + l31;
        if (BASE != FLASH_DNEG) {
            const float* row = reinterpret_cast<const float*>(base) + l31 * FL_YB + 8 * h;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 f = *reinterpret_cast<const float4*>(row + 16 * s_ + 4 * q);
                    v[8 * s_ + 4 * q] = f.x;
                    v[8 * s_ + 4 * q + 1] = f.y;
                    v[8 * s_ + 4 * q + 2] = f.z;
                    v[8 * s_ + 4 * q + 3] = f.w;
                }
            }
        } else {
            const float* col = reinterpret_cast<const float*>(base) + l31;
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) v[r_] = col[(16 * (r_ >> 3) + 8 * h + (r_ & 7)) * FL_YB];
        }
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) {
            const int yq = yph_ * FL_YB + 16 * (r_ >> 3) + 8 * h + (r_ & 7);
            if (!(xq < a.Xrows && yq < a.Yrows)) v[r_] = -INFINITY;
        }
    };

    // stationary fragments (B operand of the S-MFMA): lane holds X[x = l31][k = 16 ks + 8 h .. + 7], hi and lo
    v8bf xh[KS], xl[KS];
    float2 lsec_x = make_float2(0.f, 0.f);  // DADJ: (hi, lo) of lsec - VSH of the lane's own adj row (flash_merge_kernel)
    v16f out[BASE == FLASH_FWD ? 1 : NCT];
    float m2 = -INFINITY, lsum = 0.f;  // FWD: running max of S log2(e) and sum of exp2 over this lane's columns
    float mref = 0.f;                  // FDADJ: reference of the lane's own row x = l31 (log2 units; the same in both lane halves); lsum: this lane's share of sum V

    int cur_tile = -1;
    int cd = 0, xt = 0, c_ = 0, dir = 0;
    int y_first = 0;  // first streamed block of the current tile segment

    auto load_x = [&](int tile) {
        cd = tile / a.XT;
        xt = tile - cd * a.XT;
        dir = cd / a.C;
        c_ = cd - dir * a.C;
        const int x = xt * FL_XT + wave * 32 + l31;  // logical row inside the block; rows >= XR do not exist: clamp (never stored)
        const int xc = x < a.XR ? x : a.XR - 1;
        const char* r = a.xrec + ((int64_t)cd * a.XR + fl_rho(xc)) * P;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xh[ks] = *reinterpret_cast<const v8bf*>(r + 32 * ks + 16 * h);
            if (!(TAIL && ks == KS - 1)) xl[ks] = *reinterpret_cast<const v8bf*>(r + 2 * KP + 32 * ks + 16 * h);
        }
        if constexpr (TAIL) {  // the piece just read is T = [h0..3 | l0..3] (lanes h = 1: the zero piece): B operands [h h | .] and [l 0 | .]
            union { v8bf v; unsigned u[4]; } t, xa, xb;
            t.v = xh[KS - 1];
            xa.u[0] = t.u[0]; xa.u[1] = t.u[1]; xa.u[2] = t.u[0]; xa.u[3] = t.u[1];
            xb.u[0] = t.u[2]; xb.u[1] = t.u[3]; xb.u[2] = 0u; xb.u[3] = 0u;
            xh[KS - 1] = xa.v;
            xl[KS - 1] = xb.v;
        }
        if (BASE == FLASH_DADJ) lsec_x = *reinterpret_cast<const float2*>(r + LSEC);
        if (MODE == FLASH_FDADJ) {  // the row's positive score is a term of the same softmax: start from it (rows past Xrows are never stored)
            mref = (x < a.Xrows) ? a.pos[(int64_t)dir * a.Bp + (int64_t)c_ * a.Bc + x] * FL_LOG2E : 0.f;
            lsum = 0.f;
        }
        // land the fragments HERE: hipcc's own wait for them would otherwise sit at their first use inside the item loop, and since
        // it cannot see the DMAs of the asm statements it would read as vmcnt(0..3) there — draining the ring every iteration
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (BASE != FLASH_FWD) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) out[ct][r_] = 0.f;
        } else {
            m2 = -INFINITY;
            lsum = 0.f;
        }
    };

    auto flush = [&](int y_last_excl) {
        const bool first = (y_first == 0), last = (y_last_excl == a.YB);
        if (SACC && !a.chunk_last) return;  // the statistics exist only once the last column chunk has been added
        if (BASE == FLASH_FWD) {
            // combine the two lane halves (disjoint columns of the same row), then one (m, l) pair per row
            const float m_o = __shfl_xor(m2, 32, 64), l_o = __shfl_xor(lsum, 32, 64);
            const float mm = fmaxf(m2, m_o);
            const float ll = lsum * __builtin_amdgcn_exp2f(m2 - mm) + l_o * __builtin_amdgcn_exp2f(m_o - mm);
            const int x = xt * FL_XT + wave * 32 + l31;
            if (h == 0 && x < a.Xrows) {
                const int64_t row = (int64_t)dir * a.Bp + (int64_t)c_ * a.Bc + x;
                const int64_t prows = (a.ncd / a.C) * a.Bp;
                const float2 v = make_float2(mm * FL_LN2, ll);
                if (first) {
                    a.part[row] = v;
                    if (last) a.part[prows + row] = make_float2(-INFINITY, 0.f);
                } else {
                    a.part[prows + row] = v;
                }
            }
        } else if (MODE == FLASH_FDADJ) {
            // statistics (mref ln 2, sum V) and the unnormalised partial, slot 0 for the contributor that starts the tile, slot 1 for the other
            const float ltot = (lsum + __shfl_xor(lsum, 32, 64)) * __builtin_amdgcn_exp2f(-VSH);
            const int xl = xt * FL_XT + wave * 32 + l31;
            const int64_t prows = (a.ncd / a.C) * a.Bp;
            if (h == 0 && xl < a.Xrows) {
                const int64_t row = (int64_t)dir * a.Bp + (int64_t)c_ * a.Bc + xl;
                const float2 v = make_float2(mref * FL_LN2, ltot);
                if (first) {
                    a.part[row] = v;
                    if (last) a.part[prows + row] = make_float2(-INFINITY, 0.f);
                } else {
                    a.part[prows + row] = v;
                }
            }
            float* dst = first ? a.out : a.out2;
            const int64_t base = (int64_t)dir * a.Bp + (int64_t)c_ * a.Bc;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int col = 32 * ct + l31;
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) {
                    const int x = xt * FL_XT + wave * 32 + acc_row(r_, h);
                    const float v = (TAIL && ct == NCT - 1) ? fl_fold_tail(out[ct][r_]) : out[ct][r_];
                    if (x < a.Xrows && col < a.d) dst[(base + x) * a.out_ld + col] = v * out_unscale;
                }
            }
        } else {
            const bool sole = first && last;
            const int64_t base = (BASE == FLASH_DADJ) ? ((int64_t)dir * a.Bp + (int64_t)c_ * a.Bc) : (a.negocc_off[dir] + (int64_t)c_ * a.N);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int col = 32 * ct + l31;
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) {
                    const int x = xt * FL_XT + wave * 32 + acc_row(r_, h);
                    const float v = (TAIL && ct == NCT - 1) ? fl_fold_tail(out[ct][r_]) : out[ct][r_];
                    if (x < a.Xrows && col < a.d) {
                        float* p = a.out + (base + x) * a.out_ld + col;
                        if (sole) *p = v * out_unscale;
                        else unsafeAtomicAdd(p, v * out_unscale);
                    }
                }
            }
        }
    };

    // ---- stored scores (d > 128 modes): the [Bp, n_ld] fp32 matrix S, read / written in this lane's accumulator layout.  Register 8 s + e of
    // lane (l31, h) is S[adj row][neg column] with (adj row, neg column) = (x, y) when the adj rows are stationary and (y, x) in dNeg, where
    // x = 128 xt + 32 wave + l31 and y = 32 yph + 16 s + 8 h + e: 8 consecutive columns of one row (two 16-B accesses) in the first case, one
    // element of 16 different rows — 32 lanes side by side along a row — in the second.  Entries outside the matrix read as -inf (V = 0).
    auto s_tile = [&](int tile_, int yph_, float (&v)[16], bool store) {
        const int cd_ = tile_ / a.XT, xt_ = tile_ - cd_ * a.XT;
        const int dir_ = cd_ / a.C, cc_ = cd_ - dir_ * a.C;
        const int xq = xt_ * FL_XT + wave * 32 + l31;
        if (a.s_tiled) {
            // tile order: the 128 x 32 scores of (adj tile ta, negative block nb) of a chunk-direction are one contiguous 16 KB run, row-major
            // inside.  Forward / dAdj (adj rows stationary): this item IS such a run — a wave's access is 4 KB contiguous.  dNeg (negative
            // columns stationary): lane l31 is column l31 of negative block (4 xt + wave), register r_ is adj row 32 yph + 16 (r_ >> 3) + 8 h +
            // (r_ & 7) of adj tile yph / 4 — 32 lanes read one 128-B line, the lines of consecutive rows are adjacent.
            if (BASE != FLASH_DNEG) {
                const int nta = a.XT, nnb = a.YB;
                float* base = a.S + ((((int64_t)cd_ * nta + xt_) * nnb + yph_) * FL_XT + wave * 32 + l31) * FL_YB;
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int yl = 16 * s_ + 8 * h + 4 * q;
                        const int yq = yph_ * FL_YB + yl;
                        if (store) {
                            *reinterpret_cast<float4*>(base + yl) = make_float4(v[8 * s_ + 4 * q], v[8 * s_ + 4 * q + 1], v[8 * s_ + 4 * q + 2], v[8 * s_ + 4 * q + 3]);
                        } else {
                            const float4 f = *reinterpret_cast<const float4*>(base + yl);
                            const bool xo = xq < a.Xrows;
                            v[8 * s_ + 4 * q] = (xo && yq < a.Yrows) ? f.x : -INFINITY;
                            v[8 * s_ + 4 * q + 1] = (xo && yq + 1 < a.Yrows) ? f.y : -INFINITY;
                            v[8 * s_ + 4 * q + 2] = (xo && yq + 2 < a.Yrows) ? f.z : -INFINITY;
                            v[8 * s_ + 4 * q + 3] = (xo && yq + 3 < a.Yrows) ? f.w : -INFINITY;
                        }
                    }
                }
            } else {
                // adj tiles / negative blocks as the forward numbered them (here Yrows = adj rows per chunk, Xrows = negatives per chunk)
                const int nta = (a.Yrows + FL_XT - 1) / FL_XT, nnb = (a.Xrows + FL_YB - 1) / FL_YB;
                const int nb = min(xt_ * 4 + wave, nnb - 1);  // clamped: every lane issues every load (the counted waits below rely on it)
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) {
                    const int yrow = yph_ * FL_YB + 16 * (r_ >> 3) + 8 * h + (r_ & 7);
                    const bool ok = xq < a.Xrows && yrow < a.Yrows;
                    v[r_] = ok ? a.S[((((int64_t)cd_ * nta + (yrow >> 7)) * nnb + nb) * FL_XT + (yrow & 127)) * FL_YB + l31] : -INFINITY;
                }
            }
            return;
        }
        if (BASE != FLASH_DNEG) {
            float* row = a.S + ((int64_t)dir_ * a.Bp + (int64_t)cc_ * a.Bc + xq) * a.n_ld;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int y0 = yph_ * FL_YB + 16 * s_ + 8 * h;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int yq = y0 + 4 * q;
                    const bool ok = xq < a.Xrows && yq + 3 < (int)a.n_ld;
                    if (store) {
                        if (ok) *reinterpret_cast<float4*>(row + yq) = make_float4(v[8 * s_ + 4 * q], v[8 * s_ + 4 * q + 1], v[8 * s_ + 4 * q + 2], v[8 * s_ + 4 * q + 3]);
                    } else {
                        float4 f = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                        if (ok) f = *reinterpret_cast<const float4*>(row + yq);
                        v[8 * s_ + 4 * q] = yq < a.Yrows ? f.x : -INFINITY;
                        v[8 * s_ + 4 * q + 1] = yq + 1 < a.Yrows ? f.y : -INFINITY;
                        v[8 * s_ + 4 * q + 2] = yq + 2 < a.Yrows ? f.z : -INFINITY;
                        v[8 * s_ + 4 * q + 3] = yq + 3 < a.Yrows ? f.w : -INFINITY;
                    }
                }
            }
        } else {
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) {
                const int yrow = yph_ * FL_YB + 16 * (r_ >> 3) + 8 * h + (r_ & 7);
                const bool ok = xq < a.Xrows && yrow < a.Yrows;
                v[r_] = ok ? a.S[((int64_t)dir_ * a.Bp + (int64_t)cc_ * a.Bc + yrow) * a.n_ld + xq] : -INFINITY;
            }
        }
    };

    // ---- order of the streamed blocks inside a tile segment.  A workgroup's range covers parts of up to three stationary tiles; within a
    // segment (blocks [lo, lo + len) of one tile) the accumulation order is free, so the sweep is ROTATED to start at the block whose index
    // equals the workgroup-local time: block = lo + ((tau - lo) mod len), tau = items done so far.  Workgroups start together and advance at
    // the same rate, so the 5-6 workgroups of an XCD that sit on the same (chunk, direction) then stream the same block at the same time and
    // share it in that XCD's L2 (unrotated, each starts its sweep where its range happens to start: a block's reuse distance exceeds the L2
    // and every stationary tile re-fetches the whole streamed operand).  a.rotate = 0 (MARIUS_FLASH_ROTATE=0): natural order.
    struct Seg {
        int lo, len, r;
    };
    auto seg_init = [&](int tile_, int it_) {  // it_: first item of this workgroup inside tile_
        Seg g;
        g.lo = it_ - tile_ * a.YB;
        const int hi = min(it1 - tile_ * a.YB, a.YB);
        g.len = hi - g.lo;
        g.r = 0;
        if (a.rotate) {
            g.r = ((it_ - it0) - g.lo) % g.len;
            if (g.r < 0) g.r += g.len;
        }
        return g;
    };
    // ---- prologue: three tiles in flight.  (ptile, pyb) is the prefetch cursor; it stops at the last item of the range.
    int tile = it0 / a.YB, yb = it0 - tile * a.YB;
    int ptile = tile, pyb = yb, pit = it0;
    Seg ps = seg_init(ptile, pit), cs = ps;
    auto padvance = [&]() {
        if (pit + 1 < it1) {
            ++pit;
            if (++pyb == a.YB) {
                pyb = 0;
                ++ptile;
                ps = seg_init(ptile, pit);
            } else {
                ps.r = ps.r + 1 == ps.len ? 0 : ps.r + 1;
            }
        }
    };
    const bool s_loads = SLOAD || (SACC && !a.chunk_first);
    const bool s_lds = (SACC || SLOAD) && a.s_tiled && SSLOTS == NSLOT - 1;
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) {
        if constexpr (SACC || SLOAD) {
            if (s_lds && s_loads) dma_s(ptile, ps.lo + ps.r, i % SSLOTS);
        }
        dma(ptile, ps.lo + ps.r, i);
        padvance();
    }

    // row-major scores (MARIUS_LP_STORE_SCORES, parity runs): the next item's tile through ordinary loads, one item ahead
    float tnext[16];
    if constexpr (SACC || SLOAD) {
        if (!s_lds && s_loads) s_tile(tile, cs.lo + cs.r, tnext, false);
    }

    int slot = 0, pslot = NSLOT - 1;  // ring positions of item `it` and of the tile the next DMA fills
    int sslot = 0;                    // score-tile ring position of item `it` (read out at the top of the item, refilled at once)
    for (int it = it0; it < it1; ++it) {
        if (tile != cur_tile) {
            if (cur_tile >= 0) flush(a.YB);
            load_x(tile);
            cur_tile = tile;
            y_first = yb;
            cs = seg_init(tile, it);
        }
        const int yph = cs.lo + cs.r;  // the streamed block this item multiplies (yb stays the position inside the segment's logical range)
        cs.r = cs.r + 1 == cs.len ? 0 : cs.r + 1;
        // tile `it` has landed once at most the NSLOT - 2 younger tiles' pieces are outstanding (loads retire in order; the x fragments
        // loaded above are younger still, so this over-waits at a tile switch, never under-waits)
        float tcur[16];
        if constexpr (SACC || SLOAD) {
            if (s_lds) {
                // issued after this item's requests (its score tile, then its block): the requests of the NSLOT - 2 items behind it and the
                // score stores of the (at most NSLOT - 1) items finished since; in-order retirement turns `at most that many outstanding` into
                // `this item has landed`
                constexpr int AHEAD = NSLOT - 2;  // whole items requested after this one
                const int k = SACC ? min(it - it0, NSLOT - 1) : 0;
                if (s_loads) {
                    if (k == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * (S_DMA + DMA_PER_WAVE)) : "memory");
                    else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * (S_DMA + DMA_PER_WAVE) + 4) : "memory");
                    else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * (S_DMA + DMA_PER_WAVE) + 8) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * (S_DMA + DMA_PER_WAVE) + 12) : "memory");
                } else {
                    if (k == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * DMA_PER_WAVE) : "memory");
                    else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * DMA_PER_WAVE + 4) : "memory");
                    else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * DMA_PER_WAVE + 8) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * DMA_PER_WAVE + 12) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (s_loads) {
                    s_take(xt, yph, sslot, tcur);
                    dma_s(ptile, ps.lo + ps.r, sslot);  // the slot just read out takes the tile of the item the block DMA below fetches
                    sslot = sslot + 1 == SSLOTS ? 0 : sslot + 1;
                }
            } else {
                // ordinary loads one item ahead; everything (this item's streamed block included) is drained at the top
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (s_loads) {
#pragma unroll
                    for (int r_ = 0; r_ < 16; ++r_) tcur[r_] = tnext[r_];
                    if (it + 1 < it1) {  // next item: same tile (cs has already advanced) or the first block of the next tile's segment
                        int n_tile = tile, n_yb = yb + 1;
                        if (n_yb == a.YB) ++n_tile;
                        const Seg ns = (n_tile != tile) ? seg_init(n_tile, it + 1) : cs;
                        s_tile(n_tile, ns.lo + ns.r, tnext, false);
                    }
                }
            }
        } else {
            if (short_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * (DMA_PER_WAVE - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * DMA_PER_WAVE) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        // the next block's DMA: FL_DMA_INTERLEAVE spreads its pieces over the first k-steps of this item's score contraction (the order of the
        // loads and the counted waits are unchanged: all pieces are issued before the next item's wait)
        constexpr bool DMA_IL = FL_DMA_INTERLEAVE && !SLOAD && !SACC && KS >= FL_DMA_KS0 + FL_DMA_STRIDE * (DMA_PER_WAVE - 1) + 1;
        const char* nsrc = dma_src(ptile, ps.lo + ps.r);
        const unsigned ndst = dma_dst(pslot);
        if constexpr (!DMA_IL) {
#pragma unroll
            for (int i = 0; i < DMA_PER_WAVE; ++i)
                if (!(short_wave && i == DMA_PER_WAVE - 1)) dma_piece(nsrc, ndst, i);
        }
        padvance();
        pslot = pslot + 1 == NSLOT ? 0 : pslot + 1;
        const unsigned char* T = smem + slot * SLOT;
        uint32_t fo0 = 0u, fo1 = 0u;  // filter entries of this item (loaded ahead of the matrix phase that hides the latency)
        if (BASE != FLASH_FWD && a.foff) {
            const int64_t key = ((int64_t)cd * a.XT + xt) * a.YB + yph;
            fo0 = a.foff[key];
            fo1 = a.foff[key + 1];
        }

        // ---- S tile: D[y][x] = sum_k Y[y][k] X[x][k]
        v16f accS;
        if constexpr (!SLOAD)
            accS = fl_score_tile<KS, F16, TAIL>(T, a_off, xh, xl, [&](int ks) {
                if constexpr (DMA_IL) {
                    if (ks >= FL_DMA_KS0 && (ks - FL_DMA_KS0) % FL_DMA_STRIDE == 0 && (ks - FL_DMA_KS0) / FL_DMA_STRIDE < DMA_PER_WAVE) {
                        if (!(short_wave && (ks - FL_DMA_KS0) / FL_DMA_STRIDE == DMA_PER_WAVE - 1)) dma_piece(nsrc, ndst, (ks - FL_DMA_KS0) / FL_DMA_STRIDE);
                    }
                }
            });

        if (BASE == FLASH_FWD) {
            float t[16];
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) t[r_] = accS[r_];
            if constexpr (SACC) {  // S (true units) = what the earlier chunks left + this chunk's part; every chunk but the last only stores it
                const float invs = 1.f / (sc_.s_adj * sc_.s_neg);
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) t[r_] = a.chunk_first ? t[r_] * invs : fmaf(t[r_], invs, tcur[r_] == -INFINITY ? 0.f : tcur[r_]);
                s_tile(tile, yph, t, true);
            }
            if (SACC && !a.chunk_last) {
                // nothing else to do for this item
            } else {
            if (STORE_S) {  // parity / debug only: scattered 4-B stores
                const int x = xt * FL_XT + wave * 32 + l31;
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) {
                    const int y = yph * FL_YB + 16 * (r_ >> 3) + 8 * h + (r_ & 7);
                    if (x < a.Xrows && y < a.Yrows)
                        a.S[((int64_t)dir * a.Bp + (int64_t)c_ * a.Bc + x) * a.n_ld + y] = t[r_] * inv_xy;
                }
            }
            if ((yph + 1) * FL_YB > a.Yrows) {  // last block of the chunk: columns past N do not exist
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) {
                    const int y = yph * FL_YB + 16 * (r_ >> 3) + 8 * h + (r_ & 7);
                    if (y >= a.Yrows) t[r_] = -INFINITY;
                }
            }
            float tm = t[0];
#pragma unroll
            for (int r_ = 1; r_ < 16; ++r_) tm = fmaxf(tm, t[r_]);
            const float mn = fmaxf(m2, tm * c_s);
            float s0 = 0.f, s1 = 0.f;
            if (mn > -INFINITY) {
#pragma unroll
                for (int r_ = 0; r_ < 16; r_ += 2) {
                    s0 += __builtin_amdgcn_exp2f(fmaf(t[r_], c_s, -mn));
                    s1 += __builtin_amdgcn_exp2f(fmaf(t[r_ + 1], c_s, -mn));
                }
                lsum = lsum * __builtin_amdgcn_exp2f(m2 - mn) + (s0 + s1);
                m2 = mn;
            }
            }
        } else {
            float t[16];
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) t[r_] = SLOAD ? tcur[r_] : accS[r_];
            if (fo1 != fo0) {  // filtered scores of this item: the reference overwrites them with -1e9 before the loss
                for (uint32_t q = fo0; q < fo1; ++q) {
                    const int ent = a.fent[q];
                    const int xl = ent >> 5, yl = ent & 31;
                    const bool mine = (wave == (xl >> 5)) && (l31 == (xl & 31)) && (h == ((yl >> 3) & 1));
                    const int reg = 8 * (yl >> 4) + (yl & 7);
#pragma unroll
                    for (int r_ = 0; r_ < 16; ++r_)
                        if (mine && r_ == reg) t[r_] = -INFINITY;  // exp(-1e9 - lse) is exactly 0 in the reference's fp32 as well
                }
            }
            if (MODE == FLASH_FDADJ) {
                // streamed rows past N do not exist: they must not enter sum V (their records are zero rows: S = 0, not -inf)
                if ((yph + 1) * FL_YB > a.Yrows) {
#pragma unroll
                    for (int r_ = 0; r_ < 16; ++r_) {
                        const int y = yph * FL_YB + 16 * (r_ >> 3) + 8 * h + (r_ & 7);
                        if (y >= a.Yrows) t[r_] = -INFINITY;
                    }
                }
                float tm = t[0];
#pragma unroll
                for (int r_ = 1; r_ < 16; ++r_) tm = fmaxf(tm, t[r_]);
                tm = fmaxf(tm, __shfl_xor(tm, 32, 64)) * c_s;  // this block's maximum of row x = l31, both halves (log2 units)
                const bool raise = tm > mref + FL_TAU;
                if (__builtin_amdgcn_ballot_w64(raise) != 0ull) {    // rare: a reference moves, the row's sums are rescaled to it
                    const float mnew = raise ? tm : mref;
                    const float alpha = __builtin_amdgcn_exp2f(mref - mnew);
                    lsum *= alpha;
                    mref = mnew;
#pragma unroll
                    for (int r_ = 0; r_ < 16; ++r_) {
                        const float ax = __shfl(alpha, acc_row(r_, h), 64);  // accumulator register r_ of this lane belongs to row acc_row(r_, h)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) out[ct][r_] *= ax;
                    }
                }
            }
            // ---- V = exp2(S log2(e) - lsec) in the accumulator layout == A operand layout (k = 16 s + 8 h + e <-> reg 8 s + e).  V <= 2^14 by
            // construction (the reference of the fused sweep, lse elsewhere), so the packed conversions need no saturation.
            union { v8bf v; unsigned u[4]; } wh[2], wl[2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    float w2[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int r_ = 8 * s_ + e + q;
                        if constexpr (MODE == FLASH_FDADJ) {  // against the row's running reference (the same rounding in the sum and in the partial: it cancels)
                            w2[q] = __builtin_amdgcn_exp2f(fmaf(t[r_], c_s, VSH - mref));
                            lsum += w2[q];
                        } else {  // against lsec - VSH, carried as (hi, lo): its fp32 representation error would be a relative error of V
                            const float2 ls = BASE == FLASH_DADJ ? lsec_x : *reinterpret_cast<const float2*>(T + (16 * s_ + fl_rho(8 * h + e + q)) * P + LSEC);
                            w2[q] = __builtin_amdgcn_exp2f(fmaf(t[r_], c_s, -ls.x) - ls.y);
                        }
                    }
                    const unsigned hi = fl_cvt16x2<F16>(w2[0], w2[1]);
                    wh[s_].u[e >> 1] = hi;
                    float lo0, lo1;
                    fl_lo_pair<F16>(w2[0], w2[1], hi, lo0, lo1);
                    wl[s_].u[e >> 1] = fl_cvt16x2<F16>(lo0, lo1);

                }
            }
            // ---- out[x][col] += sum_y V[y][x] Y[y][col]
            if constexpr (BASE != FLASH_FWD) {
                const v8bf whv[2] = {wh[0].v, wh[1].v}, wlv[2] = {wl[0].v, wl[1].v};
                fl_grad_tile<KS, NCT, F16, TAIL>(T, tr_off, whv, wlv, out);
            }
        }
        if (++yb == a.YB) { yb = 0; ++tile; }
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    flush(yb == 0 ? a.YB : yb);
    // drain the over-issued prefetches before the LDS allocation is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------- merge
// lse = log(e^pos + sum over the (at most two) partials), row loss, dL/dpos, per-block loss sums; the contributors' weights g exp(m_k - lse);
// patches lsec into the adj records.
// Round 6: in FLOAT64, and lsec as a (hi, lo) float pair.  Found by the arithmetic check's trained-table input (flat softmax over 1001 scores
// near zero: lse ~ 6.9, lsec ~ 10): V = exp2(S log2(e) - lsec) inherits the ABSOLUTE error of lsec as a RELATIVE error, and an fp32 lsec of
// magnitude 10 carries 4.8e-7 of representation error on top of the fp32 roundings of m + log(sum) — 3e-7 RMS on every gradient entry, three
// times the reference's own fp32 evaluation, while the scores themselves were twice as accurate as the reference's.  100,000 rows of a few
// double-precision transcendentals cost nothing next to the launches either side.
__global__ __launch_bounds__(256) void flash_merge_kernel(const float2* __restrict__ part, const float* __restrict__ pos, int64_t rows, int64_t Bp,
                                                          int Bc, int C, int XR, int KP, float* __restrict__ lse, float* __restrict__ rowloss,
                                                          float* __restrict__ dpos, float gscale, float* __restrict__ blocksum, char* __restrict__ adjrec,
                                                          int nsets, int64_t set_bytes, int P, float* __restrict__ weights, float vsh) {
    __shared__ float red[256];
    const int64_t bpd = (Bp + 255) / 256;
    const int64_t dir = blockIdx.x / bpd, blk = blockIdx.x - dir * bpd;
    const int64_t r = blk * 256 + threadIdx.x;
    float mine = 0.f;
    if (r < Bp) {
        const int64_t row = dir * Bp + r;
        const double p = (double)pos[row];
        const float2 v0 = part[row], v1 = part[rows + row];
        double m = p;
        if (v0.y > 0.f) m = fmax(m, (double)v0.x);
        if (v1.y > 0.f) m = fmax(m, (double)v1.x);
        const double e0 = v0.y > 0.f ? (double)v0.y * exp((double)v0.x - m) : 0.0, e1 = v1.y > 0.f ? (double)v1.y * exp((double)v1.x - m) : 0.0;
        const double sum = exp(p - m) + e0 + e1;
        const double l = m + log(sum);
        lse[row] = (float)l;
        rowloss[row] = (float)(l - p);
        // dL/dpos = p_pos - 1 = -(sum of the negatives' probabilities): the second form has no cancellation when p_pos -> 1
        dpos[row] = (float)(-((e0 + e1) / sum) * (double)gscale);
        mine = (float)(l - p);
        // dL/dadj = w0 O0 + w1 O1 over the contributors' unnormalised partials O_k = sum_j exp(s_j - m_k) neg_j:  w_k = g exp(m_k - lse) = g exp(m_k - m) / sum
        weights[row] = v0.y > 0.f ? (float)((double)gscale * exp((double)v0.x - m) / sum) : 0.f;
        weights[rows + row] = v1.y > 0.f ? (float)((double)gscale * exp((double)v1.x - m) / sum) : 0.f;
        const int64_t c = r / Bc;
        const int x = (int)(r - c * Bc);
        // lsec - VSH (the binades V is formed up by: flash_kernel) as a float pair: the kernels compute exp2(fma(S, c, -hi) - lo)
        const double ls = l * 1.4426950408889634074 - log2((double)gscale) - (double)vsh;
        const float hi = (float)ls, lo = (float)(ls - (double)hi);
        for (int q = 0; q < nsets; ++q)  // every column chunk's record set carries the row's lsec (the 16 bytes that close a record)
            *reinterpret_cast<float2*>(adjrec + q * set_bytes + ((dir * C + c) * XR + fl_rho(x)) * (int64_t)P + P - 16) = make_float2(hi, lo);
    }
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) blocksum[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------- host side
// Column chunks of the contraction index.  d <= 128: one, the stationary operand's whole K depth lives in registers next to the gradient
// accumulators (KS <= 8).  Wider rows take the stored-score modes, which hold EITHER the K fragments (FWDS) OR the accumulators (DADJS / DNEGS)
// and therefore carry chunks of up to 256 columns (KS <= 16; round 4 — round 3 cut at 128: 400 = 4 x 100): ceil(d / 256) equal chunks, each a
// multiple of 4 columns (400 = 2 x 200).  Every chunk costs one pass over the stored scores in each of the three launches, so halving their number
// takes 3.0 -> 1.4 GB per step off cfg5's shape.  0 = this d cannot be cut that way.  MARIUS_FLASH_WIDE=0: rows wider than 128 columns stay on
// the FP32 matrix path; =128: the round-3 chunk width (A/B runs).
int flash_chunks(int d) {
    if (d <= 128) return 1;
    const int fw = kernel_env().flash_wide;
    if (fw == 0) return 0;
    const int w = fw == 128 ? 128 : 256;
    const int n = (d + w - 1) / w;
    return (d % (4 * n) == 0 && d <= 1024) ? n : 0;
}
static int fl_kc(int d) { const int n = flash_chunks(d); return n > 0 ? d / n : d; }  // columns per chunk
static int fl_ks(int d) { return (fl_kc(d) + 15) / 16; }
// folded column tail (see fl_pitch): d = 4 (mod 16) with an odd number of k-steps — d = 36, 68, 100
bool flash_tail4(int d) { return d <= 128 && (d & 15) == 4 && (fl_ks(d) & 1) && fl_ks(d) >= 3 && !kernel_env().flash_tail4_off; }
static int fl_pitch_d(int d) { return fl_pitch(fl_ks(d), flash_tail4(d)); }
bool flash_chunked(int d) { return d > 128 && flash_chunks(d) >= 1; }  // stored-score modes (possibly with a single chunk)
// bytes of the stored scores in tile order (see s_tile in flash_kernel): whole 128 x 32 tiles
size_t flash_tiled_scores_bytes(const LpDims& D) {
    const size_t xt = (D.Bc + FL_XT - 1) / FL_XT, yb = ((D.N + 31) / 32 * 32) / FL_YB;
    return (size_t)D.ndir * D.C * xt * yb * FL_XT * FL_YB * sizeof(float);
}
static size_t fl_adjset_bytes(const LpDims& D) { return (size_t)D.ndir * D.C * ((D.Bc + 31) / 32 * 32) * fl_pitch_d(D.d); }
static size_t fl_negset_bytes(const LpDims& D) { return (size_t)D.ndir * D.C * ((D.N + 31) / 32 * 32) * fl_pitch_d(D.d); }

// score-filter index (flash_filter_index_kernel): item counts of the two orientations, entry capacity, bytes behind the statistics in `fpart`
struct FlFilterDims {
    int XTa, YBa, XTn, YBn, nkeys_max;
    int64_t ent_cap;
    size_t off_bytes, bytes;
};
static FlFilterDims fl_filter_dims(const LpDims& D) {
    FlFilterDims f;
    const int XRa = (D.Bc + 31) / 32 * 32, NRn = (D.N + 31) / 32 * 32;
    f.XTa = (D.Bc + FL_XT - 1) / FL_XT;
    f.YBa = NRn / FL_YB;
    f.XTn = (D.N + FL_XT - 1) / FL_XT;
    f.YBn = XRa / FL_YB;
    const int64_t ka = (int64_t)D.ndir * D.C * f.XTa * f.YBa, kn = (int64_t)D.ndir * D.C * f.XTn * f.YBn;
    f.nkeys_max = (int)(ka > kn ? ka : kn);
    f.ent_cap = (int64_t)D.ndir * D.C * D.N;  // a DEG filter has at most one entry per (direction, chunk, negative column) (negative.cpp:21-39)
    f.off_bytes = ((size_t)2 * (f.nkeys_max + 1) * 4 + 255) / 256 * 256;
    f.bytes = f.off_bytes + ((size_t)2 * f.ent_cap * 2 + 255) / 256 * 256;
    return f;
}
constexpr size_t FL_FILTER_LDS_MAX = 150 * 1024;

bool flash_applicable(const marius_lp_desc* desc, const LpDims& D) {
    const char fe = kernel_env().flash;
    if (fe == '0') return false;
    if (!(desc->flags & MARIUS_LP_TRAIN_ONLY) && fe != 'f') return false;  // MARIUS_FLASH=f: force (tests of the API path's numbers)
    if (D.loss != MARIUS_LOSS_SOFTMAX_CE || D.cmp != MARIUS_CMP_DOT) return false;
    if ((desc->dst_filter && desc->n_dst_filter > 0) || (desc->src_filter && desc->n_src_filter > 0)) {
        // score filters (training: the DEG filter of degree-based negatives) are honoured by the fused sweep and dNeg through a per-item index;
        // not by the round-2 three-launch form, not together with stored scores, and only for lists a DEG filter can produce
        const FlFilterDims f = fl_filter_dims(D);
        if ((desc->flags & MARIUS_LP_STORE_SCORES) || (size_t)(f.nkeys_max + 1) * 4 > FL_FILTER_LDS_MAX ||
            desc->n_dst_filter + desc->n_src_filter > f.ent_cap || flash_chunked(D.d))  // (the column-chunked launches of d > 128 do not look filters up)
            return false;
    }
    const int ks = fl_ks(D.d);
    if (ks < 2 || ks > (flash_chunked(D.d) ? 16 : 8)) return false;  // instantiated K depths: d in (16, 128]; chunks of up to 256 columns in the stored-score modes
    if (D.d > 128 && !flash_chunked(D.d)) return false;
    if (D.ndir == 2 && !desc->src_neg) return false;
    return true;
}

size_t flash_adjrec_bytes(const LpDims& D) { return fl_adjset_bytes(D) * flash_chunks(D.d) + 32768; }   // one record set per column chunk
size_t flash_negrec_bytes(const LpDims& D) { return fl_negset_bytes(D) * flash_chunks(D.d) + 32768; }
// [2][ndir Bp] (mref, sum V) pairs of a row's (at most two) contributors, then [2][ndir Bp] floats: the weights g exp(m_k - lse) the merge kernel
// derives from them in float64 (what the edge backward multiplies the unnormalised dAdj partials with)
static size_t fl_pairs_bytes(const LpDims& D) { return (size_t)2 * D.ndir * D.Bp * sizeof(float2); }
static size_t fl_stats_bytes(const LpDims& D) { return (fl_pairs_bytes(D) + (size_t)2 * D.ndir * D.Bp * sizeof(float) + 255) / 256 * 256; }
const float* flash_part_weights(const LpDims& D, const float2* part) { return (const float*)((const char*)part + fl_pairs_bytes(D)); }
// [statistics | filter index (offsets of both orientations, entries of both orientations)]
size_t flash_part_bytes(const LpDims& D) { return fl_stats_bytes(D) + fl_filter_dims(D).bytes; }
// forward statistics and dAdj are one sweep (FLASH_FDADJ); the two-launch form of round 2 lost its A/B run (0.726 vs 0.653 ms per step) and is gone
bool flash_fused() { return true; }

// How many persistent workgroups a launch gets.  Every workgroup walks a contiguous range of total / nwg items, i.e. tiles / nwg stationary
// tiles: when that ratio is a fraction p / q with a small denominator the ranges repeat every q workgroups — neighbours start and finish
// their tiles in step, the two contributors of a split tile meet at the same few cut points, and the rotated sweeps of a (chunk, direction)
// stay aligned in that XCD's L2.  Measured at the bench shape (800 tiles; profiles/r4_flash_nwg_sweep.txt): 512 workgroups (25 / 16 tiles
// each) 0.623 ms per step, 496 0.627, 488 0.616, 480 (5 / 3) 0.599, 472 0.612, 464 0.612, 448 0.617, 400 (2 / 1, no split tiles but 22 % of
// the slots empty) 0.618.  So: among the multiples of 8 within 15 % below the slot count, the one whose tiles / nwg has the smallest
// denominator (ties: the larger).  The CUs this leaves without a workgroup (16 of 256 at the bench shape) are not wasted either: the step's
// other stream — the next batch's sort / sample / plan kernels — otherwise only gets CU slots in the tails of the matrix launches
// (profiles/r4_timeline_one_step_before_workgroup_rule.txt: a 10 us sort sweep took 117-132 us underneath them).
// marius_lp_desc.free_cus (MARIUS_FLASH_RESERVE overrides it): CUs to leave empty on top of that — a caller that runs other streams beside the
// matrix launches (the sharded trainer) asks for them; MARIUS_FLASH_NWG: the count itself (tests of every split pattern).

static int64_t fl_gcd(int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; }
static int fl_device_cus() {  // compute units of the current device (256 on an MI355X), asked once
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
static int fl_num_wg(int64_t tiles, int mode, int ks, int free_cus) {
    const int per_cu = fl_wg_per_cu_ks(mode, ks);
    const KernelEnv& ke = kernel_env();
    int reserve = free_cus;
    if (ke.has_flash_reserve) reserve = ke.flash_reserve;
    const int cus = fl_device_cus();
    if (reserve < 0 || reserve >= cus / 2) reserve = 0;
    int nwg = (cus - reserve) * per_cu;
    if (ke.has_flash_nwg) {
        nwg = ke.flash_nwg;
    } else if (nwg < tiles && per_cu > 1) {  // (the one-workgroup-per-CU launches of wide chunks are bound by their score traffic: 256 beats 240 there, 1.402 vs 1.418 ms)
        int best = nwg & ~7;
        int64_t best_q = best / fl_gcd(tiles, best);
        for (int n = best - 8; n >= 8 && n * 100 >= nwg * 85; n -= 8) {
            const int64_t q = n / fl_gcd(tiles, n);
            if (q < best_q) { best = n; best_q = q; }
        }
        nwg = best;
    }
    if (nwg > tiles) nwg = (int)tiles;
    if (nwg >= 8) nwg &= ~7;  // XCD-local ranges need a multiple of 8
    return nwg < 1 ? 1 : nwg;
}

template <int KS, int MODE, bool STORE_S, bool F16, bool TAIL>
static int fl_launch_t(const FlashArgs& a, hipStream_t st) {
    const size_t lds = (size_t)fl_slots(MODE) * fl_slot_bytes(KS, TAIL) + fl_sring_bytes(MODE);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_kernel<KS, MODE, STORE_S, F16, TAIL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_last_error("flash: cannot raise the dynamic LDS limit to %zu", lds);
            return MARIUS_ERR_HIP;
        }
        attr_done = true;
    }
    flash_kernel<KS, MODE, STORE_S, F16, TAIL><<<dim3((unsigned)a.nwg), dim3(FL_NT), lds, st>>>(a);
    return check_launch("flash_kernel");
}
template <int KS, int MODE, bool STORE_S>
static int fl_launch(const FlashArgs& a, hipStream_t st) {
    if constexpr ((KS & 1) && KS >= 3 && KS <= 8 && MODE < FLASH_FWDS) {
        if (flash_tail4(a.d) && (a.d + 15) / 16 == KS)
            return a.rg.absmax ? fl_launch_t<KS, MODE, STORE_S, true, true>(a, st) : fl_launch_t<KS, MODE, STORE_S, false, true>(a, st);
    }
    return a.rg.absmax ? fl_launch_t<KS, MODE, STORE_S, true, false>(a, st) : fl_launch_t<KS, MODE, STORE_S, false, false>(a, st);
}

template <int MODE, bool STORE_S>
static int fl_dispatch(int ks, const FlashArgs& a, hipStream_t st) {
    if constexpr (MODE >= FLASH_FWDS) {  // stored-score modes: chunks of up to 256 columns
        switch (ks) {
            case 9: return fl_launch<9, MODE, STORE_S>(a, st);
            case 10: return fl_launch<10, MODE, STORE_S>(a, st);
            case 11: return fl_launch<11, MODE, STORE_S>(a, st);
            case 12: return fl_launch<12, MODE, STORE_S>(a, st);
            case 13: return fl_launch<13, MODE, STORE_S>(a, st);
            case 14: return fl_launch<14, MODE, STORE_S>(a, st);
            case 15: return fl_launch<15, MODE, STORE_S>(a, st);
            case 16: return fl_launch<16, MODE, STORE_S>(a, st);
        }
    }
    switch (ks) {
        case 2: return fl_launch<2, MODE, STORE_S>(a, st);
        case 3: return fl_launch<3, MODE, STORE_S>(a, st);
        case 4: return fl_launch<4, MODE, STORE_S>(a, st);
        case 5: return fl_launch<5, MODE, STORE_S>(a, st);
        case 6: return fl_launch<6, MODE, STORE_S>(a, st);
        case 7: return fl_launch<7, MODE, STORE_S>(a, st);
        case 8: return fl_launch<8, MODE, STORE_S>(a, st);
    }
    set_last_error("flash: unsupported K depth %d", ks);
    return MARIUS_ERR_UNSUPPORTED;
}

static FlRange g_no_range = {nullptr, nullptr, FL_ADJ_NODE};
// magnitude bounds of the caller's tables -> fp16 records (marius_lp_desc.absmax); MARIUS_FLASH_F16=0 keeps the bf16 records
FlRange flash_range(const marius_lp_desc* desc, const LpDims& D) {
    FlRange r = g_no_range;
    if (desc->absmax && !kernel_env().flash_f16_off) {
        r.absmax = desc->absmax;
        r.absmax_rel = desc->absmax_rel ? desc->absmax_rel : desc->absmax + 1;
        const bool has_rel = D.edge_cols == 3 && desc->rel;
        r.adj_bound = !has_rel ? FL_ADJ_NODE
                      : D.relop == MARIUS_OP_HADAMARD ? FL_ADJ_PRODUCT
                      : D.relop == MARIUS_OP_COMPLEX_HADAMARD ? FL_ADJ_PRODUCT2
                      : D.relop == MARIUS_OP_TRANSLATION ? FL_ADJ_SUM : FL_ADJ_NODE;
    }
    return r;
}
static void fl_common(FlashArgs& a, const LpDims& D, int mode, char* adjrec, char* negrec, const FlRange& rg, int free_cus) {
    a.rg = rg;
    const int xt_rows = FL_XT;
    const int XRa = (D.Bc + 31) / 32 * 32, NRn = (D.N + 31) / 32 * 32;
    const bool xadj = (fl_base(mode) != FLASH_DNEG);
    a.pos = nullptr;
    a.out2 = nullptr;
    a.chunk_first = a.chunk_last = 1;
    a.s_tiled = 0;
    a.foff = nullptr;
    a.fent = nullptr;
    a.xrec = xadj ? adjrec : negrec;
    a.yrec = xadj ? negrec : adjrec;
    a.XR = xadj ? XRa : NRn;
    a.YR = xadj ? NRn : XRa;
    a.Xrows = xadj ? D.Bc : D.N;
    a.Yrows = xadj ? D.N : D.Bc;
    a.ncd = D.C * D.ndir;
    a.XT = (a.Xrows + xt_rows - 1) / xt_rows;
    a.YB = a.YR / FL_YB;
    a.total = (int64_t)a.ncd * a.XT * a.YB;
    a.nwg = fl_num_wg((int64_t)a.ncd * a.XT, mode, fl_ks(D.d), free_cus);
    a.rotate = kernel_env().flash_rotate_off ? 0 : 1;
    a.C = D.C;
    a.Bc = D.Bc;
    a.N = D.N;
    a.d = D.d;
    a.Bp = D.Bp;
    a.part = nullptr;
    a.S = nullptr;
    a.n_ld = D.n_ld;
    a.out = nullptr;
    a.out_ld = 0;
    a.negocc_off[0] = a.negocc_off[1] = 0;
}

// forward: pack both operands, row statistics (and, for parity tests only, the scores themselves)
// pos / dadj / dadj2: fused form only (flash_fused()): the sweep also leaves the unnormalised dAdj partials
int flash_forward(const marius_lp_desc* desc, const LpDims& D, const float* adj, char* adjrec, char* negrec, float2* part, float* S, bool adj_packed,
                  float* gocc, const int64_t negocc_off[2], float* dadj_zero, const float* pos, float* dadj, float* dadj2, hipStream_t st) {
    const bool s_tiled = flash_chunked(D.d) && !(desc->flags & MARIUS_LP_STORE_SCORES);  // nobody outside reads the scores: keep them in tile order
    const FlRange rg = flash_range(desc, D);
    const int ks = fl_ks(D.d), KP = 16 * ks;
    const int XR = (D.Bc + 31) / 32 * 32, NR = (D.N + 31) / 32 * 32;
    const int ppr = KP / 4 + 1;
    const int nch = flash_chunks(D.d), kc = fl_kc(D.d);  // column chunks of the contraction index (1 for d <= 128) and their width
    const size_t adjset = fl_adjset_bytes(D), negset = fl_negset_bytes(D);
    const bool wide = flash_chunked(D.d);
    MARIUS_REQUIRE(!wide || (!adj_packed && adj && S), "flash: rows wider than 128 need the fp32 adj rows and the score matrix");
    for (int c = 0; c < nch; ++c) {
        if (!adj_packed) {
            const int64_t n = (int64_t)D.ndir * D.C * XR * ppr;
            flash_pack_adj_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st>>>(adj, D.d_ld, D.Bp, D.Bc, D.C, D.ndir, kc, KP, XR, adjrec + c * adjset, rg, c * kc,
                                                                                      flash_tail4(D.d) ? 1 : 0);
        }
        const int64_t n = (int64_t)D.ndir * D.C * NR * ppr;
        ProfScope ps(PROF_LP_PACK, st);
        flash_pack_neg_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st>>>(desc->emb, desc->emb_ld, desc->dst_neg, desc->src_neg, D.N, D.C, D.ndir,
                                                                                 kc, KP, NR, row_vec_width(desc->emb, desc->emb_ld, 4), negrec + c * negset, gocc,
                                                                                 D.d_ld, negocc_off[0], negocc_off[1], rg, c * kc, flash_tail4(D.d) ? 1 : 0);
    }
    if (!adj_packed && (dadj_zero || wide)) flash_zero_kernel<<<dim3(1024), dim3(256), 0, st>>>(wide ? dadj : dadj_zero, D.ndir * D.Bp * D.d_ld, nullptr, 0, nullptr, 0);
    int rc = check_launch("flash_pack");
    if (rc) return rc;
    FlashArgs a;
    if (wide) {  // S += adj_c neg_c^T chunk by chunk; the last launch also leaves the row statistics
        ProfScope ps(PROF_LP_SCORES, st);
        for (int c = 0; c < nch && !rc; ++c) {
            fl_common(a, D, FLASH_FWDS, adjrec + c * adjset, negrec + c * negset, rg, desc->free_cus);
            a.part = part;
            a.S = S;
            a.chunk_first = c == 0;
            a.chunk_last = c == nch - 1;
            a.s_tiled = s_tiled ? 1 : 0;
            rc = fl_dispatch<FLASH_FWDS, false>(ks, a, st);
        }
        return rc;
    }
    if (S) {  // statistics-only sweep of the score-storing parity runs (its statistics are then rewritten below)
        fl_common(a, D, FLASH_FWD, adjrec, negrec, rg, desc->free_cus);
        a.part = part;
        a.S = S;
        ProfScope ps(PROF_LP_SCORES, st);
        rc = S ? fl_dispatch<FLASH_FWD, true>(ks, a, st) : fl_dispatch<FLASH_FWD, false>(ks, a, st);
        if (rc) return rc;
    }
    fl_common(a, D, FLASH_FDADJ, adjrec, negrec, rg, desc->free_cus);
    a.part = part;
    a.pos = pos;
    a.out = dadj;
    a.out2 = dadj2;
    a.out_ld = D.d_ld;
    const int64_t nf0 = desc->dst_filter ? desc->n_dst_filter : 0, nf1 = (D.ndir == 2 && desc->src_filter) ? desc->n_src_filter : 0;
    if (nf0 + nf1 > 0) {
        const FlFilterDims f = fl_filter_dims(D);
        MARIUS_REQUIRE(nf0 + nf1 <= f.ent_cap, "flash: the score filter has %ld entries, more than a DEG filter of this batch shape can (%ld)", (long)(nf0 + nf1), (long)f.ent_cap);
        char* fbase = (char*)part + fl_stats_bytes(D);
        uint32_t* foff = (uint32_t*)fbase;
        uint16_t* fent = (uint16_t*)(fbase + f.off_bytes);
        const size_t lds = (size_t)(f.nkeys_max + 1) * 4;
        static bool fattr = false;
        if (!fattr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_filter_index_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FL_FILTER_LDS_MAX) != hipSuccess) {
                set_last_error("flash: cannot raise the dynamic LDS limit of the filter index kernel");
                return MARIUS_ERR_HIP;
            }
            fattr = true;
        }
        flash_filter_index_kernel<<<dim3(2), dim3(1024), lds, st>>>(desc->dst_filter, nf0, desc->src_filter, nf1, D.Bc, D.C, D.N, D.ndir, f.XTa, f.YBa, f.XTn, f.YBn,
                                                                   f.nkeys_max, foff, fent, f.ent_cap);
        rc = check_launch("flash_filter_index");
        if (rc) return rc;
        a.foff = foff;
        a.fent = fent;
    }
    ProfScope ps(PROF_LP_GRAD_ADJ, st);  // accounted as the dAdj launch: 2 contractions (scores + V Neg)
    return fl_dispatch<FLASH_FDADJ, false>(ks, a, st);
}

int flash_merge(const LpDims& D, const float2* part, const float* pos, float* lse, float* rowloss, float* dpos, float* blocksum, char* adjrec,
                bool f16, hipStream_t st) {
    const int64_t bpd = cdiv(D.Bp, 256);
    const int ks = fl_ks(D.d);
    flash_merge_kernel<<<dim3((unsigned)(bpd * D.ndir)), dim3(256), 0, st>>>(part, pos, D.Bp * D.ndir, D.Bp, D.Bc, D.C, (D.Bc + 31) / 32 * 32, 16 * ks, lse,
                                                                            rowloss, dpos, D.gscale, blocksum, adjrec, flash_chunks(D.d), (int64_t)fl_adjset_bytes(D), fl_pitch_d(D.d),
                                                                            const_cast<float*>(flash_part_weights(D, part)), f16 ? 14.f : 0.f);
    return check_launch("flash_merge");
}

// backward contractions: dadj [ndir][Bp][d_ld] and the negatives' gocc rows.  Both outputs are zeroed first (split tiles accumulate).
// dadj and the negatives' gocc rows were zeroed by the forward's pack kernels (split tiles accumulate onto them)
// part / filtered: fused form with a score filter — the index the forward built (orientation 1) sits behind the statistics
int flash_backward(const marius_lp_desc* desc, const LpDims& D, char* adjrec, char* negrec, float* dadj, float* gocc, const int64_t negocc_off[2],
                   const float2* part, bool filtered, float* S, hipStream_t st) {
    const FlRange rg = flash_range(desc, D);
    const int ks = fl_ks(D.d);
    FlashArgs a;
    int rc = MARIUS_OK;
    if (flash_chunked(D.d)) {  // V = exp(S - lse) from the stored scores; one (dAdj, dNeg) pair of launches per column chunk, each writing its columns
        MARIUS_REQUIRE(S, "flash: rows wider than 128 need the score matrix");
        const int nch = flash_chunks(D.d), kc = fl_kc(D.d);
        const size_t adjset = fl_adjset_bytes(D), negset = fl_negset_bytes(D);
        for (int c = 0; c < nch && !rc; ++c) {
            fl_common(a, D, FLASH_DADJS, adjrec + c * adjset, negrec + c * negset, rg, desc->free_cus);
            a.S = S;
            a.s_tiled = (desc->flags & MARIUS_LP_STORE_SCORES) ? 0 : 1;
            a.d = kc;
            a.out = dadj + c * kc;
            a.out_ld = D.d_ld;
            {
                ProfScope ps(PROF_LP_GRAD_ADJ, st);
                rc = fl_dispatch<FLASH_DADJS, false>(ks, a, st);
            }
            if (rc) break;
            fl_common(a, D, FLASH_DNEGS, adjrec + c * adjset, negrec + c * negset, rg, desc->free_cus);
            a.S = S;
            a.s_tiled = (desc->flags & MARIUS_LP_STORE_SCORES) ? 0 : 1;
            a.d = kc;
            a.out = gocc + c * kc;
            a.out_ld = D.d_ld;
            a.negocc_off[0] = negocc_off[0];
            a.negocc_off[1] = negocc_off[1];
            ProfScope ps(PROF_LP_GRAD_NEG, st);
            rc = fl_dispatch<FLASH_DNEGS, false>(ks, a, st);
        }
        return rc;
    }
    (void)dadj;  // dAdj left the forward sweep as partials (flash_forward)
    fl_common(a, D, FLASH_DNEG, adjrec, negrec, rg, desc->free_cus);
    a.out = gocc;
    a.out_ld = D.d_ld;
    a.negocc_off[0] = negocc_off[0];
    a.negocc_off[1] = negocc_off[1];
    if (filtered && part) {
        const FlFilterDims f = fl_filter_dims(D);
        const char* fbase = (const char*)part + fl_stats_bytes(D);
        a.foff = (const uint32_t*)fbase + (f.nkeys_max + 1);
        a.fent = (const uint16_t*)(fbase + f.off_bytes) + f.ent_cap;
    }
    {
        ProfScope ps(PROF_LP_GRAD_NEG, st);
        rc = fl_dispatch<FLASH_DNEG, false>(ks, a, st);
    }
    return rc;
}

}  // namespace marius
