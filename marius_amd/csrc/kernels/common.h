// Shared helpers for the gfx950 kernels of libmarius_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "marius_hip.h"

namespace marius {

void set_last_error(const char* fmt, ...);
// MARIUS_SYNC_LAUNCH=1 (debugging): every C-ABI launch is named on stderr and followed by a device-wide synchronisation, so that a kernel
// that never returns (or faults) is the last name printed.  Off: a single predictable branch.
int launch_debug(const char* what);

// Every MARIUS_* switch the kernel library honours, read ONCE when the library is first used (no getenv in per-call code).  They select
// code that a test or an A/B run of this tree uses; a production run sets none of them.  marius_config_reload() re-reads them (tests switch a path
// inside one process; tools/ab_env.sh runs set them before the process starts and never need it).
struct KernelEnv {
    char scores;          // MARIUS_SCORES=res: the LDS-tile score kernel where the register-fragment one would apply
    bool no_fast;         // MARIUS_NO_FAST=1 / MARIUS_KERNELS=generic
    char kernels;         // MARIUS_KERNELS: 'g'eneric, 'f'ast, else the MFMA-tuned level
    bool no_vlog;         // MARIUS_NO_VLOG=1
    bool timeline_grads;  // MARIUS_TIMELINE_GRADS: cycle stamps of the backward kernel instead of the score kernel
    int flash_wide;       // MARIUS_FLASH_WIDE: -1 unset, 0 off, 128 = round-3 chunk width
    char flash;           // MARIUS_FLASH: '0' off, 'f' forced, 0 unset
    bool has_flash_reserve, has_flash_nwg;
    int flash_reserve, flash_nwg;  // MARIUS_FLASH_RESERVE / MARIUS_FLASH_NWG
    bool flash_f16_off;   // MARIUS_FLASH_F16=0
    bool flash_rotate_off;  // MARIUS_FLASH_ROTATE=0
    bool flash_tail4_off;   // MARIUS_FLASH_TAIL4=0: d = 36 / 68 / 100 keep a k-step of their own for the last four columns (round-4 record layout)
    bool seg_fused_fixup_off, seg_group_off;  // MARIUS_SEG_FUSED_FIXUP=0, MARIUS_SEG_GROUP=0
    bool sort_rocprim;    // MARIUS_SORT=rocprim
    bool maps_fused;      // MARIUS_MAPS=fused: the DataLoader prepares a batch's maps with the ONE persistent launch (marius_prepare_maps) instead of the separate
                          // launches — measured slower inside the pipeline, so it is opt-in (marius_prepare_maps_preferred)
    int pm_nwg;           // MARIUS_PM_NWG: workgroups of the fused map launch (0 = default) — A/B runs, the single-workgroup test
    int mt_threads;       // MARIUS_MT_THREADS: workgroup size of the MT19937 fill (64 / 128 / 256; 0 = default) — A/B runs
    int sync_launch;      // MARIUS_SYNC_LAUNCH
};
const KernelEnv& kernel_env();

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error("%s: %s", what, hipGetErrorString(e));
        return MARIUS_ERR_HIP;
    }
    return launch_debug(what);
}

#define MARIUS_REQUIRE(cond, ...)             \
    do {                                      \
        if (!(cond)) {                        \
            marius::set_last_error(__VA_ARGS__); \
            return MARIUS_ERR_INVALID;        \
        }                                     \
    } while (0)

inline hipStream_t as_stream(marius_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int WAVE = 64;

// optional HIP-event profiler (see error.hip): PROF_SCOPE(id, stream) { launch; }
enum ProfId { PROF_LP_SCORES = 0, PROF_LP_GRAD_ADJ, PROF_LP_GRAD_NEG, PROF_LP_PREP, PROF_LP_LSE, PROF_LP_EDGE_BWD, PROF_GATHER,
              PROF_SEG_ADAGRAD, PROF_SORT_UNIQUE, PROF_MT_FILL, PROF_LP_PACK, PROF_COUNT };
struct ProfMark {
    int id;
    hipEvent_t a, b;
};
bool prof_enabled();
void prof_begin(int id, hipStream_t st, ProfMark& m);
void prof_end(hipStream_t st, ProfMark& m);
struct ProfScope {
    ProfMark m;
    hipStream_t st;
    ProfScope(int id, hipStream_t s) : st(s) { prof_begin(id, s, m); }
    ~ProfScope() { prof_end(st, m); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// vector width usable for rows of `d` floats starting at multiples of `ld` floats from a 16-B aligned base
inline int row_vec_width(const void* base, int64_t ld, int d) {
    uintptr_t p = reinterpret_cast<uintptr_t>(base);
    if ((d % 4 == 0) && (ld % 4 == 0) && (p % 16 == 0)) return 4;
    if ((d % 2 == 0) && (ld % 2 == 0) && (p % 8 == 0)) return 2;
    return 1;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace marius
