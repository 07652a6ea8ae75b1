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
