// TEST INFRASTRUCTURE: a shim of marius_amd/csrc/kernels/common.h for the CPU build of a kernel file (tests/emul/README in build_emul.py).
// the transformed copies of the kernel files (build_emul.py) include THIS common.h: HIP's execution model emulated on host
// threads — one workgroup at a time, one std::thread per work-item, __syncthreads = a barrier over the workgroup, __shfl_* = an exchange through
// a per-wave buffer between two wave barriers, __shared__ = static storage (one workgroup runs at a time).  Nothing under marius_amd/ includes
// this; the product is the hipcc build.
#pragma once
#include <stdint.h>

#include <atomic>
#include <barrier>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <tuple>
#include <type_traits>
#include <vector>

#include "marius_hip.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
struct alignas(8) float2 {
    float x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
struct alignas(16) int4 {
    int x, y, z, w;
};
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace emul {
inline thread_local dim3 t_threadIdx, t_blockIdx;
inline thread_local unsigned t_linear = 0;
inline dim3 g_blockDim, g_gridDim;
inline std::barrier<>* g_block_barrier = nullptr;
inline std::vector<std::unique_ptr<std::barrier<>>> g_wave_barriers;
inline uint64_t g_slots[64][64];  // [wave][lane]

// One set of host threads per LAUNCH (creating 256 threads per workgroup dominated the run time, ten-fold under ASan): the threads walk the grid
// together, workgroup by workgroup; between two workgroups they meet at `sync` twice — once so that everybody has left the previous workgroup,
// once after thread 0 has rebuilt the workgroup's barriers (a work-item that returns early DROPS out of them, so they are per workgroup).
inline unsigned char* g_dyn_smem = nullptr;  // the launch's dynamic LDS (one workgroup runs at a time)
template <typename F>
void launch(dim3 grid, dim3 block, F&& body, size_t dyn_smem_bytes = 0) {
    g_blockDim = block;
    g_gridDim = grid;
    std::unique_ptr<unsigned char[]> dyn(new unsigned char[dyn_smem_bytes + 64]);
    g_dyn_smem = (unsigned char*)(((uintptr_t)dyn.get() + 63) & ~(uintptr_t)63);
    const unsigned nt = block.x * block.y * block.z, nwaves = (nt + 63) / 64;  // work-items are numbered x fastest, as the hardware packs them into waves
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    if (nt == 0 || nblocks == 0) return;
    std::barrier<> sync((std::ptrdiff_t)nt);
    std::unique_ptr<std::barrier<>> block_barrier;
    std::vector<std::thread> th;
    th.reserve(nt);
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            t_linear = t;
            t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (uint64_t b = 0; b < nblocks; ++b) {
                sync.arrive_and_wait();
                if (t == 0) {
                    block_barrier.reset(new std::barrier<>((std::ptrdiff_t)nt));
                    g_block_barrier = block_barrier.get();
                    g_wave_barriers.clear();
                    for (unsigned w = 0; w < nwaves; ++w) g_wave_barriers.emplace_back(new std::barrier<>((std::ptrdiff_t)std::min(64u, nt - 64 * w)));
                }
                sync.arrive_and_wait();
                t_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y)));
                body();
                // a work-item that returned early must not leave the others of its workgroup / wave waiting for ever
                g_block_barrier->arrive_and_drop();
                g_wave_barriers[t >> 6]->arrive_and_drop();
            }
        });
    for (auto& x : th) x.join();
}
template <typename T>
T exchange(T v, int src_lane_of_me(int lane, int arg), int arg) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 64 bits");
    const int lane = t_linear & 63, wave = t_linear >> 6;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    g_slots[wave][lane] = bits;
    g_wave_barriers[wave]->arrive_and_wait();
    const int src = src_lane_of_me(lane, arg);
    uint64_t got = (src >= 0 && src < 64) ? g_slots[wave][src] : bits;
    g_wave_barriers[wave]->arrive_and_wait();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
}  // namespace emul

#define threadIdx emul::t_threadIdx
#define blockIdx emul::t_blockIdx
#define blockDim emul::g_blockDim
#define gridDim emul::g_gridDim
inline void __syncthreads() { emul::g_block_barrier->arrive_and_wait(); }
template <typename T>
T __shfl_xor(T v, int mask, int = 64) { return emul::exchange<T>(v, [](int lane, int m) { return lane ^ m; }, mask); }
template <typename T>
T __shfl_up(T v, int delta, int = 64) { return emul::exchange<T>(v, [](int lane, int d) { return lane >= d ? lane - d : lane; }, delta); }
template <typename T>
T __shfl(T v, int src, int = 64) { return emul::exchange<T>(v, [](int, int sl) { return sl & 63; }, src); }
inline unsigned long long __ballot(int pred) {
    const int lane = emul::t_linear & 63, wave = emul::t_linear >> 6;
    emul::g_slots[wave][lane] = pred ? 1ull : 0ull;
    emul::g_wave_barriers[wave]->arrive_and_wait();
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m |= (emul::g_slots[wave][l] & 1ull) << l;   // (lanes that do not exist in a short last wave never wrote: their slots are stale)
    emul::g_wave_barriers[wave]->arrive_and_wait();
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned __float_as_uint(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
template <typename T>
T atomicMax(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
template <typename T>
T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T>
T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred ? 1 : 0); }
inline void __builtin_amdgcn_wave_barrier() { emul::g_wave_barriers[emul::t_linear >> 6]->arrive_and_wait(); }
inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
template <typename T>
T __builtin_amdgcn_readfirstlane(T v) { return emul::exchange<T>(v, [](int, int) { return 0; }, 0); }   // (lane 0 of every wave that calls it is active in the emulated files)
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
inline unsigned long long wall_clock64() {   // 100 MHz, like the device's constant clock
    return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, order)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
using std::max;
using std::min;
template <typename T>
T __shfl_down(T v, int delta, int = 64) { return emul::exchange<T>(v, [](int lane, int d) { return lane + d < 64 ? lane + d : lane; }, delta); }

// the few runtime calls kernel files make outside launches
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return hipSuccess;
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulation"; }

namespace marius {
inline thread_local char g_last_error[512] = "";
inline void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}
inline int check_launch(const char*) { return MARIUS_OK; }
// the switches the emulated files read (the hipcc build: error.hip)
struct KernelEnv {
    int mt_threads;
    bool seg_fused_fixup_off, seg_group_off;
    bool sort_rocprim, maps_fused;
    int pm_nwg;
};
inline KernelEnv read_env() {
    auto first = [](const char* name) -> char { const char* e = getenv(name); return e ? e[0] : (char)0; };
    const char* e = getenv("MARIUS_MT_THREADS");
    const char* w = getenv("MARIUS_PM_NWG");
    return KernelEnv{e ? atoi(e) : 0, first("MARIUS_SEG_FUSED_FIXUP") == '0', first("MARIUS_SEG_GROUP") == '0', first("MARIUS_SORT") == 'r', first("MARIUS_MAPS") == 'f', w ? atoi(w) : 0};
}
inline KernelEnv g_env = read_env();
inline const KernelEnv& kernel_env() { return g_env; }
enum ProfId { PROF_LP_SCORES = 0, PROF_LP_GRAD_ADJ, PROF_LP_GRAD_NEG, PROF_LP_PREP, PROF_LP_LSE, PROF_LP_EDGE_BWD, PROF_GATHER, PROF_SEG_ADAGRAD, PROF_SORT_UNIQUE, PROF_MT_FILL,
              PROF_LP_PACK, PROF_COUNT };
struct ProfScope {
    ProfScope(int, hipStream_t) {}
};
constexpr int WAVE = 64;
inline float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
inline float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
inline int row_vec_width(const void* base, int64_t ld, int d) {
    uintptr_t p = reinterpret_cast<uintptr_t>(base);
    if ((d % 4 == 0) && (ld % 4 == 0) && (p % 16 == 0)) return 4;
    if ((d % 2 == 0) && (ld % 2 == 0) && (p % 8 == 0)) return 2;
    return 1;
}
#define MARIUS_REQUIRE(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            marius::set_last_error(__VA_ARGS__); \
            return MARIUS_ERR_INVALID;           \
        }                                        \
    } while (0)
inline hipStream_t as_stream(marius_stream_t s) { return (hipStream_t)s; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
}  // namespace marius
