// TEST INFRASTRUCTURE: a shim of marius_amd/csrc/kernels/common.h for the CPU build of a kernel file (tests/emul/README in build_emul.py).
// the transformed copies of the kernel files (build_emul.py) include THIS common.h: HIP's execution model emulated on the host —
// one workgroup at a time, one fiber per work-item, __syncthreads = a barrier over the workgroup's live work-items, __shfl_* / ballot = an exchange
// through a per-wave buffer between two wave barriers, __shared__ = static storage (one workgroup runs at a time), atomics = the compiler's.
// Nothing under marius_amd/ includes this; the product is the hipcc build.
#pragma once
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <sys/mman.h>
#include <ucontext.h>
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
// fp16 / bf16 scalars g++ 11 does not have on x86-64: conversion from float rounds to nearest even (v_cvt_f16_f32 / v_cvt_pk_bf16_f32), back is exact
struct _Float16 {
    uint16_t b;
    _Float16() = default;
    _Float16(float f) {
        uint32_t x;
        memcpy(&x, &f, 4);
        const uint32_t sign = (x >> 16) & 0x8000u;
        const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
        uint32_t m = x & 0x7fffffu;
        if (((x >> 23) & 0xff) == 0xff) {
            b = (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
        } else if (e >= 31) {
            b = (uint16_t)(sign | 0x7c00u);
        } else if (e <= 0) {
            if (e < -10) {
                b = (uint16_t)sign;
            } else {
                m |= 0x800000u;
                const int shift = 14 - e;  // 13 + (1 - e)
                uint32_t h = m >> shift;
                const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
                if (rem > half || (rem == half && (h & 1))) ++h;
                b = (uint16_t)(sign | h);
            }
        } else {
            uint32_t h = ((uint32_t)e << 10) | (m >> 13);
            const uint32_t rem = m & 0x1fffu;
            if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
            b = (uint16_t)(sign | h);
        }
    }
    operator float() const {
        const uint32_t sign = (uint32_t)(b & 0x8000u) << 16, e = (b >> 10) & 0x1f, m = b & 0x3ffu;
        uint32_t x;
        if (e == 0) {
            if (m == 0) {
                x = sign;
            } else {
                int k = 0;
                uint32_t mm = m;
                while (!(mm & 0x400u)) {
                    mm <<= 1;
                    ++k;
                }
                x = sign | ((uint32_t)(127 - 15 + 1 - k) << 23) | ((mm & 0x3ffu) << 13);
            }
        } else if (e == 31) {
            x = sign | 0x7f800000u | (m << 13);
        } else {
            x = sign | ((e + 127 - 15) << 23) | (m << 13);
        }
        float f;
        memcpy(&f, &x, 4);
        return f;
    }
};
struct __bf16 {
    uint16_t b;
    __bf16() = default;
    __bf16(float f) {
        uint32_t x;
        memcpy(&x, &f, 4);
        if ((x & 0x7fffffffu) > 0x7f800000u) {
            b = (uint16_t)((x >> 16) | 0x40u);
        } else {
            x += 0x7fffu + ((x >> 16) & 1u);
            b = (uint16_t)(x >> 16);
        }
    }
    operator float() const {
        const uint32_t x = (uint32_t)b << 16;
        float f;
        memcpy(&f, &x, 4);
        return f;
    }
};
namespace emul {
template <typename T, int N>
struct vec {  // clang's ext_vector_type for the emulation build: element access, brace initialisation, elementwise conversion
    T v[N];
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
};
template <typename To, typename S, int N>
To convert(const vec<S, N>& a) {
    To r;
    for (int i = 0; i < N; ++i) r[i] = static_cast<std::remove_reference_t<decltype(r[0])>>(a[i]);
    return r;
}
}  // namespace emul
#define __builtin_convertvector(v, T) emul::convert<T>(v)
struct alignas(16) int4 {
    int x, y, z, w;
};
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace emul {
// ---- one workgroup at a time, its work-items as FIBERS of the calling thread (ucontext): a work-item runs until it reaches a barrier, a wave
// collective or its end, then the scheduler runs the next one.  (The first version gave every work-item a std::thread: a barrier of 512 kernel
// threads costs milliseconds, ten times that under ASan; a fiber switch costs well under a microsecond, so the parity tests run at their own shapes.)
// A barrier is released when every work-item of the workgroup (wave) that has not finished is waiting at it — a work-item that returned early
// drops out, as the hardware's barrier counts only live waves.  If nothing can run and not everything has finished, the launch ABORTS with the
// state of every work-item: a barrier under divergent control flow shows up as a message, not as a hang.
struct Fiber {
    ucontext_t ctx;
    int state;  // 0 runnable, 1 waiting at the workgroup barrier, 2 waiting at its wave's barrier, 3 finished
};
inline dim3 g_blockDim, g_gridDim, g_threadIdx, g_blockIdx;
inline unsigned t_linear = 0;  // the running work-item (x fastest, as the hardware packs work-items into waves)
inline std::vector<Fiber> g_fibers;
inline ucontext_t g_sched;
inline std::function<void()>* g_body = nullptr;
inline unsigned g_nt = 0;
inline uint64_t g_slots[64][64];  // [wave][lane]
inline unsigned char* g_dyn_smem = nullptr;  // the launch's dynamic LDS
// (ASan's swapcontext interceptor clears the shadow of the WHOLE target stack on every switch: the stack size is the cost of a switch there)
#if defined(__SANITIZE_ADDRESS__)
constexpr size_t FIBER_STACK = 96 * 1024;
#else
constexpr size_t FIBER_STACK = 256 * 1024;
#endif
#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
inline const void* g_sched_stack_bottom = nullptr;
inline size_t g_sched_stack_size = 0;
inline std::vector<void*> g_fake(1024, nullptr);
inline void* g_sched_fake = nullptr;
#endif
inline unsigned char* g_stacks = nullptr;
inline size_t g_stacks_for = 0;

inline void to_scheduler() {  // called on a fiber
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_start_switch_fiber(&g_fake[t_linear], g_sched_stack_bottom, g_sched_stack_size);
#endif
    const unsigned me = t_linear;
    swapcontext(&g_fibers[me].ctx, &g_sched);
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(g_fake[me], nullptr, nullptr);
#endif
}
inline void trampoline() {
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack_bottom, &g_sched_stack_size);
#endif
    (*g_body)();
    g_fibers[t_linear].state = 3;
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_start_switch_fiber(nullptr, g_sched_stack_bottom, g_sched_stack_size);  // (nullptr: this fiber's fake stack is destroyed)
#endif
    // returning resumes uc_link = the scheduler
}
inline void wait_at(int what) {
    g_fibers[t_linear].state = what;
    to_scheduler();
}

template <typename F>
void launch(dim3 grid, dim3 block, F&& body, size_t dyn_smem_bytes = 0) {
    g_blockDim = block;
    g_gridDim = grid;
    const unsigned nt = block.x * block.y * block.z, nwaves = (nt + 63) / 64;
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    if (nt == 0 || nblocks == 0) return;
    std::unique_ptr<unsigned char[]> dyn(new unsigned char[dyn_smem_bytes + 64]);
    g_dyn_smem = (unsigned char*)(((uintptr_t)dyn.get() + 63) & ~(uintptr_t)63);
    std::function<void()> fn = [&] { body(); };
    g_body = &fn;
    g_nt = nt;
    g_fibers.assign(nt, Fiber{});
    // the fibers' stacks: mapped once for the largest workgroup seen and kept (under ASan a fresh 256 MB allocation per launch was most of the run time)
    if (nt > g_stacks_for) {  // (a namespace-scope counter: a static local here would be one per call site — launch is a template over the kernel's lambda)
        if (g_stacks) munmap(g_stacks, g_stacks_for * FIBER_STACK);
        g_stacks = (unsigned char*)mmap(nullptr, (size_t)nt * FIBER_STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (unsigned char*)MAP_FAILED) abort();
        g_stacks_for = nt;
    }
    for (uint64_t b = 0; b < nblocks; ++b) {
        g_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y)));
        for (unsigned t = 0; t < nt; ++t) {
            Fiber& f = g_fibers[t];
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * FIBER_STACK;
            f.ctx.uc_stack.ss_size = FIBER_STACK;
            f.ctx.uc_link = &g_sched;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
            f.state = 0;
        }
        unsigned done = 0;
        while (done < nt) {
            bool ran = false;
            for (unsigned t = 0; t < nt; ++t) {
                if (g_fibers[t].state != 0) continue;
                ran = true;
                t_linear = t;
                g_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
#if defined(__SANITIZE_ADDRESS__)
                __sanitizer_start_switch_fiber(&g_sched_fake, g_stacks + (size_t)t * FIBER_STACK, FIBER_STACK);
#endif
                swapcontext(&g_sched, &g_fibers[t].ctx);
#if defined(__SANITIZE_ADDRESS__)
                __sanitizer_finish_switch_fiber(g_sched_fake, nullptr, nullptr);
#endif
                if (g_fibers[t].state == 3) ++done;
            }
            // release the barriers everybody alive has reached
            unsigned at_block = 0, alive = 0;
            for (unsigned t = 0; t < nt; ++t) {
                alive += g_fibers[t].state != 3;
                at_block += g_fibers[t].state == 1;
            }
            bool released = false;
            if (alive && at_block == alive) {
                for (unsigned t = 0; t < nt; ++t)
                    if (g_fibers[t].state == 1) g_fibers[t].state = 0;
                released = true;
            }
            for (unsigned w = 0; w < nwaves; ++w) {
                unsigned wa = 0, ww = 0;
                for (unsigned t = 64 * w; t < nt && t < 64 * (w + 1); ++t) {
                    wa += g_fibers[t].state != 3;
                    ww += g_fibers[t].state == 2;
                }
                if (wa && ww == wa) {
                    for (unsigned t = 64 * w; t < nt && t < 64 * (w + 1); ++t)
                        if (g_fibers[t].state == 2) g_fibers[t].state = 0;
                    released = true;
                }
            }
            if (!ran && !released && done < nt) {
                fprintf(stderr, "emulated launch is stuck (workgroup %u %u %u): work-items waiting at the workgroup barrier / a wave collective / finished:", g_blockIdx.x, g_blockIdx.y,
                        g_blockIdx.z);
                for (unsigned t = 0; t < nt; ++t) fprintf(stderr, " %u:%d", t, g_fibers[t].state);
                fprintf(stderr, "\n");
                abort();
            }
        }
    }
    g_body = nullptr;
}
inline void block_barrier() { wait_at(1); }
inline void wave_barrier() { wait_at(2); }
template <typename T>
T exchange(T v, int src_lane_of_me(int lane, int arg), int arg) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 64 bits");
    const int lane = t_linear & 63, wave = t_linear >> 6;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    g_slots[wave][lane] = bits;
    wave_barrier();
    const int src = src_lane_of_me(lane, arg);
    uint64_t got = (src >= 0 && src < 64) ? g_slots[wave][src] : bits;
    wave_barrier();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
}  // namespace emul

#define threadIdx emul::g_threadIdx
#define blockIdx emul::g_blockIdx
#define blockDim emul::g_blockDim
#define gridDim emul::g_gridDim
inline void __syncthreads() { emul::block_barrier(); }
template <typename T>
T __shfl_xor(T v, int mask, int = 64) { return emul::exchange<T>(v, [](int lane, int m) { return lane ^ m; }, mask); }
template <typename T>
T __shfl_up(T v, int delta, int = 64) { return emul::exchange<T>(v, [](int lane, int d) { return lane >= d ? lane - d : lane; }, delta); }
template <typename T>
T __shfl(T v, int src, int = 64) { return emul::exchange<T>(v, [](int, int sl) { return sl & 63; }, src); }
inline unsigned long long __ballot(int pred) {
    const int lane = emul::t_linear & 63, wave = emul::t_linear >> 6;
    emul::g_slots[wave][lane] = pred ? 1ull : 0ull;
    emul::wave_barrier();
    unsigned long long m = 0;
    const int lanes = (int)std::min(64u, emul::g_nt - 64u * (unsigned)wave);
    for (int l = 0; l < lanes; ++l) m |= (emul::g_slots[wave][l] & 1ull) << l;
    emul::wave_barrier();
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
// v_mfma_f32_32x32x2_f32 as a wave collective (lp_common.h: lane l supplies A[m = l & 31][k = l >> 5] and B[k = l >> 5][n = l & 31] and holds
// D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31] in register r): every lane publishes (a, b), then computes its 16 outputs in fp32, k = 0 then k = 1
inline emul::vec<float, 16> __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emul::vec<float, 16> c, int, int, int) {
    const int lane = emul::t_linear & 63, wave = emul::t_linear >> 6;
    uint64_t bits;
    const float ab[2] = {a, b};
    memcpy(&bits, ab, 8);
    emul::g_slots[wave][lane] = bits;
    emul::wave_barrier();
    const int n = lane & 31, h = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float am[2], bn[2];
            memcpy(am, &emul::g_slots[wave][m + 32 * k], 8);
            memcpy(bn, &emul::g_slots[wave][n + 32 * k], 8);
            acc = acc + am[0] * bn[1];
        }
        c[r] = acc;
    }
    emul::wave_barrier();
    return c;
}
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred ? 1 : 0); }
inline void __builtin_amdgcn_wave_barrier() { emul::wave_barrier(); }
inline void __builtin_amdgcn_s_sleep(int) {}  // (workgroups run one after another: what a spin waits for has already happened)
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
    // lp_decoder.hip: the CPU build is the generic level (the tuned levels live in lp_fast / lp_res / lp_flash .hip, which are not part of it)
    char scores = 0, kernels = 'g';
    bool no_fast = true, no_vlog = false, timeline_grads = false, flash_f16_off = false;
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
