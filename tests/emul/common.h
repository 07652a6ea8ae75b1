// TEST INFRASTRUCTURE: a shim of marius_amd/csrc/kernels/common.h for the CPU build of a kernel file (tests/emul/README in build_emul.py).
// `g++ -I tests/emul -I include -x c++ marius_amd/csrc/kernels/neighbor.hip` finds THIS common.h first: HIP's execution model emulated on host
// threads — one workgroup at a time, one std::thread per work-item, __syncthreads = a barrier over the workgroup, __shfl_* = an exchange through
// a per-wave buffer between two wave barriers, __shared__ = static storage (one workgroup runs at a time).  Nothing under marius_amd/ includes
// this; the product is the hipcc build.
#pragma once
#include <stdint.h>

#include <barrier>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
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

namespace emul {
inline thread_local dim3 t_threadIdx, t_blockIdx;
inline dim3 g_blockDim, g_gridDim;
inline std::barrier<>* g_block_barrier = nullptr;
inline std::vector<std::unique_ptr<std::barrier<>>> g_wave_barriers;
inline uint64_t g_slots[64][64];  // [wave][lane]

template <typename F>
void launch(dim3 grid, dim3 block, F&& body) {
    g_blockDim = block;
    g_gridDim = grid;
    const unsigned nt = block.x, nwaves = (nt + 63) / 64;
    for (unsigned b = 0; b < grid.x; ++b) {
        std::barrier<> bb((std::ptrdiff_t)nt);
        g_block_barrier = &bb;
        g_wave_barriers.clear();
        for (unsigned w = 0; w < nwaves; ++w) g_wave_barriers.emplace_back(new std::barrier<>((std::ptrdiff_t)std::min(64u, nt - 64 * w)));
        std::vector<std::thread> th;
        th.reserve(nt);
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t, b] {
                t_threadIdx = dim3(t);
                t_blockIdx = dim3(b);
                body();
                // a work-item that returned early must not leave the others of its workgroup / wave waiting for ever
                g_block_barrier->arrive_and_drop();
                g_wave_barriers[t >> 6]->arrive_and_drop();
            });
        for (auto& x : th) x.join();
    }
}
template <typename T>
T exchange(T v, int src_lane_of_me(int lane, int arg), int arg) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 64 bits");
    const int lane = t_threadIdx.x & 63, wave = t_threadIdx.x >> 6;
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
T __shfl_down(T v, int delta, int = 64) { return emul::exchange<T>(v, [](int lane, int d) { return lane + d < 64 ? lane + d : lane; }, delta); }

#define MARIUS_LAUNCH(kernel, grid, block, stream, ...) emul::launch(dim3(grid), dim3(block), [&] { kernel(__VA_ARGS__); })

namespace marius {
inline void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}
inline int check_launch(const char*) { return MARIUS_OK; }
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
