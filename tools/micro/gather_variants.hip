// Gather of random 400-byte rows out of a 34 GB table (Freebase86m shape): which launch geometry / access pattern gets closest to HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_variants gather_variants.hip ; run on an MI355X (optionally: rows d).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);        \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// V0: the shipped geometry: block (32, 8), 4 rows in flight per thread row, lane c handles 16-B piece c of its row (25 of 32 lanes busy)
template <int UNROLL>
__global__ __launch_bounds__(256) void v0(const float* __restrict__ t, int64_t ld, const int64_t* __restrict__ ids, int64_t n, int vpr, float* __restrict__ o) {
    const int tx = threadIdx.x, ty = threadIdx.y, TX = blockDim.x, TY = blockDim.y;
    int64_t rows[UNROLL], src[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        rows[k] = ((int64_t)blockIdx.x * UNROLL + k) * TY + ty;
        src[k] = rows[k] < n ? ids[rows[k]] : -1;
    }
    for (int c = tx; c < vpr; c += TX) {
        f4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
            if (src[k] >= 0) v[k] = reinterpret_cast<const f4*>(t + src[k] * ld)[c];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
            if (src[k] >= 0) reinterpret_cast<f4*>(o + rows[k] * ld)[c] = v[k];
    }
}

// V1: flat pieces: piece p = row * vpr + c; every lane busy; UNROLL pieces in flight per thread, consecutive lanes = consecutive pieces
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void v1(const float* __restrict__ t, int64_t ld, const int64_t* __restrict__ ids, int64_t n, int vpr, float* __restrict__ o) {
    const int64_t total = n * vpr;
    const int64_t base = ((int64_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    f4 v[UNROLL];
    int64_t dst[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        const int64_t p = base + (int64_t)k * 256;
        dst[k] = -1;
        if (p < total) {
            const int64_t r = p / vpr;
            const int c = (int)(p - r * vpr);
            const f4* s = reinterpret_cast<const f4*>(t + ids[r] * ld) + c;
            v[k] = NT ? __builtin_nontemporal_load(s) : *s;
            dst[k] = r * (ld / 4) + c;
        }
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k)
        if (dst[k] >= 0) reinterpret_cast<f4*>(o)[dst[k]] = v[k];
}

// V2: one wave per pair of rows at a time (lanes 0..24 row A, 32..56 row B), ROWS pairs in flight
template <int PAIRS>
__global__ __launch_bounds__(256) void v2(const float* __restrict__ t, int64_t ld, const int64_t* __restrict__ ids, int64_t n, int vpr, float* __restrict__ o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, c = lane & 31;
    const int64_t r0 = (((int64_t)blockIdx.x * 4 + wave) * PAIRS) * 2 + half;
    f4 v[PAIRS];
    int64_t src[PAIRS];
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int64_t r = r0 + 2 * k;
        src[k] = (r < n && c < vpr) ? ids[r] : -1;
    }
#pragma unroll
    for (int k = 0; k < PAIRS; ++k)
        if (src[k] >= 0) v[k] = reinterpret_cast<const f4*>(t + src[k] * ld)[c];
#pragma unroll
    for (int k = 0; k < PAIRS; ++k)
        if (src[k] >= 0) reinterpret_cast<f4*>(o + (r0 + 2 * k) * ld)[c] = v[k];
}

__global__ void copy_k(const f4* __restrict__ s, f4* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

__global__ void fill_k(float* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 1e-3f;
}

template <class F>
static float timeit(F f, int iters = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;  // us
}

int main(int argc, char** argv) {
    const int64_t nodes = argc > 1 ? atoll(argv[1]) : 86054151;
    const int d = argc > 2 ? atoi(argv[2]) : 100;
    const int64_t n = 200000;
    const int vpr = d / 4;
    float *table, *out;
    int64_t* ids;
    CK(hipMalloc(&table, (size_t)nodes * d * 4));
    CK(hipMalloc(&out, (size_t)n * d * 4));
    fill_k<<<4096, 256>>>(table, nodes * d);
    std::mt19937_64 rng(1);
    // a fresh id set per launch: repeating one set would serve the rows from the 256 MB Infinity Cache after the first pass
    constexpr int SETS = 32;
    int64_t* idsets;
    CK(hipMalloc(&idsets, (size_t)SETS * n * 8));
    std::vector<int64_t> h(n);
    for (int s = 0; s < SETS; ++s) {
        for (auto& x : h) x = (int64_t)(rng() % (uint64_t)nodes);
        std::sort(h.begin(), h.end());
        CK(hipMemcpy(idsets + (size_t)s * n, h.data(), n * 8, hipMemcpyHostToDevice));
    }
    int cursor = 0;
    auto next_ids = [&]() { ids = idsets + (size_t)(cursor++ % SETS) * n; };
    next_ids();
    CK(hipDeviceSynchronize());
    const double mb = (double)n * d * 4 / 1e6;
    auto report = [&](const char* name, float us) { printf("%-44s %7.1f us   read %.2f TB/s   read+write %.2f TB/s\n", name, us, mb / us, 2 * mb / us); };
    report("contiguous copy of the same bytes (warm)", timeit([&] { copy_k<<<2048, 256>>>((const f4*)table, (f4*)out, n * vpr); }));
    int64_t coff = 0;
    report("contiguous copy, a fresh 80 MB window each time", timeit([&] {
               coff = (coff + 4 * n * vpr) % ((nodes - 2 * n) * vpr);
               copy_k<<<2048, 256>>>((const f4*)table + coff, (f4*)out, n * vpr);
           }));
    report("v0 shipped: (32,8) block, 4 rows/thread-row", timeit([&] { next_ids(); v0<4><<<dim3((unsigned)((n + 31) / 32)), dim3(32, 8)>>>(table, d, ids, n, vpr, out); }));
    report("v0 with 8 rows in flight", timeit([&] { next_ids(); v0<8><<<dim3((unsigned)((n + 63) / 64)), dim3(32, 8)>>>(table, d, ids, n, vpr, out); }));
    report("v0 with 2 rows in flight", timeit([&] { next_ids(); v0<2><<<dim3((unsigned)((n + 15) / 16)), dim3(32, 8)>>>(table, d, ids, n, vpr, out); }));
    const int64_t pieces = n * vpr;
    report("v1 flat pieces, 4 in flight", timeit([&] { next_ids(); v1<4, false><<<dim3((unsigned)((pieces + 1023) / 1024)), 256>>>(table, d, ids, n, vpr, out); }));
    report("v1 flat pieces, 8 in flight", timeit([&] { next_ids(); v1<8, false><<<dim3((unsigned)((pieces + 2047) / 2048)), 256>>>(table, d, ids, n, vpr, out); }));
    report("v1 flat pieces, 2 in flight", timeit([&] { next_ids(); v1<2, false><<<dim3((unsigned)((pieces + 511) / 512)), 256>>>(table, d, ids, n, vpr, out); }));
    report("v1 flat pieces, 4 in flight, nontemporal", timeit([&] { next_ids(); v1<4, true><<<dim3((unsigned)((pieces + 1023) / 1024)), 256>>>(table, d, ids, n, vpr, out); }));
    report("v2 wave = 2 rows, 4 pairs in flight", timeit([&] { next_ids(); v2<4><<<dim3((unsigned)((n + 31) / 32)), 256>>>(table, d, ids, n, vpr, out); }));
    report("v2 wave = 2 rows, 8 pairs in flight", timeit([&] { next_ids(); v2<8><<<dim3((unsigned)((n + 63) / 64)), 256>>>(table, d, ids, n, vpr, out); }));
    // how much is the random access itself: same kernels on a small table (rows hit L2 / MALL)
    std::vector<int64_t> hs(n);
    for (auto& x : hs) x = (int64_t)(rng() % 200000ull);
    std::sort(hs.begin(), hs.end());
    int64_t* small;
    CK(hipMalloc(&small, n * 8));
    CK(hipMemcpy(small, hs.data(), n * 8, hipMemcpyHostToDevice));
    report("v0 shipped, ids within the first 200k rows", timeit([&] { v0<4><<<dim3((unsigned)((n + 31) / 32)), dim3(32, 8)>>>(table, d, small, n, vpr, out); }));
    report("v1 flat 4, ids within the first 200k rows", timeit([&] { v1<4, false><<<dim3((unsigned)((pieces + 1023) / 1024)), 256>>>(table, d, small, n, vpr, out); }));
    return 0;
}
