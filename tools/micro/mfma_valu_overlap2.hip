// Follow-up: VALU throughput (independent chains) next to FP32 vs BF16 MFMAs on the same SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
// KIND 0: f32 32x32x2 (64 cyc), KIND 1: bf16 32x32x16 (nominally 32 cyc... measured)
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int nvalu) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float r = 0.f;
    if (do_mfma) {
        v16f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        v8bf bx, by;
        for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)x; by[i] = (__bf16)y; }
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    if (do_valu) {
        if (mode == 2 && (iters & 1)) __builtin_amdgcn_s_setprio(3);
        float z[16];
        for (int j = 0; j < 16; ++j) z[j] = x + j;
        for (int i = 0; i < iters; ++i) {
            for (int j = 0; j < nvalu; j += 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z[c]) : "v"(y));
            }
        }
        for (int j = 0; j < 16; ++j) r += z[j];
    }
    if (r == 123.456f) out[0] = r;
}
template <int KIND>
float run(int mode, int iters, int nvalu, int threads) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<256, threads>>>(out, iters, mode, nvalu);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<256, threads>>>(out, iters, mode, nvalu);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(out); return ms;
}
int main() {
    const int iters = 20000;
    for (int odd = 0; odd < 2; ++odd) {
    if (odd) printf("==== VALU wave at s_setprio 3 ====\n");
    for (int kind = 0; kind < 2; ++kind) {
        auto R = [&](int mode, int nv, int th) { return kind == 0 ? run<0>(mode, iters + odd, nv, th) : run<1>(mode, iters + odd, nv, th); };
        printf("kind %s, 4 MFMAs per iteration, %d iterations\n", kind == 0 ? "f32 32x32x2" : "bf16 32x32x16", iters);
        printf("  MFMA only 1 wave/SIMD: %.3f ms   2 waves/SIMD: %.3f ms\n", R(0, 0, 256), R(0, 0, 512));
        for (int nv : {16, 32, 64, 128}) {
            printf("  nvalu=%3d independent v_fma per iteration: VALU only (1 wave/SIMD) %.3f | VALU only (2 waves/SIMD) %.3f | MFMA wave + VALU wave %.3f\n", nv, R(1, nv, 256), R(1, nv, 512), R(2, nv, 512));
        }
    }
    }
    return 0;
}
