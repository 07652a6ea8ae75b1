// Does VALU work of one wave overlap with FP32 MFMAs of another wave on the same SIMD?  (gfx950 microbenchmark)
// mode 0: every wave runs MFMAs; mode 1: every wave runs VALU; mode 2: even waves MFMA, odd waves VALU (two waves per SIMD with 512 threads)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int nvalu) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave < 4)) || (mode == 3);
    const bool do_valu = mode == 1 || (mode == 2 && (wave >= 4)) || (mode == 3);
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float r = 0.f;
    if (do_mfma) {
        if (KIND == 0) {
            v16f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
                if (mode == 3) {
                    for (int j = 0; j < nvalu; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
                }
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        } else {
            v4f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0}, a4 = {0}, a5 = {0}, a6 = {0}, a7 = {0};
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
                a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a5, 0, 0, 0);
                a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a6, 0, 0, 0);
                a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a7, 0, 0, 0);
                if (mode == 3) {
                    for (int j = 0; j < nvalu; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
                }
            }
            r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[0] + a6[0] + a7[0];
        }
    }
    if (do_valu && mode != 3) {
        float z0 = x, z1 = x + 1, z2 = x + 2, z3 = x + 3;
        for (int i = 0; i < iters; ++i) {
            for (int j = 0; j < nvalu; j += 4) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z0) : "v"(y));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z1) : "v"(y));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z2) : "v"(y));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z3) : "v"(y));
            }
        }
        r += z0 + z1 + z2 + z3;
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
    for (int kind = 0; kind < 2; ++kind) {
        auto R = [&](int mode, int nv, int th) { return kind == 0 ? run<0>(mode, iters, nv, th) : run<1>(mode, iters, nv, th); };
        printf("kind %s (4 x 64-cycle or 8 x 32-cycle MFMAs = 256 pipe cycles per iteration)\n", kind == 0 ? "32x32x2" : "16x16x4");
        printf("  1 wave/SIMD  MFMA only           : %.3f ms\n", R(0, 0, 256));
        printf("  2 waves/SIMD MFMA only           : %.3f ms\n", R(0, 0, 512));
        for (int nv : {16, 32, 64, 128}) {
            printf("  nvalu=%3d: VALU only 1w %.3f | MFMA wave + VALU wave per SIMD %.3f | same wave interleaved (1w) %.3f\n", nv, R(1, nv, 256), R(2, nv, 512), R(3, nv, 256));
        }
    }
    return 0;
}
