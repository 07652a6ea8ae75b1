// v_mfma_f32_32x32x16_bf16 throughput with the accumulator in ArchVGPRs vs AccVGPRs, 1 / 2 / 4 independent accumulators (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
template <int NACC, bool AGPR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v8bf bx, by;
    for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)(threadIdx.x * 1e-3f); by[i] = (__bf16)1.0001f; }
    v16f a[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) a[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4 / NACC; ++rep)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a[j]) : "v"(bx), "v"(by));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(bx), "v"(by));
            }
    }
    float r = 0.f;
    for (int j = 0; j < 4; ++j) r += a[j][0];
    if (r == 123.456f) out[0] = r;
}
template <int NACC, bool AGPR>
float run(int iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC, AGPR><<<256, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<NACC, AGPR><<<256, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(out); return ms;
}
int main() {
    const int iters = 20000;  // 4 MFMAs per iteration, one wave per SIMD
    printf("4 x v_mfma_f32_32x32x16_bf16 per iteration, %d iterations, ms (32 cycles each at 2.2 GHz = 1.16 ms)\n", iters);
    printf("  ArchVGPR accumulators: 1 acc %.3f | 2 acc %.3f | 4 acc %.3f\n", run<1, false>(iters), run<2, false>(iters), run<4, false>(iters));
    printf("  AccVGPR  accumulators: 1 acc %.3f | 2 acc %.3f | 4 acc %.3f\n", run<1, true>(iters), run<2, true>(iters), run<4, true>(iters));
    return 0;
}
