// Does a K = 8 fp16 MFMA cost half a K = 16 one on gfx950?  ("K tail 100 -> 104 with one v_mfma_f32_32x32x8_f16 step")
// Back-to-back issue on every SIMD, operands in registers, two accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const int tid = threadIdx.x;
    v8h a8, b8;
    v4h a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(tid * 1e-3f + i); b8[i] = (_Float16)1.0001f; }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    v16f c0, c1;
    for (int q = 0; q < 16; ++q) c0[q] = c1[q] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE == 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c1, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c1, 0, 0, 0);
            }
        }
    }
    float r = c0[0] + c1[1];
    if (r == 123.456f) out[0] = r;
}
template <int MODE>
float run(int iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(out); return ms;
}
int main() {
    const int iters = 20000;  // 16 MFMAs per iteration
    const float k16 = run<0>(iters), k8 = run<1>(iters);
    printf("16 MFMAs x %d iterations per wave, one wave per SIMD\n", iters);
    printf("  v_mfma_f32_32x32x16_f16 : %.3f ms  (%.1f ns per MFMA)\n", k16, k16 * 1e6 / (iters * 16.0));
    printf("  v_mfma_f32_32x32x8_f16  : %.3f ms  (%.1f ns per MFMA)   ratio K8 / K16 = %.3f\n", k8, k8 * 1e6 / (iters * 16.0), k8 / k16);
    return 0;
}
