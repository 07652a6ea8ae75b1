// v_mfma_f32_32x32x16_bf16 rate vs operand pattern and accumulator register file (gfx950).  6 MFMAs per group on 2 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
#define MF_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
// PAT 0: same (a, b) everywhere; 1: three A x three B operands, all different per MFMA (as in the split kernels); 2: A fixed per pair, B alternating
template <int PAT, bool AGPR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v8bf a[3], b[3];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 8; ++i) { a[j][i] = (__bf16)(threadIdx.x * 1e-3f + j); b[j][i] = (__bf16)(1.0001f + j); }
    v16f c0, c1;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#define M(acc, x, y) do { if (AGPR) MF_A(acc, x, y); else MF_V(acc, x, y); } while (0)
        if (PAT == 0) { M(c0, a[0], b[0]); M(c1, a[0], b[0]); M(c0, a[0], b[0]); M(c1, a[0], b[0]); M(c0, a[0], b[0]); M(c1, a[0], b[0]); }
        if (PAT == 1) { M(c0, a[2], b[0]); M(c1, a[0], b[2]); M(c0, a[1], b[1]); M(c1, a[1], b[0]); M(c0, a[0], b[1]); M(c1, a[0], b[0]); }
        if (PAT == 2) { M(c0, a[0], b[0]); M(c1, a[0], b[1]); M(c0, a[1], b[0]); M(c1, a[1], b[1]); M(c0, a[2], b[0]); M(c1, a[2], b[1]); }
#undef M
    }
    float r = c0[0] + c1[1];
    if (r == 123.456f) out[0] = r;
}
template <int PAT, bool AGPR>
float run(int iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PAT, AGPR><<<256, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<PAT, AGPR><<<256, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(out); return ms;
}
int main() {
    const int iters = 16000;
    printf("6 MFMAs per iteration, %d iterations; 32 cycles each at 2.2 GHz = %.3f ms\n", iters, iters * 6 * 32 / 2.2e6);
    printf("  ArchVGPR acc: same operands %.3f | all different %.3f | A per pair, B alternating %.3f\n", run<0, false>(iters), run<1, false>(iters), run<2, false>(iters));
    printf("  AccVGPR  acc: same operands %.3f | all different %.3f | A per pair, B alternating %.3f\n", run<0, true>(iters), run<1, true>(iters), run<2, true>(iters));
    return 0;
}
