// bf16 32x32x16 MFMA rate when the B operand is re-read from LDS for every group of MFMAs (as in the contraction kernels) (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
// MODE 0: operands in registers; 1: three ds_read_b128 per 6 MFMAs, prefetched one group ahead; 2: same plus a second wave per SIMD doing VALU
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[3 * 128 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * 128 * 32; i += blockDim.x) lds[i] = (__bf16)(i * 1e-4f);
    __syncthreads();
    float r = 0.f;
    if (wave < 4) {
        v8bf ax, ay, az;
        for (int i = 0; i < 8; ++i) { ax[i] = (__bf16)(tid * 1e-3f); ay[i] = (__bf16)1.0001f; az[i] = (__bf16)0.5f; }
        v16f a0, a1;
        for (int q = 0; q < 16; ++q) a0[q] = a1[q] = 0.f;
        const __bf16* bp = lds + (lane & 31) * 32 + 8 * (lane >> 5);
        v8bf bH[2], bM[2], bL[2];
        bH[0] = *reinterpret_cast<const v8bf*>(bp);
        bM[0] = *reinterpret_cast<const v8bf*>(bp + 4096);
        bL[0] = *reinterpret_cast<const v8bf*>(bp + 8192);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = s & 1;
                if (MODE >= 1) {
                    const __bf16* q = bp + ((s + 1) & 3) * 1024 + 16 * (i & 1);
                    bH[c ^ 1] = *reinterpret_cast<const v8bf*>(q);
                    bM[c ^ 1] = *reinterpret_cast<const v8bf*>(q + 4096);
                    bL[c ^ 1] = *reinterpret_cast<const v8bf*>(q + 8192);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int u = MODE >= 1 ? c : 0;
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az, bH[u], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bL[u], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay, bM[u], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay, bH[u], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bM[u], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bH[u], a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        r = a0[0] + a1[1];
    } else if (MODE == 2) {
        __builtin_amdgcn_s_setprio(3);
        float z[8];
        for (int j = 0; j < 8; ++j) z[j] = tid + j;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z[c]) : "v"(1.0001f));
        for (int j = 0; j < 8; ++j) r += z[j];
    }
    if (r == 123.456f) out[0] = r;
}
template <int MODE>
float run(int iters, int threads) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, threads>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256, threads>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(out); return ms;
}
int main() {
    const int iters = 4000;  // 24 MFMAs per iteration
    printf("24 x v_mfma_f32_32x32x16_bf16 per iteration, %d iterations (32 cycles each at 2.2 GHz = %.3f ms)\n", iters, iters * 24 * 32 / 2.2e6);
    printf("  operands in registers            : %.3f ms\n", run<0>(iters, 256));
    printf("  B operand from LDS (prefetched)  : %.3f ms\n", run<1>(iters, 256));
    printf("  + VALU wave on the same SIMD     : %.3f ms\n", run<2>(iters, 512));
    return 0;
}
