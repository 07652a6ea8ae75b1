// Layer::post_hook of the embedding-only encoder (src/cpp/src/nn/layers/layer.cpp:9-16, applied by GeneralEncoder::forward,
// src/cpp/src/nn/encoders/encoder.cpp:195-257, to the batch's [U, d] rows): out = act(x + bias), act in {NONE, RELU, SIGMOID}
// (src/cpp/src/nn/activation.cpp:7-21), and its backward: gx = gy * act'(x + bias), bias.grad = column sums of gx.
//
// HBM-bound elementwise passes over U x d floats (U ~ 160 k rows of 400 B at the bench shape: 64 MB read + 64 MB written); a row's 16-byte
// pieces map one to one onto lanes, rows are strided over the grid.  The bias gradient is a DETERMINISTIC two-stage column sum: every block
// reduces its row range into one [d] partial (fixed row order per lane, fixed cross-lane order in LDS), a second launch adds the partials in
// block order — no float atomics, same bits on every run.
#include "common.h"

namespace marius {

constexpr int PH_THREADS = 256, PH_MAX_BLOCKS = 512;

__device__ __forceinline__ float ph_act(float v, int act) {
    if (act == MARIUS_ACT_RELU) return v > 0.f ? v : 0.f;                   // torch::relu
    if (act == MARIUS_ACT_SIGMOID) return 1.f / (1.f + expf(-v));            // torch::sigmoid
    return v;
}
// derivative in terms of the OUTPUT y = act(x + b): relu' = [y > 0], sigmoid' = y (1 - y)
__device__ __forceinline__ float ph_dact(float y, int act) {
    if (act == MARIUS_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == MARIUS_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

template <int VEC>
__global__ __launch_bounds__(PH_THREADS) void post_hook_kernel(const float* __restrict__ x, int64_t x_ld, const float* __restrict__ bias, int act, int64_t n, int vpr,
                                                               float* __restrict__ out, int64_t out_ld) {
    const int TX = vpr < PH_THREADS ? vpr : PH_THREADS, TY = PH_THREADS / TX, ty = threadIdx.x / TX, tx = threadIdx.x - ty * TX;
    if (ty >= TY) return;
    for (int c = tx; c < vpr; c += TX) {
        float b[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) b[e] = bias ? bias[c * VEC + e] : 0.f;
        for (int64_t r = (int64_t)blockIdx.x * TY + ty; r < n; r += (int64_t)gridDim.x * TY) {
            float v[VEC];
            __builtin_memcpy(v, x + r * x_ld + (int64_t)c * VEC, sizeof(float) * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = ph_act(v[e] + b[e], act);
            __builtin_memcpy(out + r * out_ld + (int64_t)c * VEC, v, sizeof(float) * VEC);
        }
    }
}

// gx = gy * act'(y); partial[blockIdx][0:d] = this block's column sums of gx (partial == nullptr: no bias)
template <int VEC>
__global__ __launch_bounds__(PH_THREADS) void post_hook_bwd_kernel(const float* __restrict__ gy, int64_t gy_ld, const float* __restrict__ y, int64_t y_ld, int act, int64_t n,
                                                                   int vpr, float* __restrict__ gx, int64_t gx_ld, float* __restrict__ partial) {
    __shared__ float red[PH_THREADS * 4];
    const int TX = vpr < PH_THREADS ? vpr : PH_THREADS, TY = PH_THREADS / TX, ty = threadIdx.x / TX, tx = threadIdx.x - ty * TX;
    const int d = vpr * VEC;
    for (int c0 = 0; c0 < vpr; c0 += TX) {  // (uniform trip count: every thread reaches the barriers)
        const int c = c0 + tx;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        if (ty < TY && c < vpr)
            for (int64_t r = (int64_t)blockIdx.x * TY + ty; r < n; r += (int64_t)gridDim.x * TY) {
                float g[VEC], o[VEC];
                __builtin_memcpy(g, gy + r * gy_ld + (int64_t)c * VEC, sizeof(float) * VEC);
                if (act != MARIUS_ACT_NONE) __builtin_memcpy(o, y + r * y_ld + (int64_t)c * VEC, sizeof(float) * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    g[e] = act != MARIUS_ACT_NONE ? g[e] * ph_dact(o[e], act) : g[e];
                    acc[e] += g[e];
                }
                if (gx != gy || act != MARIUS_ACT_NONE) __builtin_memcpy(gx + r * gx_ld + (int64_t)c * VEC, g, sizeof(float) * VEC);
            }
        if (partial) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = acc[e];
            __syncthreads();
            if (ty == 0 && c < vpr) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float s = 0.f;
                    for (int k = 0; k < TY; ++k) s += red[(k * TX + tx) * VEC + e];  // fixed order
                    partial[(int64_t)blockIdx.x * d + c * VEC + e] = s;
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(PH_THREADS) void post_hook_bias_sum_kernel(const float* __restrict__ partial, int nblocks, int d, float* __restrict__ bias_grad) {
    const int c = blockIdx.x * PH_THREADS + threadIdx.x;
    if (c >= d) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * d + c];  // block order: deterministic
    bias_grad[c] = s;
}

static int ph_blocks(int64_t n, int ty) {
    int64_t b = cdiv(n, (int64_t)ty * 4);
    if (b > PH_MAX_BLOCKS) b = PH_MAX_BLOCKS;
    return b < 1 ? 1 : (int)b;
}

}  // namespace marius

using namespace marius;

extern "C" size_t marius_layer_post_hook_workspace_bytes(int64_t n, int32_t d) {
    (void)n;
    return d > 0 ? (size_t)PH_MAX_BLOCKS * (size_t)d * sizeof(float) : 0;
}

extern "C" int marius_layer_post_hook(const float* x, int64_t x_ld, const float* bias, int32_t activation, int64_t n, int32_t d, float* out, int64_t out_ld,
                                      marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && d > 0 && x_ld >= d && out_ld >= d, "layer_post_hook: bad sizes");
    MARIUS_REQUIRE(activation == MARIUS_ACT_NONE || activation == MARIUS_ACT_RELU || activation == MARIUS_ACT_SIGMOID, "Unsupported activation function");  // activation.cpp:19
    if (n == 0) return MARIUS_OK;
    MARIUS_REQUIRE(x && out, "layer_post_hook: null pointer");
    int vec = row_vec_width(x, x_ld, d);
    const int v2 = row_vec_width(out, out_ld, d);
    vec = vec < v2 ? vec : v2;
    const int vpr = d / vec, tx = vpr < PH_THREADS ? vpr : PH_THREADS, ty = PH_THREADS / tx;
    dim3 grid((unsigned)ph_blocks(n, ty)), block(PH_THREADS);
    hipStream_t st = as_stream(stream);
    if (vec == 4) post_hook_kernel<4><<<grid, block, 0, st>>>(x, x_ld, bias, activation, n, vpr, out, out_ld);
    else if (vec == 2) post_hook_kernel<2><<<grid, block, 0, st>>>(x, x_ld, bias, activation, n, vpr, out, out_ld);
    else post_hook_kernel<1><<<grid, block, 0, st>>>(x, x_ld, bias, activation, n, vpr, out, out_ld);
    return check_launch("layer_post_hook");
}

extern "C" int marius_layer_post_hook_backward(const float* gy, int64_t gy_ld, const float* y, int64_t y_ld, int32_t activation, int64_t n, int32_t d, float* gx,
                                               int64_t gx_ld, float* bias_grad, void* workspace, size_t workspace_bytes, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && d > 0 && gy_ld >= d && gx_ld >= d, "layer_post_hook_backward: bad sizes");
    MARIUS_REQUIRE(activation == MARIUS_ACT_NONE || activation == MARIUS_ACT_RELU || activation == MARIUS_ACT_SIGMOID, "Unsupported activation function");
    MARIUS_REQUIRE(activation == MARIUS_ACT_NONE || (y && y_ld >= d), "layer_post_hook_backward: the activation's derivative needs the forward output");
    MARIUS_REQUIRE(!bias_grad || (workspace && workspace_bytes >= marius_layer_post_hook_workspace_bytes(n, d)), "layer_post_hook_backward: workspace too small");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        if (bias_grad && hipMemsetAsync(bias_grad, 0, sizeof(float) * d, st) != hipSuccess) return MARIUS_ERR_HIP;
        return MARIUS_OK;
    }
    MARIUS_REQUIRE(gy && gx, "layer_post_hook_backward: null pointer");
    int vec = row_vec_width(gy, gy_ld, d);
    const int v2 = row_vec_width(gx, gx_ld, d), v3 = activation == MARIUS_ACT_NONE ? 4 : row_vec_width(y, y_ld, d);
    vec = vec < v2 ? vec : v2;
    vec = vec < v3 ? vec : v3;
    const int vpr = d / vec, tx = vpr < PH_THREADS ? vpr : PH_THREADS, ty = PH_THREADS / tx;
    const int nb = ph_blocks(n, ty);
    float* partial = bias_grad ? (float*)workspace : nullptr;
    dim3 grid((unsigned)nb), block(PH_THREADS);
    if (vec == 4) post_hook_bwd_kernel<4><<<grid, block, 0, st>>>(gy, gy_ld, y, y_ld, activation, n, vpr, gx, gx_ld, partial);
    else if (vec == 2) post_hook_bwd_kernel<2><<<grid, block, 0, st>>>(gy, gy_ld, y, y_ld, activation, n, vpr, gx, gx_ld, partial);
    else post_hook_bwd_kernel<1><<<grid, block, 0, st>>>(gy, gy_ld, y, y_ld, activation, n, vpr, gx, gx_ld, partial);
    if (bias_grad) post_hook_bias_sum_kernel<<<dim3((unsigned)cdiv(d, PH_THREADS)), dim3(PH_THREADS), 0, st>>>(partial, nb, d, bias_grad);
    return check_launch("layer_post_hook_backward");
}
