// Row gather / scatter-add / Adagrad kernels (HBM-bound).
//
// Layout: node table fp32 row-major [num_nodes, ld] (ld == d for Marius's on-disk layout, embeddings.bin),
// ids int64 ascending (map_tensors returns them sorted) so consecutive workgroups walk monotone addresses.
// A row of d floats is moved as 16/8/4-byte pieces by TX adjacent lanes (TX = pow2 >= d/VEC, <= 64), TY rows per
// workgroup pass, UNROLL passes in flight per thread so that >= UNROLL independent loads are outstanding per lane.
#include <cmath>

#include "common.h"

namespace marius {

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = float4; };
template <>
struct VecT<2> { using type = float2; };
template <>
struct VecT<1> { using type = float; };

constexpr int ROW_UNROLL = 4;
// rows in flight per thread row of the gather.  tools/micro/gather_variants.hip (random 400-B rows of a 34 GB table, fresh ids per
// launch, launches back to back) gives 37.8 us with 4, 34.8 us with 2, 38 us with 8; inside the training step (one launch between
// other kernels) 4 is faster: 57-59 us vs 67-70 us between events.  4 it is.
static int gather_unroll() { return 4; }

template <int VEC, int NT, int GATHER_UNROLL>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ ta, const float* __restrict__ tb,
                                                          int64_t table_ld, const int64_t* __restrict__ ids, int64_t n,
                                                          int vpr, float* __restrict__ oa, float* __restrict__ ob,
                                                          int64_t out_ld, const int64_t* __restrict__ n_dev) {
    using V = typename VecT<VEC>::type;
    const int tx = threadIdx.x, ty = threadIdx.y, TX = blockDim.x, TY = blockDim.y;
    if (n_dev) {  // capacity-sized id list: only the first *n_dev entries are rows (the tail is padding)
        const int64_t nd = *n_dev;
        n = nd < n ? nd : n;
    }
    int64_t rows[GATHER_UNROLL];
    int64_t src[GATHER_UNROLL];
#pragma unroll
    for (int k = 0; k < GATHER_UNROLL; ++k) {
        rows[k] = ((int64_t)blockIdx.x * GATHER_UNROLL + k) * TY + ty;
        src[k] = rows[k] < n ? ids[rows[k]] : -1;
    }
    for (int c = tx; c < vpr; c += TX) {
        V va[GATHER_UNROLL], vb[GATHER_UNROLL];
#pragma unroll
        for (int k = 0; k < GATHER_UNROLL; ++k) {
            if (src[k] >= 0) {
                va[k] = reinterpret_cast<const V*>(ta + src[k] * table_ld)[c];
                if (NT == 2) vb[k] = reinterpret_cast<const V*>(tb + src[k] * table_ld)[c];
            }
        }
#pragma unroll
        for (int k = 0; k < GATHER_UNROLL; ++k) {
            if (src[k] >= 0) {
                reinterpret_cast<V*>(oa + rows[k] * out_ld)[c] = va[k];
                if (NT == 2) reinterpret_cast<V*>(ob + rows[k] * out_ld)[c] = vb[k];
            }
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* __restrict__ table, int64_t table_ld,
                                                               const int64_t* __restrict__ ids, int64_t n, int vpr,
                                                               const float* __restrict__ delta, int64_t delta_ld) {
    using V = typename VecT<VEC>::type;
    const int tx = threadIdx.x, ty = threadIdx.y, TX = blockDim.x, TY = blockDim.y;
    int64_t rows[ROW_UNROLL];
    int64_t dst[ROW_UNROLL];
#pragma unroll
    for (int k = 0; k < ROW_UNROLL; ++k) {
        rows[k] = ((int64_t)blockIdx.x * ROW_UNROLL + k) * TY + ty;
        dst[k] = rows[k] < n ? ids[rows[k]] : -1;
    }
    for (int c = tx; c < vpr; c += TX) {
        float w[ROW_UNROLL][VEC], dl[ROW_UNROLL][VEC];
#pragma unroll
        for (int k = 0; k < ROW_UNROLL; ++k) {
            if (dst[k] >= 0) {
                V a = reinterpret_cast<const V*>(table + dst[k] * table_ld)[c];
                V b = reinterpret_cast<const V*>(delta + rows[k] * delta_ld)[c];
                __builtin_memcpy(w[k], &a, sizeof(V));
                __builtin_memcpy(dl[k], &b, sizeof(V));
            }
        }
#pragma unroll
        for (int k = 0; k < ROW_UNROLL; ++k) {
            if (dst[k] >= 0) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) w[k][e] += dl[k][e];
                V o;
                __builtin_memcpy(&o, w[k], sizeof(V));
                reinterpret_cast<V*>(table + dst[k] * table_ld)[c] = o;
            }
        }
    }
}

// ds = g^2 ; state += ds ; dw = -lr * (g / (sqrt(state) + eps))     (batch.cpp:67-69, same op order)
__global__ __launch_bounds__(256) void adagrad_rule_kernel(const float* __restrict__ grad, float* __restrict__ state,
                                                           float* __restrict__ dw, float* __restrict__ ds, int64_t n,
                                                           float lr, float eps) {
#pragma clang fp contract(off)  // bit-faithful to the reference's separate pow/add/sqrt/div/mul ops
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float g = grad[i];
        float u = g * g;
        float s = state[i] + u;
        state[i] = s;
        ds[i] = u;
        dw[i] = -lr * (g / (sqrtf(s) + eps));
    }
}

// optim.cpp:114-145: g' = g + wd*w ; sum += g'^2 ; w -= lr * g' / (sqrt(sum) + eps)
__global__ __launch_bounds__(256) void dense_adagrad_kernel(float* __restrict__ w, float* __restrict__ sum,
                                                            const float* __restrict__ grad, int64_t n, float lr,
                                                            float eps, float wd) {
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float g = grad[i];
        float p = w[i];
        if (wd != 0.f) g = g + wd * p;
        float s = sum[i] + g * g;
        sum[i] = s;
        w[i] = p - lr * (g / (sqrtf(s) + eps));
    }
}

// optim.cpp:186-232 (AdamOptimizer::step), same op order: g' = g + wd*w; m = m*b1 + g'*(1-b1); v = v*b2 + (1-b2)*g'*g';
// denom = sqrt(max_v or v) / sqrt(bc2) + eps; w += -(lr / bc1) * (m / denom)
__global__ __launch_bounds__(256) void dense_adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, float* __restrict__ vmax,
                                                         const float* __restrict__ grad, int64_t n, float step_size, float b1, float b2, float eps,
                                                         float wd, float sqrt_bc2) {
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    for (; i < n; i += stride) {
        float g = grad[i];
        const float p = w[i];
        if (wd != 0.f) g = g + wd * p;
        const float mn = m[i] * b1 + g * omb1;
        const float vn = v[i] * b2 + (omb2 * g) * g;
        m[i] = mn;
        v[i] = vn;
        float vv = vn;
        if (vmax) {
            vv = fmaxf(vmax[i], vn);
            vmax[i] = vv;
        }
        const float denom = sqrtf(vv) / sqrt_bc2 + eps;
        w[i] = p + (-step_size) * (mn / denom);
    }
}

static void row_geometry(int vpr, dim3& block, int& rows_per_block, int unroll = ROW_UNROLL) {
    int tx = 1;
    while (tx < vpr && tx < 64) tx <<= 1;
    int ty = 256 / tx;
    block = dim3(tx, ty, 1);
    rows_per_block = ty * unroll;
}

template <int NT>
static int launch_gather(const float* ta, const float* tb, int64_t table_ld, const int64_t* ids, int64_t n, int d,
                         float* oa, float* ob, int64_t out_ld, hipStream_t st, const int64_t* n_dev = nullptr) {
    if (n == 0) return MARIUS_OK;
    int vec = row_vec_width(ta, table_ld, d);
    int v2 = row_vec_width(oa, out_ld, d);
    vec = vec < v2 ? vec : v2;
    if (NT == 2) {
        int v3 = row_vec_width(tb, table_ld, d), v4 = row_vec_width(ob, out_ld, d);
        vec = vec < v3 ? vec : v3;
        vec = vec < v4 ? vec : v4;
    }
    int vpr = d / vec;
    dim3 block;
    int rpb;
    const int gu = gather_unroll();
    row_geometry(vpr, block, rpb, vec == 4 ? gu : 4);
    dim3 grid((unsigned)cdiv(n, rpb));
    ProfScope ps(PROF_GATHER, st);
    if (vec == 4 && gu == 2)
        gather_rows_kernel<4, NT, 2><<<grid, block, 0, st>>>(ta, tb, table_ld, ids, n, vpr, oa, ob, out_ld, n_dev);
    else if (vec == 4)
        gather_rows_kernel<4, NT, 4><<<grid, block, 0, st>>>(ta, tb, table_ld, ids, n, vpr, oa, ob, out_ld, n_dev);
    else if (vec == 2)
        gather_rows_kernel<2, NT, 4><<<grid, block, 0, st>>>(ta, tb, table_ld, ids, n, vpr, oa, ob, out_ld, n_dev);
    else
        gather_rows_kernel<1, NT, 4><<<grid, block, 0, st>>>(ta, tb, table_ld, ids, n, vpr, oa, ob, out_ld, n_dev);
    return check_launch("gather_rows");
}

}  // namespace marius

using namespace marius;

extern "C" int marius_gather_rows(const float* table, int64_t table_ld, const int64_t* ids, int64_t n, int32_t d,
                                  float* out, int64_t out_ld, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && d > 0 && table_ld >= d && out_ld >= d, "gather_rows: bad sizes n=%ld d=%d", (long)n, d);
    MARIUS_REQUIRE(n == 0 || (table && ids && out), "gather_rows: null pointer");
    return launch_gather<1>(table, nullptr, table_ld, ids, n, d, out, nullptr, out_ld, as_stream(stream));
}

extern "C" int marius_gather_rows_counted(const float* table, int64_t table_ld, const int64_t* ids, int64_t capacity, const int64_t* num_rows_dev,
                                          int32_t d, float* out, int64_t out_ld, marius_stream_t stream) {
    MARIUS_REQUIRE(capacity >= 0 && d > 0 && table_ld >= d && out_ld >= d, "gather_rows_counted: bad sizes n=%ld d=%d", (long)capacity, d);
    MARIUS_REQUIRE(capacity == 0 || (table && ids && out && num_rows_dev), "gather_rows_counted: null pointer");
    return launch_gather<1>(table, nullptr, table_ld, ids, capacity, d, out, nullptr, out_ld, as_stream(stream), num_rows_dev);
}

extern "C" int marius_gather_rows2(const float* table_a, const float* table_b, int64_t table_ld, const int64_t* ids,
                                   int64_t n, int32_t d, float* out_a, float* out_b, int64_t out_ld,
                                   marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && d > 0 && table_ld >= d && out_ld >= d, "gather_rows2: bad sizes");
    MARIUS_REQUIRE(n == 0 || (table_a && table_b && ids && out_a && out_b), "gather_rows2: null pointer");
    return launch_gather<2>(table_a, table_b, table_ld, ids, n, d, out_a, out_b, out_ld, as_stream(stream));
}

extern "C" int marius_scatter_add_rows(float* table, int64_t table_ld, const int64_t* ids, int64_t n, int32_t d,
                                       const float* delta, int64_t delta_ld, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && d > 0 && table_ld >= d && delta_ld >= d, "scatter_add_rows: bad sizes");
    MARIUS_REQUIRE(n == 0 || (table && ids && delta), "scatter_add_rows: null pointer");
    if (n == 0) return MARIUS_OK;
    int vec = row_vec_width(table, table_ld, d);
    int v2 = row_vec_width(delta, delta_ld, d);
    vec = vec < v2 ? vec : v2;
    int vpr = d / vec;
    dim3 block;
    int rpb;
    row_geometry(vpr, block, rpb);
    dim3 grid((unsigned)cdiv(n, rpb));
    hipStream_t st = as_stream(stream);
    if (vec == 4)
        scatter_add_rows_kernel<4><<<grid, block, 0, st>>>(table, table_ld, ids, n, vpr, delta, delta_ld);
    else if (vec == 2)
        scatter_add_rows_kernel<2><<<grid, block, 0, st>>>(table, table_ld, ids, n, vpr, delta, delta_ld);
    else
        scatter_add_rows_kernel<1><<<grid, block, 0, st>>>(table, table_ld, ids, n, vpr, delta, delta_ld);
    return check_launch("scatter_add_rows");
}

extern "C" int marius_adagrad_rule(const float* grad, float* state, float* dw, float* ds, int64_t n, float lr, float eps,
                                   marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0, "adagrad_rule: n < 0");
    if (n == 0) return MARIUS_OK;
    MARIUS_REQUIRE(grad && state && dw && ds, "adagrad_rule: null pointer");
    int64_t blocks = cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    adagrad_rule_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(grad, state, dw, ds, n, lr, eps);
    return check_launch("adagrad_rule");
}

extern "C" int marius_dense_adagrad_step(float* param, float* state_sum, const float* grad, int64_t n, float lr,
                                         float eps, float weight_decay, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0, "dense_adagrad_step: n < 0");
    if (n == 0) return MARIUS_OK;
    MARIUS_REQUIRE(param && state_sum && grad, "dense_adagrad_step: null pointer");
    int64_t blocks = cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    dense_adagrad_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(param, state_sum, grad, n, lr, eps,
                                                                                     weight_decay);
    return check_launch("dense_adagrad_step");
}

extern "C" int marius_dense_adam_step(float* param, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, const float* grad, int64_t n, float lr,
                                      float beta1, float beta2, float eps, float weight_decay, int64_t num_steps, marius_stream_t stream) {
    MARIUS_REQUIRE(n >= 0 && num_steps >= 0, "dense_adam_step: bad sizes");
    if (n == 0) return MARIUS_OK;
    MARIUS_REQUIRE(param && exp_avg && exp_avg_sq && grad, "dense_adam_step: null pointer");
    // float arithmetic as in the reference: 1 - std::pow(beta, num_steps + 1) on floats (optim.cpp:204-205)
    const float bc1 = 1.f - std::pow(beta1, (float)(num_steps + 1));
    const float bc2 = 1.f - std::pow(beta2, (float)(num_steps + 1));
    int64_t blocks = cdiv(n, 256);
    if (blocks > 8192) blocks = 8192;
    dense_adam_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(param, exp_avg, exp_avg_sq, max_exp_avg_sq, grad, n, lr / bc1, beta1,
                                                                                  beta2, eps, weight_decay, std::sqrt(bc2));
    return check_launch("dense_adam_step");
}
