// Entity-aware regulariser of the BPR loss, reference graph_recsys_benchmark/models/base.py:50-73:
//   item term  d_i = (|x[i] - x[e+]|^2 - |x[i] - x[e-]|^2) * mask_i          (batch columns 1, 3, 4, 5)
//   user term  d_u = (|x[u] - x[f+]|^2 - |x[u] - x[f-]|^2) * mask_u          (batch columns 0, 6, 7, 8)
//   reg = -sum_b log sigmoid(d_i) - sum_b log sigmoid(d_u)
// The reference gathers eight [B, F0] row blocks with advanced indexing and runs ~20 elementwise kernels; here one launch
// reads the six rows of a batch row once (F0 / 4 lanes x float4 per batch row), reduces the four squared distances inside
// the lane group and sums the log-sigmoid terms in a fixed order (block partials, then one block): bitwise reproducible.
// With `grad_rows` the same launch also writes the gradient of reg with respect to the six gathered rows ([6B, F0], order
// i, e+, e-, u, f+, f-): the host adds them into dx with one index_add (autograd's own scatter for an indexed read).
#include "common.h"

namespace pea {
namespace {

__device__ __forceinline__ float4 ld4e(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float sq4(float4 d) { return (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
__device__ __forceinline__ float4 mul4(float4 a, float f) { return make_float4(a.x * f, a.y * f, a.z * f, a.w * f); }

// G lanes (a power of two, 4 * G >= F) per batch row, 256 / G batch rows per workgroup
template <int G>
__global__ __launch_bounds__(256) void entity_kernel(int64_t B, int F, int64_t N, const float *__restrict__ x, int64_t ldx,
                                                     const int64_t *__restrict__ batch, int64_t stride, float *block_item,
                                                     float *block_user, float *grad_rows, int *err) {
    constexpr int RPB = 256 / G;
    __shared__ float red_i[RPB], red_u[RPB];
    const int sub = (int)threadIdx.x / G, sl = (int)threadIdx.x % G;
    const int64_t b = (int64_t)blockIdx.x * RPB + sub;
    const bool active = sl * 4 < F;
    const int c4 = active ? sl * 4 : 0;
    float ti = 0.f, tu = 0.f;
    bool valid = b < B;
    int64_t id[6] = {0, 0, 0, 0, 0, 0};
    float mi = 0.f, mu = 0.f;
    if (valid) {
        const int64_t *t = batch + b * stride;
        id[0] = t[1]; id[1] = t[3]; id[2] = t[4]; id[3] = t[0]; id[4] = t[6]; id[5] = t[7];
        mi = (float)t[5];
        mu = (float)t[8];
        for (int q = 0; q < 6; ++q)
            if (id[q] < 0 || id[q] >= N) valid = false;
        if (!valid && sl == 0) atomicOr(err, 1);
    }
    float4 r[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) r[q] = (valid && active) ? ld4e(x + id[q] * ldx + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 dip = sub4(r[0], r[1]), din = sub4(r[0], r[2]), dup = sub4(r[3], r[4]), dun = sub4(r[3], r[5]);
    float ip = sq4(dip), in_ = sq4(din), up = sq4(dup), un = sq4(dun);
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        ip += __shfl_xor(ip, off);
        in_ += __shfl_xor(in_, off);
        up += __shfl_xor(up, off);
        un += __shfl_xor(un, off);
    }
    if (valid) {
        const float di = (ip - in_) * mi, du = (up - un) * mu;
        const float si = 1.0f / (1.0f + expf(-di)), su = 1.0f / (1.0f + expf(-du));
        ti = logf(si);   // sigmoid then log in fp32, no clamp: the reference's own formula (may give -inf)
        tu = logf(su);
        if (grad_rows && active) {
            // d(-log sigmoid(d))/dd = -(1 - sigmoid(d));  d = (|a - p|^2 - |a - n|^2) * m
            const float ci = -(1.0f - si) * mi * 2.0f, cu = -(1.0f - su) * mu * 2.0f;
            float *g = grad_rows + (size_t)b * 6 * (size_t)F + c4;
            *reinterpret_cast<float4 *>(g + 0 * (size_t)F) = mul4(sub4(dip, din), ci);    // d/dx[i]  = 2 m c ((a-p) - (a-n))
            *reinterpret_cast<float4 *>(g + 1 * (size_t)F) = mul4(dip, -ci);              // d/dx[e+] = -2 m c (a - p)
            *reinterpret_cast<float4 *>(g + 2 * (size_t)F) = mul4(din, ci);               // d/dx[e-] = +2 m c (a - n)
            *reinterpret_cast<float4 *>(g + 3 * (size_t)F) = mul4(sub4(dup, dun), cu);
            *reinterpret_cast<float4 *>(g + 4 * (size_t)F) = mul4(dup, -cu);
            *reinterpret_cast<float4 *>(g + 5 * (size_t)F) = mul4(dun, cu);
        }
    } else if (grad_rows && active && b < B) {
        float *g = grad_rows + (size_t)b * 6 * (size_t)F + c4;
        for (int q = 0; q < 6; ++q) *reinterpret_cast<float4 *>(g + q * (size_t)F) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (sl == 0) {
        red_i[sub] = ti;
        red_u[sub] = tu;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int k = 0; k < RPB; ++k) {   // batch-row order inside the block
            a += red_i[k];
            c += red_u[k];
        }
        block_item[blockIdx.x] = a;
        block_user[blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(256) void entity_final_kernel(int n_blocks, const float *block_item, const float *block_user,
                                                           const int *err, float *out) {
    __shared__ float ri[256], ru[256];
    float a = 0.f, c = 0.f;
    for (int k = threadIdx.x; k < n_blocks; k += 256) {
        a += block_item[k];
        c += block_user[k];
    }
    ri[threadIdx.x] = a;
    ru[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            ri[threadIdx.x] += ri[threadIdx.x + off];
            ru[threadIdx.x] += ru[threadIdx.x + off];
        }
        __syncthreads();
    }
    // item_reg_los + user_reg_los, each a negated sum (models/base.py:70-72); NaN if a node id was out of range
    if (threadIdx.x == 0) out[0] = err[0] ? __int_as_float(0x7fc00000) : (-ri[0]) + (-ru[0]);
}

int rows_per_block(int F) {
    int g = 4;
    while (g * 4 < F) g <<= 1;
    return 256 / g;
}

}  // namespace
}  // namespace pea

using namespace pea;

extern "C" size_t pea_entity_reg_workspace_bytes(int64_t B, int emb_dim) {
    if (B < 0 || emb_dim <= 0 || emb_dim > 256) return 0;
    const int64_t blocks = (B + rows_per_block(emb_dim) - 1) / rows_per_block(emb_dim);
    return 16 + (size_t)(2 * blocks + 2) * sizeof(float);
}

extern "C" int pea_entity_reg(int64_t B, int emb_dim, int64_t num_nodes, const float *x, int64_t ldx, const int64_t *batch,
                              int64_t batch_stride, float *out_reg, float *grad_rows, void *workspace, size_t workspace_bytes,
                              void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(B >= 0 && emb_dim > 0 && emb_dim % 4 == 0 && emb_dim <= 256, PEA_ERR_ARG,
                "entity_reg: B=%lld emb_dim=%d (a multiple of 4, <= 256)", (long long)B, emb_dim);
    PEA_REQUIRE(x && batch && out_reg && workspace, PEA_ERR_ARG, "entity_reg: null pointer");
    PEA_REQUIRE(batch_stride >= 9 && ldx >= emb_dim && ldx % 4 == 0, PEA_ERR_ARG,
                "entity_reg: the batch needs 9 columns (u, i+, i-, 6 entity columns), x rows a stride that is a multiple of 4");
    PEA_REQUIRE(workspace_bytes >= pea_entity_reg_workspace_bytes(B, emb_dim), PEA_ERR_NOMEM, "entity_reg: workspace too small");
    const int rpb = rows_per_block(emb_dim);
    const int blocks = (int)((B + rpb - 1) / rpb);
    int *err = (int *)workspace;
    float *bi = (float *)workspace + 4, *bu = bi + blocks;
    PEA_MEMSET_ASYNC(err, 0, sizeof(int), stream);
    ProfScope ps("entity_reg", stream, (double)B * (72.0 + 24.0 * emb_dim));
    if (blocks > 0) {
#define PEA_ENT_CASE(g)                                                                                                   \
    case g:                                                                                                               \
        PEA_LAUNCH(entity_kernel<g>, dim3(blocks), dim3(256), 0, stream, B, emb_dim, num_nodes, x, ldx, batch,   \
                           batch_stride, bi, bu, grad_rows, err);                                                         \
        break;
        switch (256 / rpb) {
            PEA_ENT_CASE(4)
            PEA_ENT_CASE(8)
            PEA_ENT_CASE(16)
            PEA_ENT_CASE(32)
            default:
                PEA_LAUNCH(entity_kernel<64>, dim3(blocks), dim3(256), 0, stream, B, emb_dim, num_nodes, x, ldx, batch,
                                   batch_stride, bi, bu, grad_rows, err);
        }
#undef PEA_ENT_CASE
        PEA_HIP(hipGetLastError());
    }
    PEA_LAUNCH(entity_final_kernel, dim3(1), dim3(256), 0, stream, blocks, bi, bu, err, out_reg);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}
