// Dense per-layer transforms of the PEA path for gfx950: h = x.W^T (GAT), x.W (GCN),
// lin_rel(mean) + lin_root(x) (SAGE) -- the torch.nn.Linear / matmul calls inside the PyG convs
// (reference call site graph_recsys_benchmark/models/base.py:138-139, SURVEY.md 2.1 "dense transform").
//
// The weights of every layer are re-packed each forward (they change every optimizer step) into
// k-major blocks  B[k][col]  so one GEMM job can serve several channels that share the same input
// (all P first-layer channels read the same x, models/base.py:192-193).  For GAT the two attention
// projections are folded into the GEMM as extra output columns:
//     a_src[n,h] = sum_f att_j[h,f] * (W x_n)[h,f] = x_n . (W_h^T att_j[h])      (likewise a_dst / att_i)
// so the per-edge logits need one scalar per endpoint and no [M, F] temporaries.
//
// fp32 VALU kernel, 64x64 output tile per 256-thread block, 4x4 micro-tile per thread, operands staged
// through LDS in k-chunks of 64.  (The transforms are ~10 % of the forward's bytes; MFMA f32 is a later step.)
#include "common.h"

namespace pea {
namespace {

constexpr int TM = 64, TN = 64, TK = 64, LDS_LD = 68;  // 68 floats: keeps float4 alignment, breaks bank stride

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

__device__ __forceinline__ void store_cols(const GemmJob &J, int64_t orow, int c, float4 v) {
    float vv[4] = {v.x, v.y, v.z, v.w};
    // fast path: the four columns sit inside one segment, 16-byte aligned at the destination
    for (int s = 0; s < J.n_seg; ++s) {
        const GemmSegment &S = J.seg[s];
        if (c >= S.c0 && c + 4 <= S.c1 && ((c - S.c0) & 3) == 0 && (S.ld & 3) == 0) {
            if (S.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<float4 *>(S.dst + orow * S.ld + (c - S.c0)) = v;
            return;
        }
    }
    for (int i = 0; i < 4; ++i) {
        const int cc = c + i;
        for (int s = 0; s < J.n_seg; ++s) {
            const GemmSegment &S = J.seg[s];
            if (cc >= S.c0 && cc < S.c1) {
                S.dst[orow * S.ld + (cc - S.c0)] = S.relu ? fmaxf(vv[i], 0.f) : vv[i];
                break;
            }
        }
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(const GemmJob J, const int *__restrict__ rows, int64_t n_rows) {
    __shared__ float As[TK][LDS_LD];
    __shared__ float Bs[TK][LDS_LD];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TM;
    const int col0 = blockIdx.y * TN;
    const int K = J.K1 + J.K2;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int kc = 0; kc < K; kc += TK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;
            const int r = idx >> 4, kq = idx & 15;
            const int64_t grow = row0 + r;
            const int k = kc + kq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow < n_rows && k < K) {
                const int64_t srow = rows ? rows[grow] : grow;
                v = k < J.K1 ? ld4(J.A1 + srow * J.lda1 + k) : ld4(J.A2 + srow * J.lda2 + (k - J.K1));
            }
            As[kq * 4 + 0][r] = v.x;
            As[kq * 4 + 1][r] = v.y;
            As[kq * 4 + 2][r] = v.z;
            As[kq * 4 + 3][r] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;
            const int k = idx >> 4, cq = idx & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kc + k < K && col0 + cq * 4 < J.n_out) v = ld4(J.B + (size_t)(kc + k) * J.ldb + col0 + cq * 4);
            *reinterpret_cast<float4 *>(&Bs[k][cq * 4]) = v;
        }
        __syncthreads();
        const int kn = min(TK, K - kc);
#pragma unroll 8
        for (int k = 0; k < kn; ++k) {
            const float4 a = *reinterpret_cast<const float4 *>(&As[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4 *>(&Bs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    const int c = col0 + tx * 4;
    if (c >= J.n_out) return;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (J.bias) bias = ld4(J.bias + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t grow = row0 + ty * 4 + i;
        if (grow >= n_rows) continue;
        const int64_t orow = rows ? rows[grow] : grow;
        store_cols(J, orow, c,
                   make_float4(acc[i][0] + bias.x, acc[i][1] + bias.y, acc[i][2] + bias.z, acc[i][3] + bias.w));
    }
}

struct PackLaunch {
    int n;
    PackJob j[24];
};

// one block per layer: writes the k-major weight block (and the folded attention columns)
__global__ __launch_bounds__(256) void pack_kernel(const PackLaunch L) {
    const PackJob &J = L.j[blockIdx.x];
    const int tid = threadIdx.x;
    int krows = J.in;
    if (J.kind == PEA_KIND_GAT) {
        for (int idx = tid; idx < J.HF * J.in; idx += 256) {
            const int o = idx / J.in, k = idx % J.in;
            J.B[(size_t)k * J.ldb + o] = J.w0[idx];
        }
        const int heads = J.HF / J.F;
        for (int idx = tid; idx < heads * J.in; idx += 256) {
            const int h = idx / J.in, k = idx % J.in;
            float vs = 0.f, vd = 0.f;
            for (int f = 0; f < J.F; ++f) {
                const float w = J.w0[(size_t)(h * J.F + f) * J.in + k];
                vd = fmaf(J.w1[h * J.F + f], w, vd);  // att_i multiplies the TARGET row
                vs = fmaf(J.w2[h * J.F + f], w, vs);  // att_j multiplies the SOURCE row
            }
            J.B[(size_t)k * J.ldb + J.a_col + 2 * h] = vs;
            J.B[(size_t)k * J.ldb + J.a_col + 2 * h + 1] = vd;
        }
    } else if (J.kind == PEA_KIND_GCN) {
        for (int idx = tid; idx < J.in * J.HF; idx += 256) {
            const int k = idx / J.HF, o = idx % J.HF;
            J.B[(size_t)k * J.ldb + o] = J.w0[idx];
        }
    } else {
        krows = 2 * J.in;
        for (int idx = tid; idx < J.HF * J.in; idx += 256) {
            const int o = idx / J.in, k = idx % J.in;
            J.B[(size_t)k * J.ldb + o] = J.w0[idx];
            J.B[(size_t)(J.in + k) * J.ldb + o] = J.w1[idx];
        }
    }
    for (int idx = tid; idx < krows * J.zero_n; idx += 256) {
        const int k = idx / J.zero_n, c = idx % J.zero_n;
        J.B[(size_t)k * J.ldb + J.zero_col + c] = 0.f;
    }
    for (int o = tid; o < J.HF; o += 256) J.bias[o] = J.w3 ? J.w3[o] : 0.f;
}

}  // namespace

int launch_gemm(const GemmJob &job, const int *rows, int64_t n_rows, hipStream_t stream) {
    PEA_REQUIRE(job.K1 > 0 && job.K1 % 4 == 0 && job.K2 % 4 == 0, PEA_ERR_ARG,
                "gemm: input widths (%d, %d) must be multiples of 4", job.K1, job.K2);
    PEA_REQUIRE(job.n_out > 0 && job.n_out % 4 == 0 && job.ldb % 4 == 0 && job.ldb >= job.n_out, PEA_ERR_ARG,
                "gemm: output width %d / ldb %d must be multiples of 4", job.n_out, job.ldb);
    PEA_REQUIRE(job.lda1 % 4 == 0 && (job.K2 == 0 || job.lda2 % 4 == 0), PEA_ERR_ARG,
                "gemm: input row strides must be multiples of 4 floats");
    PEA_REQUIRE(job.n_seg > 0 && job.n_seg <= kMaxSegments, PEA_ERR_ARG, "gemm: %d segments", job.n_seg);
    if (n_rows <= 0) return PEA_OK;
    dim3 grid((unsigned)((n_rows + TM - 1) / TM), (unsigned)((job.n_out + TN - 1) / TN));
    ProfScope ps("gemm", stream, 4.0 * (double)n_rows * (job.K1 + job.K2 + job.n_out));
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, stream, job, rows, n_rows);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

int launch_pack(const PackJob *jobs, int n_jobs, hipStream_t stream) {
    for (int base = 0; base < n_jobs; base += 24) {
        PackLaunch L;
        L.n = n_jobs - base < 24 ? n_jobs - base : 24;
        for (int i = 0; i < L.n; ++i) L.j[i] = jobs[base + i];
        ProfScope ps("pack_weights", stream);
        hipLaunchKernelGGL(pack_kernel, dim3(L.n), dim3(256), 0, stream, L);
        PEA_HIP(hipGetLastError());
    }
    return PEA_OK;
}

}  // namespace pea
