// Channel fusion, BPR scoring and the batched evaluator for gfx950.
//   fuse_kernel   graph_recsys_benchmark/models/base.py:193-203  (stack, ablation mask, 'att'/'mean')
//   bpr kernels   models/base.py:208-214 (predict) + models/base.py:46-48 (-sum log sigmoid(pos-neg))
//   rank kernel   solvers.py:85-96 (score 1 + C-1 candidates, rank of the positive, auc, eval loss)
// All reductions have a fixed order (no float atomics): results are bitwise reproducible.
#include <algorithm>
#include <mutex>

#include "common.h"

namespace pea {
namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// G lanes x float4 cover the R columns of one node; 64/G nodes per wave.  Softmax over the P channels
// is taken online in channel order.
// Sharded ranks: workgroups past the `fuse_blocks` that cover the rank's own rows fuse the rows a BPR batch names
// (FuseSelect): the same arithmetic on the same stack row as the table entry, written to sel.out[k] when this rank owns the
// node, zeros otherwise (the loss all-reduce sums these; round 2 ran a separate select launch over the finished table).
template <int G>
__global__ __launch_bounds__(256) void fuse_kernel(int64_t n_rows, const int *__restrict__ rows, int P, int R,
                                                   const float *__restrict__ stack, int64_t ld,
                                                   const ChanCols col_of_channel,
                                                   const float *__restrict__ att, int masked, int mode,
                                                   float *__restrict__ out, float *__restrict__ out_stack,
                                                   unsigned fuse_blocks, const FuseSelect sel, int64_t N) {
    const bool pick = blockIdx.x >= fuse_blocks;
    const int64_t item = (int64_t)(pick ? blockIdx.x - fuse_blocks : blockIdx.x) * (256 / G) + threadIdx.x / G;
    const int sl = threadIdx.x % G;
    bool valid = item < (pick ? sel.n : n_rows);
    int64_t n = 0;
    if (pick) {
        const int64_t id = valid ? sel.ids[item * sel.id_stride] : 0;
        const bool in_range = id >= 0 && id < N;
        if (valid && !in_range && sl == 0) atomicOr(sel.err, 1);
        const bool mine = in_range && (id / sel.tile) % sel.world == sel.rank;
        if (valid && !mine && sl * 4 < R) *reinterpret_cast<float4 *>(sel.out + item * R + sl * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        valid = valid && mine;
        n = valid ? id : 0;
        out_stack = nullptr;
    } else {
        n = valid ? (rows ? rows[item] : item) : 0;
    }
    float *dst_row = pick ? sel.out + item * R : (out ? out + n * R : nullptr);
    const bool active = valid && sl * 4 < R;
    const int c4 = sl * 4 < R ? sl * 4 : 0;
    float m = -3.0e38f, s = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < P; ++p) {
        float4 x = ld4(stack + n * ld + col_of_channel.c[p] + c4);
        if (p == masked) x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (out_stack && active) *reinterpret_cast<float4 *>(out_stack + (n * P + p) * R + c4) = x;
        if (mode == PEA_FUSE_MEAN) {
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            continue;
        }
        const float4 a = ld4(att + p * R + c4);
        float sc = sl * 4 < R ? (x.x * a.x + x.y * a.y) + (x.z * a.z + x.w * a.w) : 0.f;
#pragma unroll
        for (int off = 1; off < G; off <<= 1) sc += __shfl_xor(sc, off);
        const float mn = fmaxf(m, sc);
        const float f = expf(m - mn), w = expf(sc - mn);
        s = s * f + w;
        acc.x = acc.x * f + w * x.x;
        acc.y = acc.y * f + w * x.y;
        acc.z = acc.z * f + w * x.z;
        acc.w = acc.w * f + w * x.w;
        m = mn;
    }
    if (!active || !dst_row) return;
    const float inv = mode == PEA_FUSE_MEAN ? 1.0f / (float)P : 1.0f / s;
    *reinterpret_cast<float4 *>(dst_row + c4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// fc2(relu(fc1([u || i])))  with fc1_w [R, 2R] staged in LDS by the caller
__device__ __forceinline__ float mlp_score(const float *__restrict__ ur, const float *__restrict__ ir, int R,
                                           const float *w1, const float *b1, const float *w2, float b2) {
    float o = 0.f;
    for (int k = 0; k < R; ++k) {
        const float *w = w1 + k * 2 * R;
        float a = 0.f;
        for (int c = 0; c < R; c += 4) {
            const float4 u = ld4(ur + c), wu = ld4(w + c);
            a += (u.x * wu.x + u.y * wu.y) + (u.z * wu.z + u.w * wu.w);
        }
        for (int c = 0; c < R; c += 4) {
            const float4 v = ld4(ir + c), wi = ld4(w + R + c);
            a += (v.x * wi.x + v.y * wi.y) + (v.z * wi.z + v.w * wi.w);
        }
        a += b1[k];
        o = fmaf(fmaxf(a, 0.f), w2[k], o);
    }
    return o + b2;
}

// same arithmetic with both rows held in registers (R = 4*R4 known at compile time)
template <int R4>
__device__ __forceinline__ float mlp_score_reg(const float4 (&u)[R4], const float4 (&v)[R4], const float *w1, const float *b1,
                                               const float *w2, float b2) {
    constexpr int R = 4 * R4;
    float o = 0.f;
    for (int k = 0; k < R; ++k) {
        const float *w = w1 + k * 2 * R;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < R4; ++c) {
            const float4 wu = ld4(w + 4 * c);
            a += (u[c].x * wu.x + u[c].y * wu.y) + (u[c].z * wu.z + u[c].w * wu.w);
        }
#pragma unroll
        for (int c = 0; c < R4; ++c) {
            const float4 wi = ld4(w + R + 4 * c);
            a += (v[c].x * wi.x + v[c].y * wi.y) + (v[c].z * wi.z + v[c].w * wi.w);
        }
        a += b1[k];
        o = fmaf(fmaxf(a, 0.f), w2[k], o);
    }
    return o + b2;
}

template <int R4>
__device__ __forceinline__ void load_row(const float *p, float4 (&r)[R4]) {
#pragma unroll
    for (int c = 0; c < R4; ++c) r[c] = ld4(p + 4 * c);
}

extern __shared__ float smem[];

__device__ __forceinline__ void stage_mlp(int R, const float *fc1_w, const float *fc1_b, const float *fc2_w) {
    for (int i = threadIdx.x; i < 2 * R * R; i += blockDim.x) smem[i] = fc1_w[i];
    for (int i = threadIdx.x; i < R; i += blockDim.x) {
        smem[2 * R * R + i] = fc1_b[i];
        smem[2 * R * R + R + i] = fc2_w[i];
    }
    __syncthreads();
}

__device__ __forceinline__ float log_sigmoid_ref(float d) {
    // the reference takes sigmoid then log in fp32 with no clamp (may give -inf); keep that
    return logf(1.0f / (1.0f + expf(-d)));
}

__global__ __launch_bounds__(256) void bpr_kernel(int64_t B, int R, int64_t N, const float *__restrict__ repr,
                                                  const int64_t *__restrict__ triples, int64_t stride,
                                                  const float *fc1_w, const float *fc1_b, const float *fc2_w,
                                                  const float *fc2_b, float *pos, float *neg, float *block_sums,
                                                  int *err) {
    stage_mlp(R, fc1_w, fc1_b, fc2_w);
    const float *w1 = smem, *b1 = smem + 2 * R * R, *w2 = b1 + R;
    __shared__ float red[256];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float term = 0.f;
    if (b < B) {
        const int64_t u = triples[b * stride], ip = triples[b * stride + 1], in = triples[b * stride + 2];
        if (u < 0 || u >= N || ip < 0 || ip >= N || in < 0 || in >= N) {
            atomicOr(err, 1);
        } else {
            float sp, sn;
            if (R == 16) {  // the reference's repr_dim: all three rows fetched once, up front
                float4 ur[4], pr[4], nr[4];
                load_row<4>(repr + u * R, ur);
                load_row<4>(repr + ip * R, pr);
                load_row<4>(repr + in * R, nr);
                sp = mlp_score_reg<4>(ur, pr, w1, b1, w2, fc2_b[0]);
                sn = mlp_score_reg<4>(ur, nr, w1, b1, w2, fc2_b[0]);
            } else {
                sp = mlp_score(repr + u * R, repr + ip * R, R, w1, b1, w2, fc2_b[0]);
                sn = mlp_score(repr + u * R, repr + in * R, R, w1, b1, w2, fc2_b[0]);
            }
            if (pos) pos[b] = sp;
            if (neg) neg[b] = sn;
            term = log_sigmoid_ref(sp - sn);
        }
    }
    red[threadIdx.x] = term;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}

// An out-of-range triple contributed nothing to its block sum: the loss would be silently too small, so it is poisoned
// with NaN instead (the reference raises at cached_repr[unids]; the host mirror raises IndexError when it next reads the
// flag, see engine.check_pending_errors).
__global__ __launch_bounds__(256) void bpr_final_kernel(int n_blocks, const float *block_sums, const int *err, float *loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) s += block_sums[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = err[0] ? __int_as_float(0x7fc00000) : -red[0];
}

__global__ __launch_bounds__(256) void predict_kernel(int64_t B, int R, int64_t N, const float *__restrict__ repr,
                                                      const int64_t *__restrict__ unids,
                                                      const int64_t *__restrict__ inids, const float *fc1_w,
                                                      const float *fc1_b, const float *fc2_w, const float *fc2_b,
                                                      float *pred, int *err) {
    stage_mlp(R, fc1_w, fc1_b, fc2_w);
    const float *w1 = smem, *b1 = smem + 2 * R * R, *w2 = b1 + R;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t u = unids[b], i = inids[b];
    if (u < 0 || u >= N || i < 0 || i >= N) {
        atomicOr(err, 1);
        return;
    }
    if (R == 16) {
        float4 ur[4], ir[4];
        load_row<4>(repr + u * R, ur);
        load_row<4>(repr + i * R, ir);
        pred[b] = mlp_score_reg<4>(ur, ir, w1, b1, w2, fc2_b[0]);
    } else {
        pred[b] = mlp_score(repr + u * R, repr + i * R, R, w1, b1, w2, fc2_b[0]);
    }
}

// one wave per user: lanes score candidates, then rank / auc / loss by wave reductions
__global__ __launch_bounds__(256) void rank_kernel(int64_t U, int C, int R, int64_t N, const float *__restrict__ repr,
                                                   const int64_t *__restrict__ unids,
                                                   const int64_t *__restrict__ cand, const float *fc1_w,
                                                   const float *fc1_b, const float *fc2_w, const float *fc2_b,
                                                   float *scores, int32_t *rank, float *auc, float *loss, int *err) {
    stage_mlp(R, fc1_w, fc1_b, fc2_w);
    const float *w1 = smem, *b1 = smem + 2 * R * R, *w2 = b1 + R;
    const int lane = threadIdx.x % kWave;
    const int64_t uidx = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    if (uidx >= U) return;
    const int64_t u = unids[uidx];
    if (u < 0 || u >= N) {
        if (lane == 0) atomicOr(err, 1);
        return;
    }
    float pos = 0.f;
    int higher = 0, gt = 0;
    float lsum = 0.f;
    float4 ur[4];
    if (R == 16) load_row<4>(repr + u * R, ur);   // the reference's repr_dim: the user's row stays in registers
    for (int base = 0; base < C; base += kWave) {
        const int c = base + lane;
        float sc = 0.f;
        bool ok = c < C;
        if (ok) {
            const int64_t i = cand[uidx * C + c];
            if (i < 0 || i >= N) {
                atomicOr(err, 1);
                ok = false;
            } else {
                if (R == 16) {   // same arithmetic, same order as mlp_score (rows held in registers)
                    float4 ir[4];
                    load_row<4>(repr + i * R, ir);
                    sc = mlp_score_reg<4>(ur, ir, w1, b1, w2, fc2_b[0]);
                } else {
                    sc = mlp_score(repr + u * R, repr + i * R, R, w1, b1, w2, fc2_b[0]);
                }
                if (scores) scores[uidx * C + c] = sc;
            }
        }
        if (base == 0) pos = __shfl(sc, 0);
        if (ok && c > 0) {
            // torch.sort(descending) places a negative ahead of the positive only if it scores strictly
            // higher (stable order keeps index 0 first among ties)
            higher += sc > pos ? 1 : 0;
            gt += pos > sc ? 1 : 0;
            lsum += log_sigmoid_ref(pos - sc);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        higher += __shfl_xor(higher, off);
        gt += __shfl_xor(gt, off);
        lsum += __shfl_xor(lsum, off);
    }
    if (lane == 0) {
        if (rank) rank[uidx] = higher;
        if (auc) auc[uidx] = (float)gt / (float)(C - 1);
        if (loss) loss[uidx] = -lsum;
    }
}

int lanes_for_r(int R) {
    int g = 1;
    while (g * 4 < R) g <<= 1;
    return g;
}

}  // namespace

int launch_fuse(int64_t N, int P, int R, const float *stack, int64_t ld, const ChanCols &col_of_channel,
                const float *att, int masked, int mode, const int *rows, int64_t n_rows, float *out,
                float *out_stack, hipStream_t stream, const FuseSelect *sel_in) {
    PEA_REQUIRE(P > 0 && P <= kMaxChannels && R > 0 && R % 4 == 0 && R <= 256, PEA_ERR_ARG,
                "fuse: P=%d R=%d (P <= 64, R a multiple of 4 and <= 256)", P, R);
    PEA_REQUIRE(masked < P, PEA_ERR_ARG, "fuse: masked channel %d out of range (P=%d)", masked, P);
    PEA_REQUIRE(mode == PEA_FUSE_MEAN || att != nullptr, PEA_ERR_ARG, "fuse: att is required for 'att' fusion");
    PEA_REQUIRE(ld % 4 == 0, PEA_ERR_ARG, "fuse: stack row stride must be a multiple of 4");
    if (!rows) n_rows = N;
    FuseSelect sel;
    if (sel_in && sel_in->n > 0) {
        sel = *sel_in;
        PEA_REQUIRE(sel.ids && sel.out && sel.err && sel.world >= 1 && sel.tile >= 1 && out != nullptr, PEA_ERR_ARG,
                    "fuse: batch-row selection needs ids, an output, an error flag and the fused table");
    }
    if (n_rows <= 0 && sel.n <= 0) return PEA_OK;
    const int G = lanes_for_r(R);
    ProfScope ps("fuse", stream, 4.0 * (double)(n_rows + sel.n) * R * (P + 1));
    const unsigned fuse_blocks = (unsigned)((std::max<int64_t>(n_rows, 0) + (256 / G) - 1) / (256 / G));
    const unsigned blocks = fuse_blocks + (unsigned)((sel.n + (256 / G) - 1) / (256 / G));
#define PEA_FUSE_CASE(g)                                                                                       \
    case g:                                                                                                    \
        PEA_LAUNCH(fuse_kernel<g>, dim3(blocks), dim3(256), 0, stream, n_rows, rows, P, R, stack, ld, \
                           col_of_channel, att, masked, mode, out, out_stack, fuse_blocks, sel, N);            \
        break;
    switch (G) {
        PEA_FUSE_CASE(1)
        PEA_FUSE_CASE(2)
        PEA_FUSE_CASE(4)
        PEA_FUSE_CASE(8)
        PEA_FUSE_CASE(16)
        PEA_FUSE_CASE(32)
        default:
            PEA_LAUNCH(fuse_kernel<64>, dim3(blocks), dim3(256), 0, stream, n_rows, rows, P, R, stack, ld,
                               col_of_channel, att, masked, mode, out, out_stack, fuse_blocks, sel, N);
    }
#undef PEA_FUSE_CASE
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

}  // namespace pea

// ---------------------------------------------------------------------------------------------- C ABI
using pea::set_error;

extern "C" size_t pea_bpr_workspace_bytes(int64_t B) {
    const int64_t blocks = (B + 255) / 256;
    return (size_t)(blocks + 1) * sizeof(float) + 16;
}

static int check_r(int R) {
    PEA_REQUIRE(R > 0 && R % 4 == 0 && R <= 64, PEA_ERR_ARG, "repr_dim %d must be a multiple of 4, <= 64", R);
    return PEA_OK;
}

// One 4-byte error flag per device, allocated on first use and kept for the life of the process (pea_predict /
// pea_rank_eval used to hipMalloc + hipFree one per call).  The callers are the reference's single Python thread; two
// host threads scoring at once on one device would share the flag (an error is then reported to at least one of them).
static int *err_flag_for_current_device() {
    static std::mutex mu;
    static int *flags[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!flags[dev] && hipMalloc(&flags[dev], 256) != hipSuccess) flags[dev] = nullptr;
    return flags[dev];
}

static int read_err_flag(int *err_dev, hipStream_t stream, const char *what) {
    int h = 0;
    PEA_HIP(hipMemcpyAsync(&h, err_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    PEA_HIP(hipStreamSynchronize(stream));
    PEA_REQUIRE(h == 0, PEA_ERR_RANGE, "%s: node id outside [0, num_nodes)", what);
    return PEA_OK;
}

extern "C" int pea_bpr_score(int64_t B, int R, int64_t num_nodes, const float *repr, const int64_t *triples,
                             int64_t triple_stride, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                             const float *fc2_b, float *pos, float *neg, float *loss, void *workspace,
                             size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_TRY(check_r(R));
    PEA_REQUIRE(B >= 0 && triple_stride >= 3, PEA_ERR_ARG, "bpr: B=%lld stride=%lld", (long long)B, (long long)triple_stride);
    PEA_REQUIRE(repr && triples && fc1_w && fc1_b && fc2_w && fc2_b && loss && workspace, PEA_ERR_ARG, "bpr: null pointer");
    PEA_REQUIRE(workspace_bytes >= pea_bpr_workspace_bytes(B), PEA_ERR_NOMEM, "bpr: workspace too small");
    const int blocks = (int)((B + 255) / 256);
    int *err = (int *)workspace;
    float *sums = (float *)workspace + 4;
    PEA_MEMSET_ASYNC(err, 0, sizeof(int), stream);
    const size_t sh = (size_t)(2 * R * R + 2 * R) * sizeof(float);
    pea::ProfScope ps("bpr_score", stream, (double)B * (12.0 + 12.0 * R + 4.0));
    if (blocks > 0) {
        PEA_LAUNCH(pea::bpr_kernel, dim3(blocks), dim3(256), sh, stream, B, R, num_nodes, repr, triples,
                           triple_stride, fc1_w, fc1_b, fc2_w, fc2_b, pos, neg, sums, err);
        PEA_HIP(hipGetLastError());
    }
    PEA_LAUNCH(pea::bpr_final_kernel, dim3(1), dim3(256), 0, stream, blocks, sums, err, loss);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

extern "C" int pea_predict(int64_t B, int R, int64_t num_nodes, const float *repr, const int64_t *unids,
                           const int64_t *inids, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                           const float *fc2_b, float *pred, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_TRY(check_r(R));
    PEA_REQUIRE(B >= 0 && repr && unids && inids && fc1_w && fc1_b && fc2_w && fc2_b && pred, PEA_ERR_ARG, "predict: bad argument");
    if (B == 0) return PEA_OK;
    int *err = err_flag_for_current_device();
    PEA_REQUIRE(err != nullptr, PEA_ERR_HIP, "predict: no error-flag buffer on this device");
    PEA_MEMSET_ASYNC(err, 0, sizeof(int), stream);
    const size_t sh = (size_t)(2 * R * R + 2 * R) * sizeof(float);
    PEA_LAUNCH(pea::predict_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), sh, stream, B, R, num_nodes,
                       repr, unids, inids, fc1_w, fc1_b, fc2_w, fc2_b, pred, err);
    int rc = hipGetLastError() == hipSuccess ? PEA_OK : PEA_ERR_HIP;
    if (rc == PEA_OK) rc = read_err_flag(err, stream, "predict");
    return rc;
}

extern "C" int pea_rank_eval(int64_t U, int C, int R, int64_t num_nodes, const float *repr, const int64_t *unids,
                             const int64_t *cand, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                             const float *fc2_b, float *scores, int32_t *rank, float *auc, float *loss,
                             void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_TRY(check_r(R));
    PEA_REQUIRE(U >= 0 && C >= 2 && repr && unids && cand && fc1_w && fc1_b && fc2_w && fc2_b, PEA_ERR_ARG, "rank_eval: bad argument");
    if (U == 0) return PEA_OK;
    int *err = err_flag_for_current_device();
    PEA_REQUIRE(err != nullptr, PEA_ERR_HIP, "rank_eval: no error-flag buffer on this device");
    PEA_MEMSET_ASYNC(err, 0, sizeof(int), stream);
    const size_t sh = (size_t)(2 * R * R + 2 * R) * sizeof(float);
    PEA_LAUNCH(pea::rank_kernel, dim3((unsigned)((U + 3) / 4)), dim3(256), sh, stream, U, C, R, num_nodes, repr,
                       unids, cand, fc1_w, fc1_b, fc2_w, fc2_b, scores, rank, auc, loss, err);
    int rc = hipGetLastError() == hipSuccess ? PEA_OK : PEA_ERR_HIP;
    if (rc == PEA_OK) rc = read_err_flag(err, stream, "rank_eval");
    return rc;
}

extern "C" int pea_fuse(int64_t num_nodes, int P, int R, const float *stack, int64_t ld_stack,
                        const int *col_of_channel_host, const float *att, int masked_channel, int fuse_mode, float *out,
                        void *stream) {
    PEA_REQUIRE(num_nodes > 0 && stack && out && col_of_channel_host, PEA_ERR_ARG, "fuse: null argument");
    PEA_REQUIRE(P > 0 && P <= pea::kMaxChannels, PEA_ERR_ARG, "fuse: P=%d (1..%d)", P, pea::kMaxChannels);
    PEA_REQUIRE(fuse_mode == PEA_FUSE_ATT || fuse_mode == PEA_FUSE_MEAN, PEA_ERR_ARG, "fuse: mode %d", fuse_mode);
    PEA_REQUIRE(masked_channel >= -1, PEA_ERR_ARG, "fuse: masked channel %d", masked_channel);
    pea::ChanCols cols{};
    for (int p = 0; p < P; ++p) {
        PEA_REQUIRE(col_of_channel_host[p] >= 0 && col_of_channel_host[p] % 4 == 0 && col_of_channel_host[p] + R <= ld_stack,
                    PEA_ERR_ARG, "fuse: channel %d column %d outside the stack row", p, col_of_channel_host[p]);
        cols.c[p] = col_of_channel_host[p];
    }
    return pea::launch_fuse(num_nodes, P, R, stack, ld_stack, cols, att, masked_channel, fuse_mode, nullptr, num_nodes, out,
                            nullptr, (hipStream_t)stream);
}
