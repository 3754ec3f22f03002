// Training-step head of the PEA models, forward AND backward in one launch: for the B triples of a BPR batch
//   fused rows   models/base.py:193-203   softmax_p(sum_c S[p, c] att[p, c]) weighted sum of the P channel rows ('att') or mean
//   scores       models/base.py:208-214   fc2(relu(fc1([user || item])))  for (user, pos) and (user, neg)
//   loss         models/base.py:46-48     -sum_b log sigmoid(pos_b - neg_b)
// and the gradient of that loss with respect to the 3B stack rows it read, plus the operands from which the (tiny)
// parameter gradients follow as two fixed-order reductions on the host side (pea_grad_weight):
//   dhx [2B, R + 4]   = [d hidden | d score | 0 0 0]            row 2b = (user, pos), row 2b + 1 = (user, neg)
//   zx  [2B, 3R + 4]  = [user row | item row | relu(hidden) | 1 0 0 0]
//       dhx^T zx  ->  rows 0..R-1: [d fc1.weight (R x 2R) | . | d fc1.bias],  row R: [. | d fc2.weight (R) | d fc2.bias]
//   dsc [3B, P4]      = d (channel score) per stack row:  d att[p] = sum_rows dsc[row, p] * S[row, p, :]
// The reference runs ~40 elementwise / index kernels and 6 library GEMM calls for this under autograd
// (solvers.py:213-214 loss.backward()); here one thread owns a triple: its three rows stay in registers between the
// forward and the backward, the fc weights and the attention vectors sit in LDS.  No atomics: the loss is summed per
// block in thread order and the blocks in index order.
#include <algorithm>

#include "common.h"

namespace pea {
namespace {

constexpr int kTB = 64;   // triples per workgroup (B = 4096 -> 64 workgroups)

extern __shared__ float tsm[];

__device__ __forceinline__ float4 ld4t(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// channel score of one row chunk list, summed in the order fuse_kernel sums it (per float4 (xy) + (zw), then a pairwise
// tree over the G = pow2 >= R4 lane slots of that kernel)
template <int R4>
__device__ __forceinline__ float chan_score(const float4 (&x)[R4], const float *a) {
    constexpr int G = R4 <= 1 ? 1 : R4 <= 2 ? 2 : R4 <= 4 ? 4 : R4 <= 8 ? 8 : 16;
    float q[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        if (i < R4) {
            const float4 w = ld4t(a + 4 * i);
            q[i] = (x[i].x * w.x + x[i].y * w.y) + (x[i].z * w.z + x[i].w * w.w);
        } else {
            q[i] = 0.f;
        }
    }
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        float t[G];
#pragma unroll
        for (int i = 0; i < G; ++i) t[i] = q[i] + q[i ^ off];
#pragma unroll
        for (int i = 0; i < G; ++i) q[i] = t[i];
    }
    return q[0];
}

template <int R4>
__global__ __launch_bounds__(kTB) void bpr_train_kernel(int64_t B, int P, const float *__restrict__ rows, int64_t ld,
                                                        const float *__restrict__ att, const float *fc1_w,
                                                        const float *fc1_b, const float *fc2_w, const float *fc2_b,
                                                        float *__restrict__ grad_rows, float *__restrict__ dhx,
                                                        float *__restrict__ zx, float *__restrict__ dsc, int P4,
                                                        float *block_sums) {
    constexpr int R = 4 * R4;
    float *w1 = tsm, *b1 = tsm + 2 * R * R, *w2 = b1 + R, *hid = w2 + R;   // hid: [2 R][kTB] hidden units, a column per thread
    float *av = hid + 2 * R * kTB;                                          // av: [P, R] attention vectors ('att' mode)
    for (int i = threadIdx.x; i < 2 * R * R; i += kTB) w1[i] = fc1_w[i];
    for (int i = threadIdx.x; i < R; i += kTB) {
        b1[i] = fc1_b[i];
        w2[i] = fc2_w[i];
    }
    if (att)
        for (int i = threadIdx.x; i < P * R; i += kTB) av[i] = att[i];
    __syncthreads();
    __shared__ float red[kTB];
    const int64_t b = (int64_t)blockIdx.x * kTB + threadIdx.x;
    float term = 0.f;
    if (b < B) {
        // ---- forward: the three fused rows (same online softmax over the channels as fuse_kernel)
        float4 f[3][R4];
        float fm[3], fs[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float *row = rows + (3 * b + j) * ld;
            float m = -3.0e38f, s = 0.f;
            float4 acc[R4];
#pragma unroll
            for (int c = 0; c < R4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = 0; p < P; ++p) {
                float4 x[R4];
#pragma unroll
                for (int c = 0; c < R4; ++c) x[c] = ld4t(row + p * R + 4 * c);
                if (!att) {
#pragma unroll
                    for (int c = 0; c < R4; ++c) {
                        acc[c].x += x[c].x; acc[c].y += x[c].y; acc[c].z += x[c].z; acc[c].w += x[c].w;
                    }
                    continue;
                }
                const float sc = chan_score<R4>(x, av + p * R);
                const float mn = fmaxf(m, sc);
                const float fo = expf(m - mn), w = expf(sc - mn);
                s = s * fo + w;
#pragma unroll
                for (int c = 0; c < R4; ++c) {
                    acc[c].x = acc[c].x * fo + w * x[c].x;
                    acc[c].y = acc[c].y * fo + w * x[c].y;
                    acc[c].z = acc[c].z * fo + w * x[c].z;
                    acc[c].w = acc[c].w * fo + w * x[c].w;
                }
                m = mn;
            }
            const float inv = att ? 1.0f / s : 1.0f / (float)P;
#pragma unroll
            for (int c = 0; c < R4; ++c) f[j][c] = make_float4(acc[c].x * inv, acc[c].y * inv, acc[c].z * inv, acc[c].w * inv);
            fm[j] = m;
            fs[j] = s;
        }
        // ---- forward: the two scores (the arithmetic order of mlp_score_reg in fuse_score.hip).  The hidden vectors live
        //      in LDS (one column per thread) so that the loops over the hidden units stay loops: fully unrolled, the
        //      compiler kept the whole fc1 image in registers and spilled kilobytes per lane
        float sc2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float o = 0.f;
#pragma unroll 1
            for (int k = 0; k < R; ++k) {
                const float *w = w1 + k * 2 * R;
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < R4; ++c) {
                    const float4 wu = ld4t(w + 4 * c);
                    a += (f[0][c].x * wu.x + f[0][c].y * wu.y) + (f[0][c].z * wu.z + f[0][c].w * wu.w);
                }
#pragma unroll
                for (int c = 0; c < R4; ++c) {
                    const float4 wi = ld4t(w + R + 4 * c);
                    a += (f[1 + e][c].x * wi.x + f[1 + e][c].y * wi.y) + (f[1 + e][c].z * wi.z + f[1 + e][c].w * wi.w);
                }
                a += b1[k];
                hid[(e * R + k) * kTB + threadIdx.x] = a;
                o = fmaf(fmaxf(a, 0.f), w2[k], o);
            }
            sc2[e] = o + fc2_b[0];
        }
        const float d = sc2[0] - sc2[1];
        const float sig = 1.0f / (1.0f + expf(-d));
        term = logf(sig);                       // sigmoid then log in fp32, no clamp: the reference's own formula
        const float g = -(1.0f - sig);          // d loss / d pos = -(1 - sigmoid(d)),  d loss / d neg = +(1 - sigmoid(d))
        // ---- backward through the scorer; the operands of the parameter gradients go out as rows of dhx / zx
        float4 df[3][R4];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < R4; ++c) df[j][c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float ds = e == 0 ? g : -g;
            float *dh_row = dhx + (2 * b + e) * (R + 4);
            float *z_row = zx + (2 * b + e) * (3 * R + 4);
#pragma unroll
            for (int c = 0; c < R4; ++c) {
                *reinterpret_cast<float4 *>(z_row + 4 * c) = f[0][c];
                *reinterpret_cast<float4 *>(z_row + R + 4 * c) = f[1 + e][c];
            }
#pragma unroll 1
            for (int k = 0; k < R; ++k) {
                const float hv = hid[(e * R + k) * kTB + threadIdx.x];
                const float dh = hv > 0.f ? ds * w2[k] : 0.f;
                dh_row[k] = dh;
                z_row[2 * R + k] = fmaxf(hv, 0.f);
                const float *w = w1 + k * 2 * R;
#pragma unroll
                for (int c = 0; c < R4; ++c) {
                    const float4 wu = ld4t(w + 4 * c), wi = ld4t(w + R + 4 * c);
                    df[0][c].x = fmaf(dh, wu.x, df[0][c].x);
                    df[0][c].y = fmaf(dh, wu.y, df[0][c].y);
                    df[0][c].z = fmaf(dh, wu.z, df[0][c].z);
                    df[0][c].w = fmaf(dh, wu.w, df[0][c].w);
                    df[1 + e][c].x = fmaf(dh, wi.x, df[1 + e][c].x);
                    df[1 + e][c].y = fmaf(dh, wi.y, df[1 + e][c].y);
                    df[1 + e][c].z = fmaf(dh, wi.z, df[1 + e][c].z);
                    df[1 + e][c].w = fmaf(dh, wi.w, df[1 + e][c].w);
                }
            }
            *reinterpret_cast<float4 *>(dh_row + R) = make_float4(ds, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(z_row + 3 * R) = make_float4(1.f, 0.f, 0.f, 0.f);
        }
        // ---- backward through the fusion: fused = sum_p a_p S_p, a = softmax(sc), sc_p = S_p . att_p
        //      d S_p = a_p dF + dsc_p att_p,   dsc_p = a_p (dF . S_p - sum_q a_q dF . S_q)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float *row = rows + (3 * b + j) * ld;
            float *grow = grad_rows + (3 * b + j) * (int64_t)P * R;
            if (!att) {
                const float ip = 1.0f / (float)P;
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int c = 0; c < R4; ++c)
                        *reinterpret_cast<float4 *>(grow + p * R + 4 * c) =
                            make_float4(df[j][c].x * ip, df[j][c].y * ip, df[j][c].z * ip, df[j][c].w * ip);
                continue;
            }
            const float inv = 1.0f / fs[j];
            float t = 0.f;
            for (int p = 0; p < P; ++p) {
                float4 x[R4];
#pragma unroll
                for (int c = 0; c < R4; ++c) x[c] = ld4t(row + p * R + 4 * c);
                const float a = expf(chan_score<R4>(x, av + p * R) - fm[j]) * inv;
                float da = 0.f;
#pragma unroll
                for (int c = 0; c < R4; ++c)
                    da += (df[j][c].x * x[c].x + df[j][c].y * x[c].y) + (df[j][c].z * x[c].z + df[j][c].w * x[c].w);
                t = fmaf(a, da, t);
            }
            float *drow = dsc + (3 * b + j) * (int64_t)P4;
            for (int p = 0; p < P; ++p) {
                float4 x[R4];
#pragma unroll
                for (int c = 0; c < R4; ++c) x[c] = ld4t(row + p * R + 4 * c);
                const float a = expf(chan_score<R4>(x, av + p * R) - fm[j]) * inv;
                float da = 0.f;
#pragma unroll
                for (int c = 0; c < R4; ++c)
                    da += (df[j][c].x * x[c].x + df[j][c].y * x[c].y) + (df[j][c].z * x[c].z + df[j][c].w * x[c].w);
                const float ds_p = a * (da - t);
                drow[p] = ds_p;
#pragma unroll
                for (int c = 0; c < R4; ++c) {
                    const float4 w = ld4t(av + p * R + 4 * c);
                    *reinterpret_cast<float4 *>(grow + p * R + 4 * c) =
                        make_float4(fmaf(a, df[j][c].x, ds_p * w.x), fmaf(a, df[j][c].y, ds_p * w.y),
                                    fmaf(a, df[j][c].z, ds_p * w.z), fmaf(a, df[j][c].w, ds_p * w.w));
                }
            }
            for (int p = P; p < P4; ++p) drow[p] = 0.f;
        }
    }
    red[threadIdx.x] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int k = 0; k < kTB; ++k) a += red[k];   // triple order inside the block
        block_sums[blockIdx.x] = a;
    }
}

__global__ __launch_bounds__(256) void bpr_train_final_kernel(int n_blocks, const float *block_sums, float *loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) s += block_sums[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = -red[0];
}

// dst[id, col[p] + c] = sum over the positions k with ids[k] == id of src[k, p * R + c], added in increasing k: the
// scatter of the batch's gradient rows into the node-indexed output-gradient buffer (autograd's index backward: torch
// sorts the ids for this with ~25 launches for 12 k ids).  Two launches: (1) ONE workgroup groups the (id, position) keys
// of the whole batch in LDS (<= 16384 keys: 128 KB), so the positions of a node become one run ordered by position;
// (2) one wave per run adds the run's rows in that order, NB rows in flight, and writes the node's row.  Same additions in the same order as a sequential loop over the batch: bitwise reproducible,
// no atomics.  ids < 0 sort to the end and are skipped.
constexpr int kSortThreads = 1024;
constexpr int kSortMax = 16384;
constexpr int kBuckets = 2048;

// Groups the batch's (id, position) keys: bucket = id % 2048 (LDS histogram + prefix sum; integer atomics only decide
// scratch slots, never the result), then every key finds its rank inside its bucket by counting the smaller keys there
// (buckets hold a handful of keys; a node named 500 times makes 500 threads read 500 keys each).  Output: the keys ordered
// by (bucket, id, position): the positions of one node are one run, in position order.  Keys of ids < 0 go last (~0).
__global__ __launch_bounds__(kSortThreads) void sort_ids_kernel(int n, int64_t num_rows, const int64_t *__restrict__ ids,
                                                                unsigned long long *__restrict__ sorted) {
    unsigned long long *tmp = reinterpret_cast<unsigned long long *>(tsm);     // [n] keys in bucket order, unordered inside
    __shared__ int start[kBuckets + 1], cursor[kBuckets];
    __shared__ int wave_tot[kSortThreads / 64];
    __shared__ int n_valid;
    for (int b = threadIdx.x; b < kBuckets; b += kSortThreads) cursor[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kSortThreads) {
        const int64_t id = ids[i];
        if (id >= 0 && id < num_rows) atomicAdd(&cursor[(int)(id % kBuckets)], 1);   // ids outside the table are skipped like negative ones
    }
    __syncthreads();
    {   // exclusive prefix sum of the 2048 counts: 2 per thread, wave scan, then the 16 wave totals
        const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
        const int c0 = cursor[2 * t], c1 = cursor[2 * t + 1];
        int v = c0 + c1;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(v, off);
            if (lane >= off) v += o;
        }
        if (lane == 63) wave_tot[wave] = v;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wave_tot[w];
        const int excl = base + v - (c0 + c1);
        start[2 * t] = excl;
        start[2 * t + 1] = excl + c0;
        if (t == kSortThreads - 1) {
            start[kBuckets] = excl + c0 + c1;
            n_valid = excl + c0 + c1;
        }
        __syncthreads();
        cursor[2 * t] = 0;
        cursor[2 * t + 1] = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kSortThreads) {
        const int64_t id = ids[i];
        if (id < 0 || id >= num_rows) continue;
        const int b = (int)(id % kBuckets);
        tmp[start[b] + atomicAdd(&cursor[b], 1)] = ((unsigned long long)id << 32) | (unsigned)i;
    }
    __syncthreads();
    const int nv = n_valid;
    for (int j = threadIdx.x; j < nv; j += kSortThreads) {
        const unsigned long long key = tmp[j];
        const int b = (int)((key >> 32) % kBuckets);
        int rank = 0;
        for (int r = start[b]; r < start[b + 1]; ++r) rank += tmp[r] < key;
        sorted[start[b] + rank] = key;
    }
    for (int j = nv + threadIdx.x; j < n; j += kSortThreads) sorted[j] = ~0ull;
}

template <int V>   // float4 columns per lane: 1 for rows of <= 256 floats, else 4
__global__ __launch_bounds__(256) void scatter_runs_kernel(int n, const unsigned long long *__restrict__ sorted,
                                                           const float *__restrict__ src, int64_t ld_src, int W, int R,
                                                           const ChanCols cols, float *__restrict__ dst, int64_t ld_dst) {
    constexpr int NB = 32 / V;
    const int lane = threadIdx.x & 63;
    const int i0 = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i0 >= n) return;
    const unsigned long long k0 = sorted[i0];
    if (k0 == ~0ull) return;                                          // skipped ids (and everything after them)
    const unsigned id = (unsigned)(k0 >> 32);
    if (i0 > 0 && (unsigned)(sorted[i0 - 1] >> 32) == id) return;     // not the first position of this node's run
    float4 acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = i0; i < n; i += NB) {
        // the next NB sorted keys (one per lane), how many of them continue the run
        const unsigned long long kk = (lane < NB && i + lane < n) ? sorted[i + lane] : ~0ull;
        const unsigned long long same = __ballot(lane < NB && kk != ~0ull && (unsigned)(kk >> 32) == id);
        const int cnt = __builtin_ctzll(~same);                        // leading run of set bits (lane 0 upwards)
        float4 t[NB][V];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const unsigned pos = (unsigned)__shfl(kk, u < cnt ? u : 0);   // a slot past the run re-reads its first row, weight 0
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int c4 = 4 * (lane + 64 * v);
                t[u][v] = c4 < W ? ld4t(src + (int64_t)pos * ld_src + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const float f = u < cnt ? 1.f : 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v)
                acc[v] = make_float4(fmaf(f, t[u][v].x, acc[v].x), fmaf(f, t[u][v].y, acc[v].y), fmaf(f, t[u][v].z, acc[v].z),
                                     fmaf(f, t[u][v].w, acc[v].w));
        }
        if (cnt < NB) break;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int c4 = 4 * (lane + 64 * v);
        if (c4 < W) *reinterpret_cast<float4 *>(dst + (int64_t)id * ld_dst + cols.c[c4 / R] + c4 % R) = acc[v];
    }
}

}  // namespace
}  // namespace pea

using namespace pea;

extern "C" size_t pea_rows_scatter_sum_workspace_bytes(int64_t n) { return n < 0 ? 0 : (size_t)n * 8 + 256; }

extern "C" int pea_rows_scatter_sum(int64_t n, const int64_t *ids, const float *src, int64_t ld_src, int P, int R,
                                    const int *col_of_channel_host, float *dst, int64_t ld_dst, int64_t num_rows,
                                    void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(n >= 0 && n <= kSortMax, PEA_ERR_ARG, "rows_scatter_sum: %lld positions (the batch is sorted in LDS: <= %d)",
                (long long)n, kSortMax);
    PEA_REQUIRE(P > 0 && P <= kMaxChannels && R > 0 && R % 4 == 0 && P * R <= 1024, PEA_ERR_ARG,
                "rows_scatter_sum: P=%d R=%d (R a multiple of 4, P * R <= 1024)", P, R);
    PEA_REQUIRE(ids && src && dst && col_of_channel_host && ld_src >= (int64_t)P * R && ld_src % 4 == 0 && ld_dst % 4 == 0 &&
                    num_rows >= 0 && num_rows < ((int64_t)1 << 31),
                PEA_ERR_ARG, "rows_scatter_sum: bad argument");
    PEA_REQUIRE(workspace && workspace_bytes >= pea_rows_scatter_sum_workspace_bytes(n), PEA_ERR_NOMEM, "rows_scatter_sum: workspace too small");
    if (n == 0) return PEA_OK;
    ChanCols cols;
    for (int p = 0; p < P; ++p) {
        PEA_REQUIRE(col_of_channel_host[p] >= 0 && col_of_channel_host[p] % 4 == 0, PEA_ERR_ARG, "rows_scatter_sum: column %d", col_of_channel_host[p]);
        cols.c[p] = col_of_channel_host[p];
    }
    unsigned long long *sorted = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
    static bool attr_set = false;
    if (!attr_set) {
        PEA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_ids_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kSortMax * 8));
        attr_set = true;
    }
    ProfScope ps("scatter_sum", stream, (double)n * P * R * 8.0);
    PEA_LAUNCH(sort_ids_kernel, dim3(1), dim3(kSortThreads), (size_t)n * 8, stream, (int)n, num_rows, ids, sorted);
    PEA_HIP(hipGetLastError());
    const int blocks = (int)((n + 3) / 4);
    if (P * R > 256) {
        PEA_LAUNCH(scatter_runs_kernel<4>, dim3(blocks), dim3(256), 0, stream, (int)n, sorted, src, ld_src, P * R, R, cols, dst, ld_dst);
    } else {
        PEA_LAUNCH(scatter_runs_kernel<1>, dim3(blocks), dim3(256), 0, stream, (int)n, sorted, src, ld_src, P * R, R, cols, dst, ld_dst);
    }
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

extern "C" size_t pea_bpr_train_workspace_bytes(int64_t B) {
    return B < 0 ? 0 : (size_t)((B + kTB - 1) / kTB + 4) * sizeof(float);
}

extern "C" int pea_bpr_train_supported(int P, int R) {
    const int r4 = R / 4;
    return P > 0 && P <= kMaxChannels && R > 0 && R % 4 == 0 && r4 <= 8 ? 1 : 0;
}

extern "C" int pea_bpr_train(int64_t B, int P, int R, const float *rows, int64_t ld_rows, const float *att,
                             const float *fc1_w, const float *fc1_b, const float *fc2_w, const float *fc2_b,
                             float *out_loss, float *grad_rows, float *dhx, float *zx, float *dsc, void *workspace,
                             size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(B >= 0 && pea_bpr_train_supported(P, R), PEA_ERR_ARG,
                "bpr_train: B=%lld P=%d R=%d (repr_dim a multiple of 4, <= 32)", (long long)B, P, R);
    PEA_REQUIRE(rows && fc1_w && fc1_b && fc2_w && fc2_b && out_loss && grad_rows && dhx && zx && workspace, PEA_ERR_ARG,
                "bpr_train: null pointer");
    PEA_REQUIRE(att == nullptr || dsc != nullptr, PEA_ERR_ARG, "bpr_train: 'att' fusion needs the dsc buffer");
    PEA_REQUIRE(ld_rows >= (int64_t)P * R && ld_rows % 4 == 0, PEA_ERR_ARG, "bpr_train: row stride %lld", (long long)ld_rows);
    PEA_REQUIRE(workspace_bytes >= pea_bpr_train_workspace_bytes(B), PEA_ERR_NOMEM, "bpr_train: workspace too small");
    float *sums = (float *)workspace;
    const int blocks = (int)((B + kTB - 1) / kTB);
    const int P4 = (P + 3) / 4 * 4;
    const size_t sh = (size_t)(2 * R * R + 2 * R + 2 * R * kTB + (att ? P * R : 0)) * sizeof(float);
    if (blocks > 0) {
        ProfScope ps("bpr_train", stream, (double)B * 3.0 * P * R * 8.0);
#define PEA_BT_CASE(r4)                                                                                               \
    case r4:                                                                                                          \
        PEA_LAUNCH(bpr_train_kernel<r4>, dim3(blocks), dim3(kTB), sh, stream, B, P, rows, ld_rows, att, fc1_w, \
                           fc1_b, fc2_w, fc2_b, grad_rows, dhx, zx, dsc, P4, sums);                                   \
        break;
        switch (R / 4) {
            PEA_BT_CASE(1)
            PEA_BT_CASE(2)
            PEA_BT_CASE(3)
            PEA_BT_CASE(4)
            PEA_BT_CASE(5)
            PEA_BT_CASE(6)
            PEA_BT_CASE(7)
            default:
                PEA_LAUNCH(bpr_train_kernel<8>, dim3(blocks), dim3(kTB), sh, stream, B, P, rows, ld_rows, att,
                                   fc1_w, fc1_b, fc2_w, fc2_b, grad_rows, dhx, zx, dsc, P4, sums);
        }
#undef PEA_BT_CASE
        PEA_HIP(hipGetLastError());
    }
    PEA_LAUNCH(bpr_train_final_kernel, dim3(1), dim3(256), 0, stream, blocks, sums, out_loss);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}
