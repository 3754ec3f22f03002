// Two-step inference schedule, dense half: for every node n and every 2-step channel c
//     T_1[n, c] = relu( in_c(n) . W0_c + b0_c ) . W1_c            in_c(n) = A_0[n, c]  if n has incoming edges under the
//                                                                            channel's first relation (the aggregate the
//                                                                            first-layer gather wrote), else s_c(n) * x[n]
// i.e. the first conv layer's transform (reference models/base.py:138: conv_0 = GATConv / GCNConv.lin, then F.relu) applied
// AFTER the neighbour aggregation -- sum_j alpha_ij (W x_j) = W sum_j alpha_ij x_j, the reassociation SURVEY.md 8(a) A5
// notes for SAGE, here for GAT (logits from x . (W^T att), mlp2_pack_kernel) and GCN -- chained with the second layer's
// transform in ONE kernel: the [N, P * hidden] tables T_0 / O_0 of the level-wise schedule (630 MB each on the 25m-shaped
// graph, written once and read once) are never materialised, x is the only gather source of the first layer, and a
// sharded rank transforms exactly the rows it owns (x is replicated: no source-row redundancy).
//
// Per wave: 32 rows.  GEMM 1 (hidden^T = W0^T x^T) keeps the data row on the lane: v_mfma_f32_32x32x2_f32 with the weights
// as the A operand and the row's inputs as B, so the accumulator tile [hidden unit][row] has its column (the data row) on
// the lane and feeds GEMM 2 (out^T = W1^T hidden^T) as the B operand with no lane movement and no LDS (guide: "an
// accumulator tile as the next MFMA's operand"): step (t, v) of GEMM 2 takes register v of tile t, bias + relu applied on
// the way, against W1 rows laid out in the k order the accumulator registers carry (unit 32 t + (v & 3) + 8 (v >> 2) + 4 half).
// fp32 products, fp32 accumulation (f32-input MFMA = an fmaf chain in k order): exact fp32 like the level-wise kernels.
#include "common.h"

namespace pea {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4m(const float *p) { return *reinterpret_cast<const float4 *>(p); }


// image of one channel, in floats: [Wt0: HT * ET * 64 * 4][Wt1: HT * 4 * 64 * 4][b0p: 2 * HT * 16]
__host__ __device__ inline int mlp2_image_floats(int ET, int HT) { return HT * ET * 256 + HT * 1024 + 2 * HT * 16; }

// gridDim.x / n blocks per channel (the launch is pure latency: 2048 image elements per block): the LDS images of its two
// weight matrices + (GAT) the attention vectors in x space + the second layer's bias / attention rows
__global__ __launch_bounds__(256) void mlp2_pack_kernel(const Mlp2Launch L) {
    const int kPackSplit = (int)gridDim.x / L.n;
    const int chan = (int)blockIdx.x / kPackSplit, part = (int)blockIdx.x % kPackSplit;
    const Mlp2Chan &C = L.c[chan];
    const int EMB = L.emb, HID = L.hid, OUT = L.out;
    const int ET = EMB / 8, HT = HID / 32;
    const int t0 = part * 256 + (int)threadIdx.x, ts = 256 * kPackSplit;
    if (L.kind == PEA_KIND_SAGE) {
        // K = 2 * EMB inputs: k-steps [0, ET) multiply the neighbour mean (lin_rel), [ET, 2 ET) the row itself (lin_root);
        // 2 * OUT outputs: [0, OUT) = T_1 (lin_rel of layer 2), [OUT, 2 OUT) = its root term (lin_root).  All SAGE
        // weights are [out, in] (torch Linear).
        const int KT = 2 * ET;
        float *img = L.images + (size_t)chan * mlp2_image_floats(KT, HT);
        float *wt0 = img, *wt1 = img + HT * KT * 256, *b0p = wt1 + HT * 1024;
        for (int idx = t0; idx < HT * KT * 256; idx += ts) {
            const int e = idx & 3, lane = (idx >> 2) & 63, qq = (idx >> 8) % KT, t = (idx >> 8) / KT;
            const int i = 32 * t + (lane & 31), k = 4 * (2 * (qq % ET) + (lane >> 5)) + e;
            wt0[idx] = (qq < ET ? C.w0 : C.w0_root)[(size_t)i * EMB + k];
        }
        for (int idx = t0; idx < HT * 1024; idx += ts) {
            const int e = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 3, t = idx >> 10;
            const int j = lane & 31, i = 32 * t + 8 * g + 4 * (lane >> 5) + e;
            wt1[idx] = j < OUT ? C.w1[(size_t)j * HID + i] : j < 2 * OUT ? C.w1_root[(size_t)(j - OUT) * HID + i] : 0.f;
        }
        for (int idx = t0; idx < 2 * HT * 16; idx += ts) {
            const int v = idx & 15, t = (idx >> 4) % HT, half = (idx >> 4) / HT;
            const int i = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * half;
            b0p[idx] = C.b0 ? C.b0[i] : 0.f;
        }
        return;
    }
    float *img = L.images + (size_t)chan * mlp2_image_floats(ET, HT);
    float *wt0 = img, *wt1 = img + HT * ET * 256, *b0p = wt1 + HT * 1024;
    const bool gat = L.kind == PEA_KIND_GAT;
    // W0(i, k): hidden unit i from input k.  GAT lin.weight is [HID, EMB] (out-major), GCN weight is [EMB, HID]
    for (int idx = t0; idx < HT * ET * 256; idx += ts) {
        const int e = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) % ET, t = (idx >> 8) / ET;
        const int i = 32 * t + (lane & 31), k = 4 * (2 * q + (lane >> 5)) + e;
        wt0[idx] = gat ? C.w0[(size_t)i * EMB + k] : C.w0[(size_t)k * HID + i];
    }
    // W1(j, i): output j from hidden unit i, rows >= OUT are zero.  GAT: [OUT, HID]; GCN: [HID, OUT]
    for (int idx = t0; idx < HT * 1024; idx += ts) {
        const int e = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 3, t = idx >> 10;
        const int j = lane & 31, i = 32 * t + 8 * g + 4 * (lane >> 5) + e;
        wt1[idx] = j < OUT ? (gat ? C.w1[(size_t)j * HID + i] : C.w1[(size_t)i * OUT + j]) : 0.f;
    }
    for (int idx = t0; idx < 2 * HT * 16; idx += ts) {
        const int v = idx & 15, t = (idx >> 4) % HT, half = (idx >> 4) / HT;
        const int i = 32 * t + (v & 3) + 8 * (v >> 2) + 4 * half;
        b0p[idx] = C.b0 ? C.b0[i] : 0.f;
    }
    for (int o = t0; o < OUT; o += ts) {
        L.bias1[C.t1_col + o] = C.b1 ? C.b1[o] : 0.f;
        if (gat) {
            L.att_src1[C.t1_col + o] = C.att_src1[o];   // att_j multiplies the SOURCE row (natural units, agg_common.h: Soft)
            L.att_dst1[C.t1_col + o] = C.att_dst1[o];   // att_i multiplies the TARGET row
        }
    }
    if (gat) {
        // logits of the first layer from x itself: (W x) . att = x . (W^T att), natural units
        for (int k = (int)threadIdx.x * kPackSplit + part; k < EMB; k += ts) {
            float s = 0.f, d = 0.f;
            for (int i = 0; i < HID; ++i) {
                const float w = C.w0[(size_t)i * EMB + k];
                s = fmaf(C.att_src0[i], w, s);
                d = fmaf(C.att_dst0[i], w, d);
            }
            C.ws[k] = s;
            C.wd[k] = d;
        }
    }
}

extern __shared__ float mlp2_lds[];
// workgroup size: 1024 threads (16 waves per CU: one workgroup, the weight images fill the LDS) where 128 registers per
// lane suffice (emb = hidden = 64), else 512
template <int ET, int HT>
struct Mlp2Cfg {
    static constexpr int kThreads = (ET == 8 && HT == 2) ? 1024 : 512;
};

// one work item = (32-row tile, channel of the resident pass); consecutive waves take the channels of one tile (its x rows
// stay in L1), items are dealt round-robin over all waves of the grid (balanced to one item), and the input rows of the
// NEXT item are loaded before the current item's MFMA chains start
template <int ET>
struct Mlp2In {
    float4 xa[ET];
    int64_t row;
    bool valid;
};

template <int ET>
__device__ __forceinline__ void mlp2_load(const Mlp2Launch &L, const int *rows, int64_t n_rows, int64_t tile, int ch, int r32,
                                          int half, Mlp2In<ET> &in) {
    const Mlp2Chan &C = L.c[ch];
    const int64_t q0 = tile * 32 + r32;
    in.valid = q0 < n_rows;
    in.row = in.valid ? (rows ? (int64_t)rows[q0] : q0) : 0;
    // input row: the first layer's aggregate, or x itself where the row has no incoming edge there
    const bool lone = C.deg0[in.row] != 0;
    const float *src = lone ? L.x + in.row * L.ldx : L.a0 + in.row * L.ld_a0 + C.a0_col;
    float sc = 1.f;
    if (C.dinv && lone) {   // GCN: the self loop alone, norm = dinv_i^2
        const float di = C.dinv[in.row];
        sc = di * di;
    }
    if (!in.valid) sc = 0.f;
#pragma unroll
    for (int q = 0; q < ET; ++q) {
        const float4 t4 = ld4m(src + 4 * (2 * q + half));
        in.xa[q] = make_float4(sc * t4.x, sc * t4.y, sc * t4.z, sc * t4.w);
    }
}

template <int ET, int HT, bool TRAIN>
__global__ __launch_bounds__((Mlp2Cfg<ET, HT>::kThreads)) void mlp2_kernel(const Mlp2Launch L, const int *__restrict__ rows, int64_t n_rows) {
    constexpr int IMG = HT * ET * 256 + HT * 1024 + 2 * HT * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r32 = lane & 31;
    const int64_t n_tiles = (n_rows + 31) / 32;
    constexpr int kMlp2Threads = Mlp2Cfg<ET, HT>::kThreads;
    constexpr int WPB = kMlp2Threads / 64;
    int grp = 0;
    while (grp + 1 < L.n_groups && (int)blockIdx.x >= L.blk_start[grp + 1]) ++grp;
    const int64_t wave_global = (int64_t)((int)blockIdx.x - L.blk_start[grp]) * WPB + wave;
    const int64_t n_waves = (int64_t)(L.blk_start[grp + 1] - L.blk_start[grp]) * WPB;
    {
        const int c0 = grp * L.per_pass;
        const int nc = min(L.per_pass, L.n - c0);
        for (int idx = threadIdx.x * 4; idx < nc * IMG; idx += kMlp2Threads * 4)   // IMG is a multiple of 4
            *reinterpret_cast<float4 *>(mlp2_lds + idx) = ld4m(L.images + (size_t)c0 * IMG + idx);
        __syncthreads();
        const int64_t n_items = n_tiles * nc;
        Mlp2In<ET> cur, nxt;
        if (wave_global < n_items) mlp2_load<ET>(L, rows, n_rows, wave_global / nc, c0 + (int)(wave_global % nc), r32, half, cur);
        for (int64_t item = wave_global; item < n_items; item += n_waves) {
            const int cc = (int)(item % nc);
            const int64_t item2 = item + n_waves;
            if (item2 < n_items) mlp2_load<ET>(L, rows, n_rows, item2 / nc, c0 + (int)(item2 % nc), r32, half, nxt);
            const Mlp2Chan &C = L.c[c0 + cc];
            const float *img = mlp2_lds + (size_t)cc * IMG;
            const float *wt0 = img, *wt1 = img + HT * ET * 256, *b0p = wt1 + HT * 1024 + half * HT * 16;
            f32x16 acc[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
#pragma unroll
            for (int q = 0; q < ET; ++q) {
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    const float4 w = ld4m(wt0 + ((size_t)(t * ET + q) * 64 + lane) * 4);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, cur.xa[q].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, cur.xa[q].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, cur.xa[q].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, cur.xa[q].w, acc[t], 0, 0, 0);
                }
            }
            f32x16 out;
#pragma unroll
            for (int v = 0; v < 16; ++v) out[v] = 0.f;
            // training: keep the hidden tile (register v of tile t = hidden unit 32 t + (v & 3) + 8 (v >> 2) + 4 half of the
            // lane's row: registers 4g .. 4g+3 are 4 consecutive units, one float4 store)
            float *hrow = (TRAIN && cur.valid) ? L.h0 + cur.row * L.ld_h0 + C.h0_col + 4 * half : nullptr;
#pragma unroll
            for (int t = 0; t < HT; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 w = ld4m(wt1 + ((size_t)(t * 4 + g) * 64 + lane) * 4);
                    const float4 b = ld4m(b0p + t * 16 + 4 * g);
                    const float4 hv = make_float4(fmaxf(acc[t][4 * g + 0] + b.x, 0.f), fmaxf(acc[t][4 * g + 1] + b.y, 0.f),
                                                  fmaxf(acc[t][4 * g + 2] + b.z, 0.f), fmaxf(acc[t][4 * g + 3] + b.w, 0.f));
                    if (TRAIN && hrow) *reinterpret_cast<float4 *>(hrow + 32 * t + 8 * g) = hv;
                    out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, hv.x, out, 0, 0, 0);
                    out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, hv.y, out, 0, 0, 0);
                    out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, hv.z, out, 0, 0, 0);
                    out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, hv.w, out, 0, 0, 0);
                }
            }
            if (cur.valid) {
                float *dst = L.t1 + cur.row * L.ld_t1 + C.t1_col;
                const int slot = C.x_slot ? C.x_slot[cur.row] : -1;   // sharded: a gather source of layer 2 -> its exchange row too
                float *dst_x = C.x_buf + (int64_t)(slot < 0 ? 0 : slot) * C.x_ld;
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // registers 4g .. 4g+3 = outputs 8g + 4 half .. + 3
                    const int j0 = 8 * g + 4 * half;
                    if (j0 < L.out) {
                        const float4 o = make_float4(out[4 * g], out[4 * g + 1], out[4 * g + 2], out[4 * g + 3]);
                        *reinterpret_cast<float4 *>(dst + j0) = o;
                        if (slot >= 0) *reinterpret_cast<float4 *>(dst_x + j0) = o;
                    }
                }
            }
            cur = nxt;
        }
    }
}

// SAGE: in_c(n) = [M_0[n, rel(c)] | x[n]]  (M_0 = mean of the in-neighbours' x rows, 0 for rows without any), hidden =
// relu(in . [lin_rel0 ; lin_root0]^T + bias0), then ONE second product gives both T_1 = hidden . lin_rel1^T (layer 2 gathers
// it) and the root term hidden . lin_root1^T + bias1 (layer 2 adds the neighbour mean to it): reference nn SAGEConv as
// called at models/base.py:138, lin_rel(mean_j x_j) + lin_root(x_i), reassociated like the inference schedule of
// model.hip.  The two input halves are loaded a phase ahead: x of this item before its mean phase runs, the mean rows of
// the next item before this item's x phase runs.
template <int ET>
__device__ __forceinline__ void sage_rows(const Mlp2Launch &L, const int *rows, int64_t n_rows, int64_t tile, int r32,
                                          int64_t &row, bool &valid) {
    const int64_t q0 = tile * 32 + r32;
    valid = q0 < n_rows;
    row = valid ? (rows ? (int64_t)rows[q0] : q0) : 0;
}

template <int ET>
__device__ __forceinline__ void sage_load_mean(const Mlp2Launch &L, const Mlp2Chan &C, int64_t row, bool valid, int half,
                                               float4 (&a)[ET]) {
    const bool have = valid && C.deg0[row] == 0;   // rows without incoming edges were not aggregated: their mean is 0
    const float *src = L.a0 + row * L.ld_a0 + C.a0_col;
#pragma unroll
    for (int q = 0; q < ET; ++q) a[q] = have ? ld4m(src + 4 * (2 * q + half)) : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int ET, int HT>
__global__ __launch_bounds__(512) void mlp2_sage_kernel(const Mlp2Launch L, const int *__restrict__ rows, int64_t n_rows) {
    constexpr int KT = 2 * ET;
    constexpr int IMG = HT * KT * 256 + HT * 1024 + 2 * HT * 16;
    constexpr int WPB = 512 / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r32 = lane & 31;
    const int64_t n_tiles = (n_rows + 31) / 32;
    int grp = 0;
    while (grp + 1 < L.n_groups && (int)blockIdx.x >= L.blk_start[grp + 1]) ++grp;
    const int64_t wave_global = (int64_t)((int)blockIdx.x - L.blk_start[grp]) * WPB + wave;
    const int64_t n_waves = (int64_t)(L.blk_start[grp + 1] - L.blk_start[grp]) * WPB;
    const int c0 = grp * L.per_pass;
    const int nc = min(L.per_pass, L.n - c0);
    for (int idx = threadIdx.x * 4; idx < nc * IMG; idx += 512 * 4)
        *reinterpret_cast<float4 *>(mlp2_lds + idx) = ld4m(L.images + (size_t)c0 * IMG + idx);
    __syncthreads();
    const int64_t n_items = n_tiles * nc;
    float4 a[ET], b[ET];
    int64_t row = 0;
    bool valid = false;
    if (wave_global < n_items) {
        sage_rows<ET>(L, rows, n_rows, wave_global / nc, r32, row, valid);
        sage_load_mean<ET>(L, L.c[c0 + (int)(wave_global % nc)], row, valid, half, a);
    }
    for (int64_t item = wave_global; item < n_items; item += n_waves) {
        const int cc = (int)(item % nc);
        const Mlp2Chan &C = L.c[c0 + cc];
        const float *img = mlp2_lds + (size_t)cc * IMG;
        const float *wt0 = img, *wt1 = img + HT * KT * 256, *b0p = wt1 + HT * 1024 + half * HT * 16;
        {
            const float *src = L.x + row * L.ldx;
#pragma unroll
            for (int q = 0; q < ET; ++q) b[q] = valid ? ld4m(src + 4 * (2 * q + half)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x16 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
#pragma unroll
        for (int q = 0; q < ET; ++q) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const float4 w = ld4m(wt0 + ((size_t)(t * KT + q) * 64 + lane) * 4);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a[q].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a[q].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a[q].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a[q].w, acc[t], 0, 0, 0);
            }
        }
        const int64_t cur_row = row;
        const bool cur_valid = valid;
        const int64_t item2 = item + n_waves;
        if (item2 < n_items) {   // the mean rows of the next item travel while this item's x phase and second product run
            sage_rows<ET>(L, rows, n_rows, item2 / nc, r32, row, valid);
            sage_load_mean<ET>(L, L.c[c0 + (int)(item2 % nc)], row, valid, half, a);
        }
#pragma unroll
        for (int q = 0; q < ET; ++q) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const float4 w = ld4m(wt0 + ((size_t)(t * KT + ET + q) * 64 + lane) * 4);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, b[q].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, b[q].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, b[q].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, b[q].w, acc[t], 0, 0, 0);
            }
        }
        f32x16 out;
#pragma unroll
        for (int v = 0; v < 16; ++v) out[v] = 0.f;
        // training (two-step training schedule): keep the hidden tile for the backward, same positions as mlp2_kernel<TRAIN>
        float *hrow = (L.h0 && cur_valid) ? L.h0 + cur_row * L.ld_h0 + C.h0_col + 4 * half : nullptr;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = ld4m(wt1 + ((size_t)(t * 4 + g) * 64 + lane) * 4);
                const float4 bb = ld4m(b0p + t * 16 + 4 * g);
                const float4 hv = make_float4(fmaxf(acc[t][4 * g + 0] + bb.x, 0.f), fmaxf(acc[t][4 * g + 1] + bb.y, 0.f),
                                              fmaxf(acc[t][4 * g + 2] + bb.z, 0.f), fmaxf(acc[t][4 * g + 3] + bb.w, 0.f));
                if (hrow) *reinterpret_cast<float4 *>(hrow + 32 * t + 8 * g) = hv;
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, hv.x, out, 0, 0, 0);
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, hv.y, out, 0, 0, 0);
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, hv.z, out, 0, 0, 0);
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, hv.w, out, 0, 0, 0);
            }
        }
        if (cur_valid) {
            float *dst_t = L.t1 + cur_row * L.ld_t1 + C.t1_col;
            float *dst_r = L.r1 + cur_row * L.ld_r1 + C.r1_col;
            const int slot = C.x_slot ? C.x_slot[cur_row] : -1;   // sharded: a gather source of layer 2 -> its exchange row too
            float *dst_x = C.x_buf + (int64_t)(slot < 0 ? 0 : slot) * C.x_ld;
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // registers 4g .. 4g+3 = outputs 8g + 4 half .. + 3
                const int j0 = 8 * g + 4 * half;
                float4 o = make_float4(out[4 * g], out[4 * g + 1], out[4 * g + 2], out[4 * g + 3]);
                if (j0 < L.out) {
                    *reinterpret_cast<float4 *>(dst_t + j0) = o;
                    if (slot >= 0) *reinterpret_cast<float4 *>(dst_x + j0) = o;
                } else if (j0 < 2 * L.out) {
                    const int j = j0 - L.out;
                    if (C.b1) {
                        const float4 b1 = ld4m(C.b1 + j);
                        o = make_float4(o.x + b1.x, o.y + b1.y, o.z + b1.z, o.w + b1.w);
                    }
                    *reinterpret_cast<float4 *>(dst_r + j) = o;
                }
            }
        }
    }
}

}  // namespace

size_t mlp2_image_bytes(int kind, int emb, int hid) {
    return (size_t)mlp2_image_floats((kind == PEA_KIND_SAGE ? 2 : 1) * emb / 8, hid / 32) * sizeof(float);
}

bool mlp2_supported(int kind, int emb, int hid, int out) {
    // SAGE: the second product carries T_1 and the root term side by side in its 32 output rows
    return (emb == 64 || emb == 128) && (hid == 64 || hid == 128) && out >= 4 && out % 4 == 0 &&
           out <= (kind == PEA_KIND_SAGE ? 16 : 32);
}

int launch_mlp2_pack(const Mlp2Launch &L, hipStream_t stream) {
    ProfScope ps("pack_weights2", stream);
    const int kt = (L.kind == PEA_KIND_SAGE ? 2 : 1) * L.emb / 8, ht = L.hid / 32;
    const int split = std::max(4, std::min(32, mlp2_image_floats(kt, ht) / 2048));
    PEA_LAUNCH(mlp2_pack_kernel, dim3((unsigned)(L.n * split)), dim3(256), 0, stream, L);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

template <int ET, int HT, bool SAGE>
static int launch_mlp2_v(Mlp2Launch L, const int *rows, int64_t n_rows, hipStream_t stream) {
    constexpr size_t img = (size_t)(HT * (SAGE ? 2 * ET : ET) * 256 + HT * 1024 + 2 * HT * 16) * sizeof(float);
    constexpr size_t budget = 160 * 1024 - 2048;
    L.per_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)L.n, budget / img));
    // even out the passes (9 channels at 6 per pass: 5 + 4 instead of 6 + 3)
    const int passes = (L.n + L.per_pass - 1) / L.per_pass;
    L.per_pass = (L.n + passes - 1) / passes;
    const size_t lds = (size_t)L.per_pass * img;
    const bool train = !SAGE && L.h0 != nullptr;
    static size_t lds_set_v[2] = {0, 0};          // per kernel instantiation
    size_t &lds_set = lds_set_v[train ? 1 : 0];
    const void *fn = SAGE ? reinterpret_cast<const void *>(&mlp2_sage_kernel<ET, HT>)
                          : train ? reinterpret_cast<const void *>(&mlp2_kernel<ET, HT, true>)
                                  : reinterpret_cast<const void *>(&mlp2_kernel<ET, HT, false>);
    if (lds > lds_set) {
        PEA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        int dev = 0;
        PEA_HIP(hipGetDevice(&dev));
        PEA_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
    }
    const int64_t n_tiles = (n_rows + 31) / 32;
    constexpr int kMlp2Threads = SAGE ? 512 : Mlp2Cfg<ET, HT>::kThreads;
    constexpr int WPB = kMlp2Threads / 64;
    // workgroups per channel group in proportion to its channels; no more than its items can feed
    L.n_groups = passes;
    int blocks = 0;
    for (int g = 0; g < passes; ++g) {
        const int nc = std::min(L.per_pass, L.n - g * L.per_pass);
        int64_t want = std::max<int64_t>(1, ((int64_t)n_cu * nc + L.n / 2) / L.n);
        want = std::min<int64_t>(want, std::max<int64_t>(1, (n_tiles * nc + WPB - 1) / WPB));
        L.blk_start[g] = blocks;
        blocks += (int)want;
    }
    L.blk_start[passes] = blocks;
    const int grid = blocks;
    ProfScope ps("mlp2_fused", stream, 4.0 * (double)n_rows * (L.emb + (double)L.n * L.out * (SAGE ? 2 : 1)));
    if (SAGE) {
        PEA_REQUIRE(L.h0 == nullptr || L.ld_h0 % 4 == 0, PEA_ERR_ARG, "mlp2: the training variant needs an aligned hidden table");
        PEA_LAUNCH((mlp2_sage_kernel<ET, HT>), dim3((unsigned)grid), dim3(kMlp2Threads), lds, stream, L, rows, n_rows);
    } else if (train) {
        PEA_REQUIRE(L.ld_h0 % 4 == 0, PEA_ERR_ARG, "mlp2: the training variant needs an aligned hidden table");
        PEA_LAUNCH((mlp2_kernel<ET, HT, true>), dim3((unsigned)grid), dim3(kMlp2Threads), lds, stream, L, rows, n_rows);
    } else {
        PEA_LAUNCH((mlp2_kernel<ET, HT, false>), dim3((unsigned)grid), dim3(kMlp2Threads), lds, stream, L, rows, n_rows);
    }
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

int launch_mlp2(const Mlp2Launch &L, const int *rows, int64_t n_rows, hipStream_t stream) {
    PEA_REQUIRE(mlp2_supported(L.kind, L.emb, L.hid, L.out) && L.n > 0 && L.n <= kMaxMlp2Chan, PEA_ERR_ARG,
                "mlp2: unsupported widths (%d, %d, %d) or %d channels", L.emb, L.hid, L.out, L.n);
    if (n_rows <= 0) return PEA_OK;
    if (L.kind == PEA_KIND_SAGE) {
        PEA_REQUIRE(L.r1 != nullptr, PEA_ERR_ARG, "mlp2: SAGE needs the root-term destination");
        if (L.emb == 64 && L.hid == 64) return launch_mlp2_v<8, 2, true>(L, rows, n_rows, stream);
        if (L.emb == 64 && L.hid == 128) return launch_mlp2_v<8, 4, true>(L, rows, n_rows, stream);
        if (L.emb == 128 && L.hid == 64) return launch_mlp2_v<16, 2, true>(L, rows, n_rows, stream);
        return launch_mlp2_v<16, 4, true>(L, rows, n_rows, stream);
    }
    if (L.emb == 64 && L.hid == 64) return launch_mlp2_v<8, 2, false>(L, rows, n_rows, stream);
    if (L.emb == 64 && L.hid == 128) return launch_mlp2_v<8, 4, false>(L, rows, n_rows, stream);
    if (L.emb == 128 && L.hid == 64) return launch_mlp2_v<16, 2, false>(L, rows, n_rows, stream);
    return launch_mlp2_v<16, 4, false>(L, rows, n_rows, stream);
}

}  // namespace pea
