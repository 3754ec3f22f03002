// Two-step training schedule, dense half of the first layer's backward (data path), for every node n and channel c:
//     dZ_0[n, c] = (dT_1[n, c] . W1_c)  where  H[n, c] > 0, else 0        gradient of the first transform's pre-activations
//     dA_0[n, c] =  dZ_0[n, c] . W0_c                                       gradient of that transform's input row
// i.e. loss.backward() (reference solvers.py:215) through  conv_1.lin -> F.relu -> conv_0.lin  of models/base.py:138-139 for
// GATConv layers, both products chained in ONE kernel like the forward's csrc/mlp2.hip: the [32 rows x hidden] gradient tile
// stays in the MFMA accumulators between them (accumulator tile = the next product's B operand, no LDS, no lane movement).
// Round 3 first ran them as two launches of the generic transform kernel (pea_dense_batch: dZ_0 gated, then dA_0 = dZ_0 W_0:
// 0.32 + 0.38 ms on the 25m-shaped graph, dZ_0 written and read again in between).
//
// SAGE (pea_mlp2_backward_data_sage): the hidden row feeds two second-layer products and is fed by two first-layer ones,
//     dZ_0 = (dT_1 . lin_rel1 + dR_1 . lin_root1) gated,   dM_0 = dZ_0 . lin_rel0,   dX_root = dZ_0 . lin_root0
// -- product 1 runs over the two gradient rows side by side (k = 2 x out), product 2 twice on the same gated registers.
//
// Per wave: 32 rows.  Product 1 (dH^T = W1^T dT_1^T): v_mfma_f32_32x32x2_f32, A = W1^T tile [hidden unit][output j] from the
// LDS image, B = the lane's row of dT_1.  Gate: register v of tile t is hidden unit 32 t + (v & 3) + 8 (v >> 2) + 4 half of the
// lane's row -- the positions the forward stored H at, read back as float4s.  Product 2 (dA^T = W0^T dZ^T): A = W0^T tile
// [input k][hidden unit in accumulator-register order], B = the gated accumulator registers.  fp32 products and accumulation.
#include <algorithm>

#include "common.h"

namespace pea {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4b(const float *p) { return *reinterpret_cast<const float4 *>(p); }

constexpr int kMaxBwdChan = 32;
struct Mlp2BwdChan {
    const float *w0, *w1;               // GAT lin.weight of layer 1 [hid, emb] and of layer 2 [out, hid] (SAGE: lin_rel.weight)
    const float *w0b, *w1b;             // SAGE: lin_root.weight of the two layers, same shapes
    int dt1_col, h_col, dz_col, da_col;
    int dr1_col, db_col;                // SAGE: columns of the root term's output gradient / of the second input gradient
};
struct Mlp2BwdLaunch {
    int n, emb, hid, out, per_pass, n_groups;   // out: k of product 1 (SAGE: 2 x the layer's output width)
    int out_a;              // columns of product 1 that come from dt1 (the rest from dr1)
    int in_out;             // 1: weights are [in, out] (GCNConv.weight), 0: [out, in] (GATConv.lin.weight)
    int blk_start[kMaxBwdChan + 1];
    const float *dt1, *h, *dr1;
    float *dz, *da, *db;
    int64_t ld_dt1, ld_h, ld_dz, ld_da, ld_dr1, ld_db;
    float *images;
    const int *rows;        // optional: only these rows (ids), *count of them (device memory); null: rows 0 .. n_rows - 1
    const int *count;
    Mlp2BwdChan c[kMaxBwdChan];
};

__host__ __device__ inline int bwd_image_floats(int HT, int OT, int NQ, bool sage = false) {
    return HT * NQ * 256 + (sage ? 2 : 1) * OT * HT * 4 * 256;
}

__global__ __launch_bounds__(256) void mlp2_bwd_pack_kernel(const Mlp2BwdLaunch L) {
    const int split = (int)gridDim.x / L.n;
    const int chan = (int)blockIdx.x / split, part = (int)blockIdx.x % split;
    const Mlp2BwdChan &C = L.c[chan];
    const int EMB = L.emb, HID = L.hid, OUT = L.out;
    const int HT = HID / 32, OT = EMB / 32, NQ = (OUT + 7) / 8;
    const bool sage = L.dr1 != nullptr;
    float *img = L.images + (size_t)chan * bwd_image_floats(HT, OT, NQ, sage);
    float *w1t = img, *w0t = img + HT * NQ * 256;
    const int t0 = part * 256 + (int)threadIdx.x, ts = 256 * split;
    for (int idx = t0; idx < HT * NQ * 256; idx += ts) {          // A(i = hidden unit, k = output j) = W1[j, i]
        const int e = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) % NQ, t = (idx >> 8) / NQ;
        const int i = 32 * t + (lane & 31), k = 4 * (2 * q + (lane >> 5)) + e;
        float v = 0.f;
        if (k < L.out_a) v = L.in_out ? C.w1[(size_t)i * OUT + k] : C.w1[(size_t)k * HID + i];
        else if (k < OUT) v = C.w1b[(size_t)(k - L.out_a) * HID + i];        // SAGE: the root term's rows of product 1
        w1t[idx] = v;
    }
    for (int idx = t0; idx < OT * HT * 1024; idx += ts) {         // A(i = input k, hidden unit u in register order) = W0[u, i]
        const int e = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 3, t = ((idx >> 10) % HT), te = (idx >> 10) / HT;
        const int i = 32 * te + (lane & 31), u = 32 * t + 8 * g + 4 * (lane >> 5) + e;
        w0t[idx] = L.in_out ? C.w0[(size_t)i * HID + u] : C.w0[(size_t)u * EMB + i];
        if (sage) w0t[OT * HT * 1024 + idx] = C.w0b[(size_t)u * EMB + i];
    }
}

extern __shared__ float bwd_lds[];

template <int HT, int NQ>
struct BwdIn {
    float4 d[NQ];            // the lane's row of dT_1, chunk q = columns 4 (2 q + half) ..
    float4 h[HT * 4];        // the lane's pieces of the H row (gate), tile t / group g at 32 t + 8 g + 4 half
    int64_t row;
    bool valid;
};

template <int HT, int NQ>
__device__ __forceinline__ void bwd_load(const Mlp2BwdLaunch &L, int64_t n_rows, int64_t tile, int ch, int r32, int half,
                                         BwdIn<HT, NQ> &in) {
    const Mlp2BwdChan &C = L.c[ch];
    const int64_t q0 = tile * 32 + r32;
    in.valid = q0 < n_rows;
    in.row = in.valid ? (L.rows ? (int64_t)L.rows[q0] : q0) : 0;
    const float *dsrc = L.dt1 + in.row * L.ld_dt1 + C.dt1_col;
    const float *rsrc = L.dr1 ? L.dr1 + in.row * L.ld_dr1 + C.dr1_col - L.out_a : dsrc;   // columns out_a .. out - 1 (SAGE)
    const float *hsrc = L.h + in.row * L.ld_h + C.h_col + 4 * half;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = 4 * (2 * q + half);
        in.d[q] = (in.valid && c < L.out) ? ld4b((c < L.out_a ? dsrc : rsrc) + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) in.h[t * 4 + g] = ld4b(hsrc + 32 * t + 8 * g);   // row 0 for invalid lanes: harmless
}

template <int HT, int OT, int NQ, bool SAGE>
__global__ __launch_bounds__(512) void mlp2_bwd_kernel(const Mlp2BwdLaunch L, int64_t n_rows_max) {
    const int64_t n_rows = L.count ? (int64_t)*L.count : n_rows_max;   // a listed row set keeps its length on the device
    constexpr int IMG = HT * NQ * 256 + (SAGE ? 2 : 1) * OT * HT * 4 * 256;
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
        *reinterpret_cast<float4 *>(bwd_lds + idx) = ld4b(L.images + (size_t)c0 * IMG + idx);
    __syncthreads();
    const int64_t n_items = n_tiles * nc;
    BwdIn<HT, NQ> cur, nxt;
    if (wave_global < n_items) bwd_load<HT, NQ>(L, n_rows, wave_global / nc, c0 + (int)(wave_global % nc), r32, half, cur);
    for (int64_t item = wave_global; item < n_items; item += n_waves) {
        const int cc = (int)(item % nc);
        const int64_t item2 = item + n_waves;
        if (item2 < n_items) bwd_load<HT, NQ>(L, n_rows, item2 / nc, c0 + (int)(item2 % nc), r32, half, nxt);
        const Mlp2BwdChan &C = L.c[c0 + cc];
        const float *img = bwd_lds + (size_t)cc * IMG;
        const float *w1t = img, *w0t = img + HT * NQ * 256;
        f32x16 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const float4 w = ld4b(w1t + ((size_t)(t * NQ + q) * 64 + lane) * 4);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, cur.d[q].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, cur.d[q].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, cur.d[q].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, cur.d[q].w, acc[t], 0, 0, 0);
            }
        }
        f32x16 out[OT];
#pragma unroll
        for (int te = 0; te < OT; ++te)
#pragma unroll
            for (int v = 0; v < 16; ++v) out[te][v] = 0.f;
        float *zrow = L.dz + cur.row * L.ld_dz + C.dz_col + 4 * half;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 hv = cur.h[t * 4 + g];
                const float4 z = make_float4(hv.x > 0.f ? acc[t][4 * g + 0] : 0.f, hv.y > 0.f ? acc[t][4 * g + 1] : 0.f,
                                             hv.z > 0.f ? acc[t][4 * g + 2] : 0.f, hv.w > 0.f ? acc[t][4 * g + 3] : 0.f);
                if (cur.valid) *reinterpret_cast<float4 *>(zrow + 32 * t + 8 * g) = z;
#pragma unroll
                for (int te = 0; te < OT; ++te) {
                    const float4 w = ld4b(w0t + ((size_t)((te * HT + t) * 4 + g) * 64 + lane) * 4);
                    out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, z.x, out[te], 0, 0, 0);
                    out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, z.y, out[te], 0, 0, 0);
                    out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, z.z, out[te], 0, 0, 0);
                    out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, z.w, out[te], 0, 0, 0);
                }
            }
        }
        if (cur.valid) {
            float *arow = L.da + cur.row * L.ld_da + C.da_col + 4 * half;
#pragma unroll
            for (int te = 0; te < OT; ++te)
#pragma unroll
                for (int g = 0; g < 4; ++g)   // registers 4g .. 4g+3 = inputs 32 te + 8 g + 4 half .. + 3
                    *reinterpret_cast<float4 *>(arow + 32 * te + 8 * g) =
                        make_float4(out[te][4 * g], out[te][4 * g + 1], out[te][4 * g + 2], out[te][4 * g + 3]);
        }
        if (SAGE) {   // the same gated registers through lin_root0: the root term's input gradient
            const float *w0r = w0t + OT * HT * 1024;
#pragma unroll
            for (int te = 0; te < OT; ++te)
#pragma unroll
                for (int v = 0; v < 16; ++v) out[te][v] = 0.f;
#pragma unroll
            for (int t = 0; t < HT; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 hv = cur.h[t * 4 + g];
                    const float4 z = make_float4(hv.x > 0.f ? acc[t][4 * g + 0] : 0.f, hv.y > 0.f ? acc[t][4 * g + 1] : 0.f,
                                                 hv.z > 0.f ? acc[t][4 * g + 2] : 0.f, hv.w > 0.f ? acc[t][4 * g + 3] : 0.f);
#pragma unroll
                    for (int te = 0; te < OT; ++te) {
                        const float4 w = ld4b(w0r + ((size_t)((te * HT + t) * 4 + g) * 64 + lane) * 4);
                        out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, z.x, out[te], 0, 0, 0);
                        out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, z.y, out[te], 0, 0, 0);
                        out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, z.z, out[te], 0, 0, 0);
                        out[te] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, z.w, out[te], 0, 0, 0);
                    }
                }
            }
            if (cur.valid) {
                float *brow = L.db + cur.row * L.ld_db + C.db_col + 4 * half;
#pragma unroll
                for (int te = 0; te < OT; ++te)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4 *>(brow + 32 * te + 8 * g) =
                            make_float4(out[te][4 * g], out[te][4 * g + 1], out[te][4 * g + 2], out[te][4 * g + 3]);
            }
        }
        cur = nxt;
    }
}

template <int HT, int OT, int NQ, bool SAGE>
int launch_bwd_v(Mlp2BwdLaunch L, int64_t n_rows, hipStream_t stream) {
    constexpr size_t img = (size_t)(HT * NQ * 256 + (SAGE ? 2 : 1) * OT * HT * 4 * 256) * sizeof(float);
    // two 8-wave workgroups per CU (the kernel needs ~150 registers per lane: 512 threads): each may hold half the LDS
    // (an image larger than that -- SAGE at width 128: 144 KB -- gets the whole CU: one workgroup, one channel per pass)
    constexpr size_t budget = img > (160 * 1024 - 2048) / 2 ? (160 * 1024 - 2048) : (160 * 1024 - 2048) / 2;
    static_assert(img <= 160 * 1024 - 2048, "mlp2_bwd: a channel's weight image must fit the CU's LDS");
    L.per_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)L.n, budget / img));
    const int passes = (L.n + L.per_pass - 1) / L.per_pass;
    L.per_pass = (L.n + passes - 1) / passes;
    const size_t lds = (size_t)L.per_pass * img;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        PEA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp2_bwd_kernel<HT, OT, NQ, SAGE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
    constexpr int WPB = 512 / 64;
    L.n_groups = passes;
    int blocks = 0;
    for (int g = 0; g < passes; ++g) {
        const int nc = std::min(L.per_pass, L.n - g * L.per_pass);
        int64_t want = std::max<int64_t>(1, ((int64_t)(2 * lds <= 160 * 1024 - 2048 ? 2 : 1) * n_cu * nc + L.n / 2) / L.n);
        want = std::min<int64_t>(want, std::max<int64_t>(1, (n_tiles * nc + WPB - 1) / WPB));
        L.blk_start[g] = blocks;
        blocks += (int)want;
    }
    L.blk_start[passes] = blocks;
    {
        ProfScope ps("pack_weights2b", stream);
        const int split = std::max(2, std::min(32, bwd_image_floats(HT, OT, NQ, SAGE) / 2048));
        PEA_LAUNCH(mlp2_bwd_pack_kernel, dim3((unsigned)(L.n * split)), dim3(256), 0, stream, L);
        PEA_HIP(hipGetLastError());
    }
    ProfScope ps("mlp2_bwd_fused", stream, 4.0 * (double)n_rows * L.n * (L.out + 2.0 * L.hid + (SAGE ? 2.0 : 1.0) * L.emb));
    PEA_LAUNCH((mlp2_bwd_kernel<HT, OT, NQ, SAGE>), dim3((unsigned)blocks), dim3(512), lds, stream, L, n_rows);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

template <int HT, int OT>
int launch_bwd_q(const Mlp2BwdLaunch &L, int64_t n_rows, hipStream_t stream) {
    const int nq = (L.out + 7) / 8;
    if (L.dr1) {     // SAGE: k = 2 x (4 .. 16)
        if (nq == 1) return launch_bwd_v<HT, OT, 1, true>(L, n_rows, stream);
        if (nq == 2) return launch_bwd_v<HT, OT, 2, true>(L, n_rows, stream);
        if (nq == 3) return launch_bwd_v<HT, OT, 3, true>(L, n_rows, stream);
        return launch_bwd_v<HT, OT, 4, true>(L, n_rows, stream);
    }
    if (nq == 1) return launch_bwd_v<HT, OT, 1, false>(L, n_rows, stream);
    if (nq == 2) return launch_bwd_v<HT, OT, 2, false>(L, n_rows, stream);
    if (nq == 3) return launch_bwd_v<HT, OT, 3, false>(L, n_rows, stream);
    return launch_bwd_v<HT, OT, 4, false>(L, n_rows, stream);
}

}  // namespace
}  // namespace pea

using namespace pea;

// (sized for the larger of the two variants: SAGE carries two first-layer images and a 2 x out wide second-layer image)
extern "C" size_t pea_mlp2_backward_data_workspace_bytes(int n_chan, int emb, int hid, int out) {
    if (n_chan <= 0 || n_chan > kMaxBwdChan || emb % 32 || hid % 32 || out <= 0) return 0;
    return (size_t)n_chan * bwd_image_floats(hid / 32, emb / 32, (2 * out + 7) / 8, true) * sizeof(float) + 256;
}

static int mlp2_backward_data_impl(int64_t n_rows, int n_chan, const pea_mlp2_bwd_chan *chans_host, const pea_mlp2_bwd_chan_sage *sage_host,
                                   int emb, int hid, int out, const float *dt1, int64_t ld_dt1, const float *dr1, int64_t ld_dr1,
                                   const float *h, int64_t ld_h, float *dz, int64_t ld_dz, float *da, int64_t ld_da, float *db,
                                   int64_t ld_db, const int32_t *rows, const int32_t *count_dev, int weights_in_out, void *workspace,
                                   size_t workspace_bytes, void *stream);

extern "C" int pea_mlp2_backward_data(int64_t n_rows, int n_chan, const pea_mlp2_bwd_chan *chans_host, int emb, int hid, int out,
                                      const float *dt1, int64_t ld_dt1, const float *h, int64_t ld_h, float *dz, int64_t ld_dz,
                                      float *da, int64_t ld_da, const int32_t *rows, const int32_t *count_dev, int weights_in_out,
                                      void *workspace, size_t workspace_bytes, void *stream) {
    return mlp2_backward_data_impl(n_rows, n_chan, chans_host, nullptr, emb, hid, out, dt1, ld_dt1, nullptr, 0, h, ld_h, dz, ld_dz, da,
                                   ld_da, nullptr, 0, rows, count_dev, weights_in_out, workspace, workspace_bytes, stream);
}

extern "C" int pea_mlp2_backward_data_sage(int64_t n_rows, int n_chan, const pea_mlp2_bwd_chan *chans_host,
                                           const pea_mlp2_bwd_chan_sage *sage_host, int emb, int hid, int out, const float *dt1,
                                           int64_t ld_dt1, const float *dr1, int64_t ld_dr1, const float *h, int64_t ld_h, float *dz,
                                           int64_t ld_dz, float *dm, int64_t ld_dm, float *dxr, int64_t ld_dxr, const int32_t *rows,
                                           const int32_t *count_dev, void *workspace, size_t workspace_bytes, void *stream) {
    PEA_REQUIRE(sage_host && dr1 && dxr && ld_dr1 % 4 == 0 && ld_dxr % 4 == 0 && out <= 16, PEA_ERR_ARG,
                "mlp2_backward_data_sage: null pointer, unaligned stride or out > 16");
    return mlp2_backward_data_impl(n_rows, n_chan, chans_host, sage_host, emb, hid, out, dt1, ld_dt1, dr1, ld_dr1, h, ld_h, dz, ld_dz, dm,
                                   ld_dm, dxr, ld_dxr, rows, count_dev, 0, workspace, workspace_bytes, stream);
}

static int mlp2_backward_data_impl(int64_t n_rows, int n_chan, const pea_mlp2_bwd_chan *chans_host, const pea_mlp2_bwd_chan_sage *sage_host,
                                   int emb, int hid, int out, const float *dt1, int64_t ld_dt1, const float *dr1, int64_t ld_dr1,
                                   const float *h, int64_t ld_h, float *dz, int64_t ld_dz, float *da, int64_t ld_da, float *db,
                                   int64_t ld_db, const int32_t *rows, const int32_t *count_dev, int weights_in_out, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    PEA_REQUIRE((rows == nullptr) == (count_dev == nullptr), PEA_ERR_ARG, "mlp2_backward_data: a row list comes with its device-side count");
    PEA_REQUIRE(n_rows >= 0 && n_chan > 0 && n_chan <= kMaxBwdChan && chans_host, PEA_ERR_ARG, "mlp2_backward_data: %d channels (1..%d)", n_chan, kMaxBwdChan);
    PEA_REQUIRE((emb == 64 || emb == 128) && (hid == 64 || hid == 128) && out >= 4 && out % 4 == 0 && out <= 32, PEA_ERR_ARG,
                "mlp2_backward_data: unsupported widths (%d, %d, %d)", emb, hid, out);
    PEA_REQUIRE(dt1 && h && dz && da && workspace, PEA_ERR_ARG, "mlp2_backward_data: null pointer");
    PEA_REQUIRE(ld_dt1 % 4 == 0 && ld_h % 4 == 0 && ld_dz % 4 == 0 && ld_da % 4 == 0, PEA_ERR_ARG, "mlp2_backward_data: row strides must be multiples of 4");
    PEA_REQUIRE(workspace_bytes >= pea_mlp2_backward_data_workspace_bytes(n_chan, emb, hid, out), PEA_ERR_NOMEM, "mlp2_backward_data: workspace too small");
    if (n_rows == 0) return PEA_OK;
    Mlp2BwdLaunch L{};
    L.n = n_chan;
    L.emb = emb;
    L.hid = hid;
    L.out = sage_host ? 2 * out : out;
    L.out_a = out;
    L.dr1 = dr1;
    L.ld_dr1 = ld_dr1;
    L.db = db;
    L.ld_db = ld_db;
    L.dt1 = dt1;
    L.h = h;
    L.dz = dz;
    L.da = da;
    L.ld_dt1 = ld_dt1;
    L.ld_h = ld_h;
    L.ld_dz = ld_dz;
    L.ld_da = ld_da;
    L.images = aligned_ws(workspace);
    L.rows = rows;
    L.count = count_dev;
    L.in_out = weights_in_out ? 1 : 0;
    for (int c = 0; c < n_chan; ++c) {
        const pea_mlp2_bwd_chan &s = chans_host[c];
        PEA_REQUIRE(s.w0 && s.w1 && s.dt1_col % 4 == 0 && s.h_col % 4 == 0 && s.dz_col % 4 == 0 && s.da_col % 4 == 0 &&
                        s.dt1_col >= 0 && s.h_col >= 0 && s.dz_col >= 0 && s.da_col >= 0 && s.dt1_col + out <= ld_dt1 &&
                        s.h_col + hid <= ld_h && s.dz_col + hid <= ld_dz && s.da_col + emb <= ld_da,
                    PEA_ERR_ARG, "mlp2_backward_data: channel %d columns out of range or unaligned", c);
        L.c[c].w0 = s.w0;
        L.c[c].w1 = s.w1;
        L.c[c].dt1_col = s.dt1_col;
        L.c[c].h_col = s.h_col;
        L.c[c].dz_col = s.dz_col;
        L.c[c].da_col = s.da_col;
        if (sage_host) {
            const pea_mlp2_bwd_chan_sage &r = sage_host[c];
            PEA_REQUIRE(r.w0_root && r.w1_root && r.dr1_col % 4 == 0 && r.dxr_col % 4 == 0 && r.dr1_col >= 0 && r.dxr_col >= 0 &&
                            r.dr1_col + out <= ld_dr1 && r.dxr_col + emb <= ld_db,
                        PEA_ERR_ARG, "mlp2_backward_data_sage: channel %d root columns out of range or unaligned", c);
            L.c[c].w0b = r.w0_root;
            L.c[c].w1b = r.w1_root;
            L.c[c].dr1_col = r.dr1_col;
            L.c[c].db_col = r.dxr_col;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (hid == 64 && emb == 64) return launch_bwd_q<2, 2>(L, n_rows, st);
    if (hid == 64 && emb == 128) return launch_bwd_q<2, 4>(L, n_rows, st);
    if (hid == 128 && emb == 64) return launch_bwd_q<4, 2>(L, n_rows, st);
    return launch_bwd_q<4, 4>(L, n_rows, st);
}
