// Dense per-layer transforms of the PEA path for gfx950: h = x.W^T (GAT), x.W (GCN),
// lin_rel(mean) + lin_root(x) (SAGE) -- the torch.nn.Linear / matmul calls inside the PyG convs
// (reference call site graph_recsys_benchmark/models/base.py:138-139, SURVEY.md 2.1 "dense transform").
//
// The weights of every layer are re-packed each forward (they change every optimizer step) into
// k-major blocks  B[k][col]  so one GEMM job can serve several channels that share the same input
// (all P first-layer channels read the same x, models/base.py:192-193).
//
// f32-input MFMA (exact fp32 products, fp32 accumulation).  Three kernels:
//   gemm_persist_kernel  (k <= 128): persistent workgroups keep the whole k-major B (+ bias row) in LDS; one wave owns a
//                        32-row tile, its A fragment in registers (lane (r, h) holds float4 chunks 2q + h of row r -- the k
//                        order is permuted identically on the B side, which a sum over k does not care about), and walks
//                        32-column tiles: 16 LDS reads ahead, 32 v_mfma_f32_32x32x2_f32, 16 stores with no memory wait
//                        anywhere in the column loop.
//   gemm_skinny_kernel   (<= 16 output columns): 64 rows per wave, 4 independent v_mfma_f32_16x16x4_f32 chains, float4 stores.
//   gemm_mfma_kernel     (k > 128): B tiles staged through a double-buffered LDS image.
// Measured (profiles/tools/gemm_bench.cpp, mfma_peak.cpp, mfma_lds.cpp): the fp32 matrix pipe sustains 154 TFLOP/s on this
// part; the [N,64]x[64,576] first-layer transform runs at 66-75 TFLOP/s because fp32 MFMA chains and an HBM store stream
// are additive here (profiles/tools/overlap_probe.cpp: two kernels on two streams, or specialised waves of one kernel,
// dword or float4 stores: t(both) = t(MFMA) + t(stores) within 3 %): 0.17 ms of MFMA chains + 0.09 ms for 630 MB of
// output = 0.26 ms, measured 0.27 ms (rocBLAS: 0.35 ms).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace pea {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxBatch = 12;
struct GemmBatch {
    int n;
    GemmJob j[kMaxBatch];
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Branch-free A fragment load: every lane issues all its float4 loads back to back (invalid rows / k-ranges are
// clamped to a valid address and zeroed afterwards), then the edge-less-row transform is applied with selects.
template <int KH, bool IL = false>
__device__ __forceinline__ void load_a(const GemmJob &J, int64_t srow, bool rv, int kbase, float (&a)[KH]) {
    // IL: the two k-halves of a row interleave by float4 (kbase = 4h): one load instruction then reads 32 contiguous
    // bytes per row instead of two 16-byte pieces 128 bytes apart
    const int K = J.K1 + J.K2;
    const bool alt = J.a1_mask != nullptr && rv && J.a1_mask[srow] != 0;
    float sc = 1.f;
    if (J.a1_scale) {
        const float di = J.a1_scale[srow];
        sc = alt ? di * di : 1.f;
    }
    const float *p1 = alt ? J.a1_alt + srow * J.lda_alt : J.A1 + srow * J.lda1;
    const float *p2 = J.K2 > 0 ? J.A2 + srow * J.lda2 : p1;
    float4 v[KH / 4], bv[KH / 4];
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
        const int k = IL ? kbase + q * 8 : kbase + q * 4;
        const bool ok = k < K;
        const float *p = (k < J.K1 || !ok) ? p1 + (ok ? k : 0) : p2 + (k - J.K1);
        v[q] = ld4(p);
        bv[q] = (J.a1_mask != nullptr && J.a1_bias != nullptr && k < J.K1) ? ld4(J.a1_bias + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
        const int k = IL ? kbase + q * 8 : kbase + q * 4;
        float4 t = v[q];
        if (alt && k < J.K1) {  // same roundings as the aggregation kernel's finish_row: product, + bias, relu
            t = make_float4(fmaxf(sc * t.x + bv[q].x, 0.f), fmaxf(sc * t.y + bv[q].y, 0.f), fmaxf(sc * t.z + bv[q].z, 0.f),
                            fmaxf(sc * t.w + bv[q].w, 0.f));
        }
        const bool keep = rv && k < K;
        a[q * 4 + 0] = keep ? t.x : 0.f;
        a[q * 4 + 1] = keep ? t.y : 0.f;
        a[q * 4 + 2] = keep ? t.z : 0.f;
        a[q * 4 + 3] = keep ? t.w : 0.f;
    }
}

// Where column c of a job's output goes.  The job is wave-uniform, so the segment table is read with scalar loads and
// the per-lane answer is a chain of selects (no memory wait in the epilogue).
struct OutCol {
    float *dst;
    int ld, relu;
};
__device__ __forceinline__ OutCol find_out(const GemmJob &J, int c) {
    OutCol o{nullptr, 0, 0};
    for (int sg = 0; sg < J.n_seg; ++sg) {
        const bool in = c >= J.seg[sg].c0 && c < J.seg[sg].c1;
        o.dst = in ? J.seg[sg].dst + (c - J.seg[sg].c0) : o.dst;
        o.ld = in ? J.seg[sg].ld : o.ld;
        o.relu = in ? J.seg[sg].relu : o.relu;
    }
    if (c >= J.n_out) o.dst = nullptr;
    return o;
}

// Accumulator rows of a lane: (reg & 3) + 8 * (reg >> 2) + 4 * h.  All 16 stores are issued back to back: nothing in
// here may wait on memory (a wait between two stores serialises them on the store acknowledgements, which is what
// bounded the first version of this epilogue).  orow: row ids of a row-list job, loaded once per tile (-1 = past the
// end); null-list jobs address arithmetically.
__device__ __forceinline__ void store_tile(const f32x16 &acc, const OutCol o, float bias, bool listed, const int (&orow)[16],
                                           unsigned valid, int64_t row0, int h) {
    if (!o.dst) return;
    float v[16];  // every value first, in straight-line code: the predicated stores below then touch no loaded register
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const float t = acc[reg] + bias;
        v[reg] = o.relu ? fmaxf(t, 0.f) : t;
    }
    if (listed) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            if ((valid >> reg) & 1u) o.dst[(int64_t)orow[reg] * o.ld] = v[reg];
        return;
    }
    float *p = o.dst + (row0 + 4 * h) * o.ld;
    if (valid == 0xffffu) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) p[(int64_t)((reg & 3) + 8 * (reg >> 2)) * o.ld] = v[reg];
    } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            if ((valid >> reg) & 1u) p[(int64_t)((reg & 3) + 8 * (reg >> 2)) * o.ld] = v[reg];
    }
}

// Row ids of the lane's 16 accumulator rows (row-list jobs) and the bit mask of those inside the job; consuming the
// loads here keeps memory waits out of the epilogue.
__device__ __forceinline__ unsigned load_orow(const int *rows, int64_t row0, int h, int64_t n_rows, int (&orow)[16]) {
    unsigned valid = 0;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t g = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        orow[reg] = g < n_rows ? (rows ? rows[g] : (int)g) : 0;
        valid |= (g < n_rows ? 1u : 0u) << reg;
    }
    if (rows) {
        int probe = 0;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) probe |= orow[reg];
        if (probe < 0) valid = 0;  // never true (ids are non-negative): makes the loads complete here
    }
    return valid;
}

// 4 waves = 4 row tiles per block share each 32-column B tile through a double-buffered LDS image
// (one barrier per stage; the next tile's global loads are in flight during the MFMAs).
template <int KH>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const GemmBatch Bt, const int *__restrict__ rows, int64_t n_rows) {
    __shared__ float Bs[2][2 * KH][32];
    constexpr int NLD = (2 * KH * 32 / 4) / 256;  // float4 loads per thread per B tile
    const GemmJob &J = Bt.j[blockIdx.y];
    if (J.rows) {
        rows = J.rows;
        n_rows = J.n_rows;
    }
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (tid >> 6)) * 32;
    const int64_t grow = row0 + r;
    const bool rv = grow < n_rows;
    const int64_t srow = rv ? (rows ? (int64_t)rows[grow] : grow) : 0;
    const int K = J.K1 + J.K2;
    const int nkc = (K + 2 * KH - 1) / (2 * KH), nct = (J.n_out + 31) / 32;
    const int n_stage = nkc * nct;
    float a[KH];
    if (nkc == 1) load_a<KH>(J, srow, rv, h * KH, a);
    int orow[16];
    const unsigned valid = load_orow(rows, row0, h, n_rows, orow);

    float4 pre[NLD];
    auto fetch = [&](int stage) {
        const int col0 = (stage / nkc) * 32, kc = (stage % nkc) * 2 * KH;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            const int k = kc + idx / 8, c = col0 + (idx & 7) * 4;
            pre[i] = (k < K && c < J.ldb) ? ld4(J.B + (size_t)k * J.ldb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            *reinterpret_cast<float4 *>(&Bs[buf][idx / 8][(idx & 7) * 4]) = pre[i];
        }
    };
    fetch(0);
    stash(0);
    f32x16 acc;
    for (int s = 0; s < n_stage; ++s) {
        __syncthreads();
        const int col0 = (s / nkc) * 32, kci = s % nkc;
        if (s + 1 < n_stage) fetch(s + 1);
        if (kci == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        }
        if (nkc > 1) load_a<KH>(J, srow, rv, kci * 2 * KH + h * KH, a);
        const float *bs = &Bs[s & 1][h * KH][r];
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bs[kk * 32], acc, 0, 0, 0);
        if (kci == nkc - 1) {
            // epilogue: this lane owns column c of rows (reg&3) + 8*(reg>>2) + 4*h
            const int c = col0 + r;
            const OutCol o = find_out(J, c);
            const float bias = (J.bias && c < J.n_out) ? J.bias[c] : 0.f;
            store_tile(acc, o, bias, rows != nullptr, orow, valid, row0, h);
        }
        if (s + 1 < n_stage) stash((s + 1) & 1);
    }
}

extern __shared__ float g_lds[];

// Deep k (K > 128: the first layer's input gradient dx = dT_0 W_cat, K = sum of the channels' hidden widths), outputs up
// to 32 * NCT columns: the staged kernel above walks (column tile, k chunk) stages and so reads every A chunk once per
// column tile and takes a barrier per stage; here a stage is one 128-deep k chunk for ALL column tiles (NCT accumulator
// tiles per wave): A is read once, half the barriers, and the next chunk's A fragment and B tile are in flight during the
// MFMAs.  Same k order per output as the staged kernel (chunks in order, within a chunk kk = 0..63 with the two k-halves
// on the two lane halves).
template <int NCT>
__global__ __launch_bounds__(256) void gemm_deep_kernel(const GemmBatch Bt, const int *__restrict__ rows, int64_t n_rows) {
    constexpr int KH = 64, W = 32 * NCT, NLD = 32 * W / 256;   // float4 loads per thread per B stage (128 x W floats)
    const GemmJob &J = Bt.j[blockIdx.y];
    if (J.rows) {
        rows = J.rows;
        n_rows = J.n_rows;
    }
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (tid >> 6)) * 32;
    const int64_t grow = row0 + r;
    const bool rv = grow < n_rows;
    const int64_t srow = rv ? (rows ? (int64_t)rows[grow] : grow) : 0;
    const int K = J.K1 + J.K2;
    const int nkc = (K + 2 * KH - 1) / (2 * KH);
    int orow[16];
    const unsigned valid = load_orow(rows, row0, h, n_rows, orow);
    float4 pre[NLD];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            const int k = kc * 2 * KH + idx / (W / 4), c = (idx % (W / 4)) * 4;
            pre[i] = (k < K && c < J.ldb) ? ld4(J.B + (size_t)k * J.ldb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            *reinterpret_cast<float4 *>(g_lds + ((size_t)buf * 2 * KH + idx / (W / 4)) * W + (idx % (W / 4)) * 4) = pre[i];
        }
    };
    float a[KH], a2[KH];
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
    fetch(0);
    stash(0);
    load_a<KH>(J, srow, rv, h * KH, a);
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
        const bool more = kc + 1 < nkc;
        if (more) {
            fetch(kc + 1);
            load_a<KH>(J, srow, rv, (kc + 1) * 2 * KH + h * KH, a2);
        }
        const float *bs = g_lds + ((size_t)(kc & 1) * 2 * KH + h * KH) * W + r;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bs[kk * W + 32 * ct], acc[ct], 0, 0, 0);
        if (more) {
            stash((kc + 1) & 1);
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) a[kk] = a2[kk];
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int c = 32 * ct + r;
        const OutCol o = find_out(J, c);
        const float bias = (J.bias && c < J.n_out) ? J.bias[c] : 0.f;
        store_tile(acc[ct], o, bias, rows != nullptr, orow, valid, row0, h);
    }
}

// Deep k with a B that fits the LDS whole (K x 32 NCT floats <= 150 KB: the 25m model's dx = dT_0 W_cat is 576 x 64 =
// 147 KB): every workgroup copies B once, then its 16 waves walk 32-row tiles with NO barrier, like the persistent kernel
// below, but over k chunks of 64.  Same k order
// per output as gemm_deep_kernel.
// lean A fragment load for jobs with one input block and no edge-less-row substitution (the backward's products): no
// temporaries beyond the fragment itself
template <int KH>
__device__ __forceinline__ void load_a_plain(const GemmJob &J, int64_t srow, bool rv, int kbase, float (&a)[KH]) {
    // the two k-halves of a row interleave by float4 (kbase = chunk start + 4h, chunks 8 floats apart): one load
    // instruction reads 32 contiguous bytes per row instead of two 16-byte pieces 128+ bytes apart
    const float *p = J.A1 + srow * J.lda1;
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
        const int k = kbase + q * 8;
        const bool keep = rv && k < J.K1;
        const float4 t = ld4(p + (k < J.K1 ? k : 0));
        a[q * 4 + 0] = keep ? t.x : 0.f;
        a[q * 4 + 1] = keep ? t.y : 0.f;
        a[q * 4 + 2] = keep ? t.z : 0.f;
        a[q * 4 + 3] = keep ? t.w : 0.f;
    }
}

template <int NCT>
__global__ __launch_bounds__(1024) void gemm_deep_resident_kernel(const GemmBatch Bt, const int *__restrict__ rows, int64_t n_rows) {
    constexpr int KH = 32, W = 32 * NCT, NW = 16;   // 64-deep k chunks: 112 registers per lane, 16 waves per CU
    const GemmJob &J = Bt.j[blockIdx.y];
    if (J.rows) {
        rows = J.rows;
        n_rows = J.n_rows;
    }
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int K = J.K1 + J.K2;
    const int nkc = (K + 2 * KH - 1) / (2 * KH);
    for (int idx = tid; idx < K * (W / 4); idx += 1024) {   // the whole k-major B: K rows (a partial last chunk re-reads row K - 1
        const int k = idx / (W / 4), c = (idx % (W / 4)) * 4;   // against A values that are zero there)
        *reinterpret_cast<float4 *>(g_lds + (size_t)k * W + c) = c < J.ldb ? ld4(J.B + (size_t)k * J.ldb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int64_t n_tiles = (n_rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NW) {
        const int64_t row0 = tile * 32, grow = row0 + r;
        const bool rv = grow < n_rows;
        const int64_t srow = rv ? (rows ? (int64_t)rows[grow] : grow) : 0;
        int orow[16];
        const unsigned valid = load_orow(rows, row0, h, n_rows, orow);
        float a[KH];   // no second fragment buffer: 16 waves per CU cover the load latency, a 128-register budget does not fit one
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
        for (int kc = 0; kc < nkc; ++kc) {
            load_a_plain<KH>(J, srow, rv, kc * 2 * KH + 4 * h, a);   // a[4q + e] = k  kc * 64 + 4 (2q + h) + e
            const float *bs = g_lds + ((size_t)kc * 2 * KH + 4 * h) * W + r;   // image row of a[kl]: + 8 (kl / 4) + kl % 4
            const bool last_partial = (kc + 1) * 2 * KH > K;            // wave-uniform
            const int kbase = kc * 2 * KH + 4 * h;                      // image row of a[0] of this lane half
            // B operands in batches of 8 LDS reads per column tile, fenced: left alone, the scheduler hoists all 64 * NCT
            // reads of the chunk ahead of the MFMAs and spills
#pragma unroll
            for (int g = 0; g < KH / 8; ++g) {
                float bb[NCT][8];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int kl = 8 * g + u;   // row kc * 128 + h * 64 + kl of the image, clamped in a partial last chunk
                        const int ko = 8 * (kl / 4) + kl % 4;   // interleaved k order of the fragment
                        bb[ct][u] = last_partial ? g_lds[(size_t)min(kbase + ko, K - 1) * W + r + 32 * ct] : bs[ko * W + 32 * ct];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * g + u], bb[ct][u], acc[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int c = 32 * ct + r;
            const OutCol o = find_out(J, c);
            const float bias = (J.bias && c < J.n_out) ? J.bias[c] : 0.f;
            store_tile(acc[ct], o, bias, rows != nullptr, orow, valid, row0, h);
        }
    }
}

// Persistent variant (the default): every workgroup first copies the WHOLE k-major B of all its jobs into LDS
// (the 9 first-layer transforms of the MovieLens model are one 64 x 596 block = 149 KiB of the CU's 160 KiB), then its
// 16 waves walk (job, 32-row tile, column group) items with no barrier at all: A fragment from global, B fragment
// from LDS, 32 MFMAs per 32x32 output tile, stores straight from the accumulators.
constexpr int kColGroup = 5;       // 32-column tiles per item
struct PersistArgs {
    int n_items;                   // all jobs
    int col_group;                 // 32-column tiles per item
    int item_start[kMaxBatch + 1]; // first item of job j
    int lds_off[kMaxBatch];        // float offset of job j's B image, row stride lds_ld[j]
    int lds_ld[kMaxBatch];
    int n_tiles[kMaxBatch];        // 32-row tiles of job j (jobs may carry their own row lists)
};

template <int KH, bool LISTED>
__global__ __launch_bounds__(KH > 32 ? 512 : 1024) void gemm_persist_kernel(const GemmBatch Bt, const PersistArgs Pa, const int *__restrict__ rows,
                                                            int64_t n_rows) {
    constexpr int NT = KH > 32 ? 512 : 1024, NW = NT / 64;  // deeper k needs more registers per lane
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    // B images, 8 float4 loads in flight per thread
    for (int j = 0; j < Bt.n; ++j) {
        const GemmJob &J = Bt.j[j];
        const int K = J.K1 + J.K2, ld = Pa.lds_ld[j], q4 = ld / 4, total = 2 * KH * q4;
        const float inv_q4 = 1.0f / (float)q4;
        float4 *dst = reinterpret_cast<float4 *>(g_lds + Pa.lds_off[j]);
        constexpr int U = 8;
        for (int base = 0; base < total; base += NT * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT + tid;
                const int k = (int)(((float)idx + 0.5f) * inv_q4), c = (idx - k * q4) * 4;
                v[u] = (idx < total && k < K && c < J.ldb) ? ld4(J.B + (size_t)k * J.ldb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT + tid;
                if (idx < total) dst[idx] = v[u];
            }
        }
        // row 2*KH of the image: the bias, so the column loop below issues no global load at all (a load there
        // makes every tile wait for the previous tile's stores to be acknowledged)
        for (int c = tid; c < ld; c += NT) g_lds[Pa.lds_off[j] + 2 * KH * ld + c] = (J.bias && c < J.n_out) ? J.bias[c] : 0.f;
    }
    __syncthreads();
    const int stride = gridDim.x * NW;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // item, job and tile are wave-uniform: scalar registers
    for (int item = blockIdx.x * NW + wave; item < Pa.n_items; item += stride) {
        int j = 0;
        while (j + 1 < Bt.n && item >= Pa.item_start[j + 1]) ++j;
        const GemmJob &J = Bt.j[j];
        const int *jrows = LISTED ? (J.rows ? J.rows : rows) : nullptr;
        const int64_t jn = (LISTED && J.rows) ? J.n_rows : n_rows;
        const int local = item - Pa.item_start[j];
        const int n_tiles = Pa.n_tiles[j];
        const int tile = local % n_tiles, grp = local / n_tiles;  // consecutive waves -> consecutive row tiles
        const int64_t row0 = (int64_t)tile * 32, grow = row0 + r;
        const bool rv = grow < jn;
        const int64_t srow = rv ? ((LISTED && jrows) ? (int64_t)jrows[grow] : grow) : 0;
        // everything the column loop needs from the job, read once per item (scalar registers): the loop itself
        // then has no scalar-memory wait, which shares its counter with the LDS reads
        const int n_out = J.n_out, n_seg = J.n_seg;
        GemmSegment S[kMaxSegments];
#pragma unroll
        for (int sg = 0; sg < kMaxSegments; ++sg) {
            S[sg] = J.seg[sg < n_seg ? sg : 0];
            if (sg >= n_seg) S[sg].c1 = S[sg].c0;
        }
        float a[KH];
        load_a<KH, true>(J, srow, rv, 4 * h, a);
        int orow[16];
        unsigned valid = 0xffffu;
        if (LISTED || row0 + 32 > jn) valid = load_orow(jrows, row0, h, jn, orow);
        const int ld = Pa.lds_ld[j];
        // k order of a lane: float4 chunks 2q + h of the row (the two halves of a row interleave by 16 bytes, so one
        // load instruction reads 32 contiguous bytes per row); the B image is read with the same permutation
        const float *bimg = g_lds + Pa.lds_off[j] + (4 * h) * ld + r;
        const float *bias_img = g_lds + Pa.lds_off[j] + 2 * KH * ld + r;
        const int nct = (n_out + 31) / 32;
        const int ct_end = min(nct, (grp + 1) * Pa.col_group);
        // gated jobs (col_group <= 2 for them): the gate values of this lane's 16 rows x 2 column tiles are fetched here,
        // before the column loop, which must not issue a global load (see the bias row of the image)
        float gv[2][16];
        const bool gated = S[0].gate != nullptr;   // wave-uniform; a gated job has the gate on every segment
        if (gated) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int c = (grp * Pa.col_group + t) * 32 + r;
                const float *gp = nullptr;
                int ldg = 0;
#pragma unroll
                for (int sg = 0; sg < kMaxSegments; ++sg) {
                    const bool in = c >= S[sg].c0 && c < S[sg].c1;
                    gp = in ? S[sg].gate + (c - S[sg].c0) : gp;
                    ldg = in ? S[sg].ld_gate : ldg;
                }
                const bool col_ok = gp != nullptr && c < n_out && grp * Pa.col_group + t < ct_end;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int64_t rr = (LISTED || row0 + 32 > jn) ? (int64_t)orow[reg] : row0 + 4 * h + (reg & 3) + 8 * (reg >> 2);
                    const bool ok = col_ok && ((valid >> reg) & 1u);
                    gv[t][reg] = ok ? gp[(ok ? rr : 0) * ldg] : 1.f;
                }
            }
        }
        for (int ct = grp * Pa.col_group; ct < ct_end; ++ct) {
            const int col0 = ct * 32, c = col0 + r;
            const float bias = bias_img[col0];
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            // B fragment in batches of 8 LDS reads, the next batch issued before this batch's MFMAs
            float bb[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bb[0][u] = bimg[(8 * (u / 4) + u % 4) * ld + col0];
#pragma unroll
            for (int g = 0; g < KH / 8; ++g) {
                if (g + 1 < KH / 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) bb[(g + 1) & 1][u] = bimg[(16 * (g + 1) + 8 * (u / 4) + u % 4) * ld + col0];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * g + u], bb[g & 1][u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: lane (r, h) owns column c of rows (reg & 3) + 8 * (reg >> 2) + 4h.  Nothing in here waits on
            // memory, so the 16 stores go out back to back.
            float *dst = nullptr;
            int ldo = 0, relu = 0;
#pragma unroll
            for (int sg = 0; sg < kMaxSegments; ++sg) {
                const bool in = c >= S[sg].c0 && c < S[sg].c1;
                dst = in ? S[sg].dst + (c - S[sg].c0) : dst;
                ldo = in ? S[sg].ld : ldo;
                relu = in ? S[sg].relu : relu;
            }
            if (c >= n_out || !dst) continue;
            float v[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float t = acc[reg] + bias;
                v[reg] = relu ? fmaxf(t, 0.f) : t;
            }
            if (gated) {
                const int t = ct - grp * Pa.col_group;   // 0 or 1
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) v[reg] = (t == 0 ? gv[0][reg] : gv[1][reg]) > 0.f ? v[reg] : 0.f;
            }
            if (LISTED) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    if ((valid >> reg) & 1u) dst[(int64_t)orow[reg] * ldo] = v[reg];
            } else {
                float *p = dst + (row0 + 4 * h) * ldo;
                if (valid == 0xffffu) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) p[(int64_t)((reg & 3) + 8 * (reg >> 2)) * ldo] = v[reg];
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if ((valid >> reg) & 1u) p[(int64_t)((reg & 3) + 8 * (reg >> 2)) * ldo] = v[reg];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- narrow outputs
// Transforms with at most 16 output columns (the last layer of every PEA channel: hidden -> repr_dim = 16) are bound
// by how fast the [rows, K] input streams in, not by the matrix pipe, and a 32x32 tile would waste half of it.  One
// wave owns 64 rows: 4 independent v_mfma_f32_16x16x4_f32 chains (one per 16-row group) with the WEIGHTS as the row
// operand, so lane (j = lane % 16, q = lane / 16) ends up with row j's columns 4q..4q+3 -- one float4 store -- and its
// 16 float4 input loads (all four groups) are issued before anything waits on them: twice the bytes in flight per
// wave of the 32-row kernel.  k order per lane: float4 chunks 4i + q of the row (the four lanes of a row read 64
// contiguous bytes per instruction); the weight image in LDS is read with the same permutation.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
    // > 0: every job has this many 64-row tiles and item i is (tile i / n_jobs, job i % n_jobs): the jobs of a level read
    // adjacent column blocks of the SAME rows (one 256-byte piece each out of a 2304-byte row), so waves that run together
    // then sweep whole rows instead of nine strided passes over the table (PEA_SKINNY_JOBMAJOR=1: the old job-major order)
    int tiles_per_job;
    int n_items;
    int item_start[kMaxBatch + 1];  // first item (64-row tile) of job j
    int lds_off[kMaxBatch];         // float offset of job j's image: [KQ*16][16] weights + [16] bias
};

template <int KQ, bool LISTED>  // KQ float4 chunks per lane and 16-row group: K <= 16 * KQ
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const GemmBatch Bt, const SkinnyArgs Sa, const int *__restrict__ rows,
                                                          int64_t n_rows) {
    constexpr int KP = 16 * KQ, G = 4;
    const int tid = threadIdx.x, lane = tid & 63, jr = lane & 15, q = lane >> 4;
    for (int j = 0; j < Bt.n; ++j) {
        const GemmJob &J = Bt.j[j];
        const int K = J.K1 + J.K2;
        float *img = g_lds + Sa.lds_off[j];
        for (int idx = tid; idx < KP * 16; idx += 512) {
            const int k = idx >> 4, c = idx & 15;
            img[idx] = (k < K && c < J.n_out) ? J.B[(size_t)k * J.ldb + c] : 0.f;
        }
        if (tid < 16) img[KP * 16 + tid] = (J.bias && tid < J.n_out) ? J.bias[tid] : 0.f;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stride = gridDim.x * 8;
    for (int item = blockIdx.x * 8 + wave; item < Sa.n_items; item += stride) {
        int j = 0, tile;
        if (Sa.tiles_per_job > 0) {
            tile = item / Bt.n;
            j = item - tile * Bt.n;
        } else {
            while (j + 1 < Bt.n && item >= Sa.item_start[j + 1]) ++j;
            tile = item - Sa.item_start[j];
        }
        const GemmJob &J = Bt.j[j];
        const int *jrows = LISTED ? (J.rows ? J.rows : rows) : nullptr;
        const int64_t jn = (LISTED && J.rows) ? J.n_rows : n_rows;
        const int64_t row0 = (int64_t)tile * (16 * G);
        const int K = J.K1 + J.K2, K1 = J.K1, n_out = J.n_out, n_seg = J.n_seg;
        GemmSegment S[kMaxSegments];
#pragma unroll
        for (int sg = 0; sg < kMaxSegments; ++sg) {
            S[sg] = J.seg[sg < n_seg ? sg : 0];
            if (sg >= n_seg) S[sg].c1 = S[sg].c0;
        }
        // ---- input rows: all G * KQ float4 loads of the lane go out before the first use
        int64_t srow[G];
        bool rv[G], alt[G];
        float sc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int64_t grow = row0 + 16 * g + jr;
            rv[g] = grow < jn;
            srow[g] = rv[g] ? ((LISTED && jrows) ? (int64_t)jrows[grow] : grow) : 0;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            alt[g] = J.a1_mask != nullptr && rv[g] && J.a1_mask[srow[g]] != 0;
            sc[g] = 1.f;
            if (J.a1_scale) {
                const float di = J.a1_scale[srow[g]];
                sc[g] = alt[g] ? di * di : 1.f;
            }
        }
        float4 xa[G][KQ];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float *p1 = alt[g] ? J.a1_alt + srow[g] * J.lda_alt : J.A1 + srow[g] * J.lda1;
            const float *p2 = J.K2 > 0 ? J.A2 + srow[g] * J.lda2 : p1;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const int k = 16 * i + 4 * q;
                const bool ok = k < K;
                const float *p = (k < K1 || !ok) ? p1 + (ok ? k : 0) : p2 + (k - K1);
                xa[g][i] = ld4(p);
            }
        }
        float4 ab[KQ];  // bias of the edge-less-row transform (same for every row)
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const int k = 16 * i + 4 * q;
            ab[i] = (J.a1_mask != nullptr && J.a1_bias != nullptr && k < K1) ? ld4(J.a1_bias + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- weights of this job: lane (col = jr, q) holds W[16i + 4q + e][jr]
        const float *img = g_lds + Sa.lds_off[j];
        float w[KQ][4];
#pragma unroll
        for (int i = 0; i < KQ; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[i][e] = img[(16 * i + 4 * q + e) * 16 + jr];
        const float4 bias = *reinterpret_cast<const float4 *>(img + KP * 16 + 4 * q);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const int k = 16 * i + 4 * q;
                float4 t = xa[g][i];
                if (alt[g] && k < K1) {  // same roundings as the aggregation kernel's finish_row: product, + bias, relu
                    t = make_float4(fmaxf(sc[g] * t.x + ab[i].x, 0.f), fmaxf(sc[g] * t.y + ab[i].y, 0.f),
                                    fmaxf(sc[g] * t.z + ab[i].z, 0.f), fmaxf(sc[g] * t.w + ab[i].w, 0.f));
                }
                const bool keep = rv[g] && k < K;
                xa[g][i] = keep ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        f32x4 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][0], xa[g][i].x, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][1], xa[g][i].y, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][2], xa[g][i].z, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][3], xa[g][i].w, acc[g], 0, 0, 0);
        }
        // ---- epilogue: columns c .. c+3 of row srow[g]
        const int c = 4 * q;
        float *dst = nullptr;
        int ldo = 0, relu = 0;
#pragma unroll
        for (int sg = 0; sg < kMaxSegments; ++sg) {
            const bool in = c >= S[sg].c0 && c < S[sg].c1;
            dst = in ? S[sg].dst + (c - S[sg].c0) : dst;
            ldo = in ? S[sg].ld : ldo;
            relu = in ? S[sg].relu : relu;
        }
        if (c >= n_out || !dst) continue;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4 v = make_float4(acc[g][0] + bias.x, acc[g][1] + bias.y, acc[g][2] + bias.z, acc[g][3] + bias.w);
            if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (rv[g]) *reinterpret_cast<float4 *>(dst + srow[g] * ldo) = v;
        }
    }
}

// a job qualifies when its output is at most 16 columns wide, k fits 128 and every segment can take float4 stores
static bool skinny_ok(const GemmJob &J) {
    if (J.no_narrow || J.seg[0].gate || J.n_out > 16 || J.K1 + J.K2 > 128 || J.ldb < J.n_out) return false;
    for (int sg = 0; sg < J.n_seg; ++sg) {
        const GemmSegment &S = J.seg[sg];
        if (S.c0 % 4 || S.c1 % 4 || S.ld % 4 || (reinterpret_cast<uintptr_t>(S.dst) & 15)) return false;
    }
    return true;
}

template <int KQ, bool LISTED>
static int launch_skinny_v(const GemmBatch &Bt, const SkinnyArgs &Sa, size_t lds, int grid, const int *rows, int64_t n_rows,
                           double bytes, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        PEA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_skinny_kernel<KQ, LISTED>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attr_set = true;
    }
    ProfScope ps("gemm_mfma_narrow", stream, bytes);
    PEA_LAUNCH((gemm_skinny_kernel<KQ, LISTED>), dim3((unsigned)grid), dim3(512), lds, stream, Bt, Sa, rows, n_rows);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

template <int KQ>
static int launch_skinny(const GemmBatch &Bt, const int *rows, int64_t n_rows, double bytes, hipStream_t stream) {
    SkinnyArgs Sa;
    int off = 0, items = 0;
    bool listed = rows != nullptr;
    for (int j = 0; j < Bt.n; ++j) {
        listed = listed || Bt.j[j].rows != nullptr;
        Sa.lds_off[j] = off;
        off += (16 * KQ + 1) * 16;
        Sa.item_start[j] = items;
        items += (int)(((Bt.j[j].rows ? Bt.j[j].n_rows : n_rows) + 63) / 64);
    }
    Sa.item_start[Bt.n] = items;
    Sa.n_items = items;
    Sa.tiles_per_job = 0;
    {
        bool same = Bt.n > 1;
        for (int j = 1; j < Bt.n; ++j) same = same && (Sa.item_start[j + 1] - Sa.item_start[j]) == Sa.item_start[1];
        const char *env = getenv("PEA_SKINNY_JOBMAJOR");
        if (same && !(env && atoi(env) != 0)) Sa.tiles_per_job = Sa.item_start[1];
    }
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        int dev = 0;
        PEA_HIP(hipGetDevice(&dev));
        PEA_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
    }
    const int grid = std::min(n_cu * 2, (items + 7) / 8);
    const size_t lds = (size_t)off * sizeof(float);
    return listed ? launch_skinny_v<KQ, true>(Bt, Sa, lds, grid, rows, n_rows, bytes, stream)
                  : launch_skinny_v<KQ, false>(Bt, Sa, lds, grid, rows, n_rows, bytes, stream);
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
        for (int o = tid; o < J.HF; o += 256) {
            // log2(e) folded in: the aggregation kernels keep logits in the log2 domain (leaky_relu commutes with a
            // positive scale), so every softmax weight is a single v_exp_f32
            J.att_dst[o] = J.w1[o];  // att_i multiplies the TARGET row (natural units: the exponent takes log2(e), agg_common.h)
            J.att_src[o] = J.w2[o];  // att_j multiplies the SOURCE row
        }
    } else if (J.kind == PEA_KIND_GCN) {
        for (int idx = tid; idx < J.in * J.HF; idx += 256) {
            const int k = idx / J.HF, o = idx % J.HF;
            J.B[(size_t)k * J.ldb + o] = J.w0[idx];
        }
    } else if (J.kind == PEA_PACK_SAGE2) {  // two k-major blocks: lin_rel^T (the gather source's transform), lin_root^T
        for (int idx = tid; idx < J.HF * J.in; idx += 256) {
            const int o = idx / J.in, k = idx % J.in;
            J.B[(size_t)k * J.ldb + o] = J.w0[idx];
            J.B2[(size_t)k * J.ldb2 + o] = J.w1[idx];
        }
        for (int idx = tid; idx < J.in * (J.ldb2 - J.HF); idx += 256) {
            const int k = idx / (J.ldb2 - J.HF), c = idx % (J.ldb2 - J.HF);
            J.B2[(size_t)k * J.ldb2 + J.HF + c] = 0.f;
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

static int check_job(const GemmJob &job) {
    PEA_REQUIRE(job.K1 > 0 && job.K1 % 4 == 0 && job.K2 % 4 == 0, PEA_ERR_ARG,
                "gemm: input widths (%d, %d) must be multiples of 4", job.K1, job.K2);
    PEA_REQUIRE(job.n_out > 0 && job.ldb >= job.n_out, PEA_ERR_ARG, "gemm: output width %d / ldb %d", job.n_out, job.ldb);
    PEA_REQUIRE(job.lda1 % 4 == 0 && (job.K2 == 0 || job.lda2 % 4 == 0), PEA_ERR_ARG,
                "gemm: input row strides must be multiples of 4 floats");
    PEA_REQUIRE(job.n_seg > 0 && job.n_seg <= kMaxSegments, PEA_ERR_ARG, "gemm: %d segments", job.n_seg);
    return PEA_OK;
}

constexpr size_t kLdsBudget = 160 * 1024 - 1024;  // dynamic LDS a workgroup may claim (one workgroup per CU)

template <int KH, bool LISTED>
int launch_persist_v(const GemmBatch &Bt, const PersistArgs &Pa, size_t lds, int grid, const int *rows, int64_t n_rows,
                     double bytes, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        PEA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_persist_kernel<KH, LISTED>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        attr_set = true;
    }
    constexpr int NT = KH > 32 ? 512 : 1024;
    ProfScope ps(Bt.n == 1 ? "gemm_mfma_shared" : "gemm_mfma_batch", stream, bytes);
    PEA_LAUNCH((gemm_persist_kernel<KH, LISTED>), dim3((unsigned)grid), dim3(NT), lds, stream, Bt, Pa, rows, n_rows);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

template <int KH>
int launch_persist(const GemmBatch &Bt, const int *rows, int64_t n_rows, double bytes, hipStream_t stream) {
    PersistArgs Pa;
    Pa.col_group = kColGroup;
    for (int j = 0; j < Bt.n; ++j)
        if (Bt.j[j].seg[0].gate) Pa.col_group = 2;   // gated epilogues hold the gate values of two column tiles
    int off = 0, items = 0;
    bool listed = rows != nullptr;
    for (int j = 0; j < Bt.n; ++j) {
        const int nct = (Bt.j[j].n_out + 31) / 32;
        const int n_tiles = (int)(((Bt.j[j].rows ? Bt.j[j].n_rows : n_rows) + 31) / 32);
        listed = listed || Bt.j[j].rows != nullptr;
        Pa.n_tiles[j] = n_tiles;
        Pa.lds_ld[j] = nct * 32;
        Pa.lds_off[j] = off;
        off += (2 * KH + 1) * Pa.lds_ld[j];
        Pa.item_start[j] = items;
        items += n_tiles * ((nct + Pa.col_group - 1) / Pa.col_group);
    }
    Pa.item_start[Bt.n] = items;
    Pa.n_items = items;
    const size_t lds = (size_t)off * sizeof(float);
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        int dev = 0;
        PEA_HIP(hipGetDevice(&dev));
        PEA_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
    }
    const int per_cu = lds * 2 <= kLdsBudget ? 2 : 1;  // two workgroups share a CU when their B images both fit
    constexpr int NT = KH > 32 ? 512 : 1024;
    const int grid = std::min(n_cu * per_cu, (items + NT / 64 - 1) / (NT / 64));
    return listed ? launch_persist_v<KH, true>(Bt, Pa, lds, grid, rows, n_rows, bytes, stream)
                  : launch_persist_v<KH, false>(Bt, Pa, lds, grid, rows, n_rows, bytes, stream);
}

// Jobs of one call share the row set; jobs with the same k-depth class go out as one launch.  A job whose B does
// not fit the LDS budget is cut into column chunks; k deeper than 128 falls back to the staged kernel.
int launch_gemm_batch(const GemmJob *jobs_in, int n_jobs_in, const int *rows, int64_t n_rows, hipStream_t stream) {
    if (n_jobs_in <= 0) return PEA_OK;
    std::vector<GemmJob> jobs;
    for (int i = 0; i < n_jobs_in; ++i) {   // a gated epilogue exists in the persistent kernel only (k <= 128)
        bool any = false, all = true;
        for (int sg = 0; sg < jobs_in[i].n_seg; ++sg) {
            any = any || jobs_in[i].seg[sg].gate != nullptr;
            all = all && jobs_in[i].seg[sg].gate != nullptr;
        }
        PEA_REQUIRE(!any || (all && jobs_in[i].K1 + jobs_in[i].K2 <= 128), PEA_ERR_ARG,
                    "gemm: a gated job needs the gate on every segment and k <= 128");
    }
    // narrow outputs first: one launch per k class
    for (int kq : {2, 4, 8}) {
        GemmBatch Bt;
        Bt.n = 0;
        double bytes = 0.0;
        auto flush = [&]() -> int {
            if (Bt.n == 0) return PEA_OK;
            int rc = kq == 2 ? launch_skinny<2>(Bt, rows, n_rows, bytes, stream)
                     : kq == 4 ? launch_skinny<4>(Bt, rows, n_rows, bytes, stream)
                               : launch_skinny<8>(Bt, rows, n_rows, bytes, stream);
            Bt.n = 0;
            bytes = 0.0;
            return rc;
        };
        for (int i = 0; i < n_jobs_in; ++i) {
            const GemmJob &J = jobs_in[i];
            const int K = J.K1 + J.K2;
            if (!skinny_ok(J) || (K <= 32 ? 2 : K <= 64 ? 4 : 8) != kq) continue;
            PEA_TRY(check_job(J));
            const int64_t nr = J.rows ? J.n_rows : n_rows;
            if (nr <= 0) continue;
            Bt.j[Bt.n++] = J;
            bytes += 4.0 * (double)nr * (K + J.n_out);
            if (Bt.n == kMaxBatch) PEA_TRY(flush());
        }
        PEA_TRY(flush());
    }
    for (int i = 0; i < n_jobs_in; ++i) {
        if (skinny_ok(jobs_in[i])) continue;
        PEA_TRY(check_job(jobs_in[i]));
        const GemmJob &J = jobs_in[i];
        if ((J.rows ? J.n_rows : n_rows) <= 0) continue;
        const int K = J.K1 + J.K2;
        const int KH = K <= 32 ? 16 : K <= 64 ? 32 : 64;
        const int max_cols = (int)(kLdsBudget / sizeof(float) / (size_t)(2 * KH + 1)) / 32 * 32;
        if (K > 128 || J.n_out <= max_cols) {
            jobs.push_back(J);
            continue;
        }
        for (int c0 = 0; c0 < J.n_out; c0 += max_cols) {  // column chunks of an oversize job
            GemmJob C = J;
            const int c1 = std::min(J.n_out, c0 + max_cols);
            C.B = J.B + c0;
            C.n_out = c1 - c0;
            C.bias = J.bias ? J.bias + c0 : nullptr;
            C.n_seg = 0;
            for (int sg = 0; sg < J.n_seg; ++sg) {
                const int a0 = std::max(J.seg[sg].c0, c0), a1 = std::min(J.seg[sg].c1, c1);
                if (a1 <= a0) continue;
                GemmSegment S = J.seg[sg];
                S.dst = J.seg[sg].dst + (a0 - J.seg[sg].c0);
                if (S.gate) S.gate = J.seg[sg].gate + (a0 - J.seg[sg].c0);
                S.c0 = a0 - c0;
                S.c1 = a1 - c0;
                C.seg[C.n_seg++] = S;
            }
            if (C.n_seg) jobs.push_back(C);
        }
    }
    const int n_jobs = (int)jobs.size();
    const int classes[3] = {16, 32, 64};
    for (int ci = 0; ci < 3; ++ci) {
        for (int deep = 0; deep < 2; ++deep) {  // deep: K > 2*KH, staged kernel
            GemmBatch Bt;
            Bt.n = 0;
            double bytes = 0.0;
            size_t lds = 0;
            auto flush = [&]() -> int {
                if (Bt.n == 0) return PEA_OK;
                int rc = PEA_OK;
                if (!deep) {
                    switch (classes[ci]) {
                        case 16: rc = launch_persist<16>(Bt, rows, n_rows, bytes, stream); break;
                        case 32: rc = launch_persist<32>(Bt, rows, n_rows, bytes, stream); break;
                        default: rc = launch_persist<64>(Bt, rows, n_rows, bytes, stream); break;
                    }
                } else {
                    int64_t max_rows = n_rows;
                    for (int q = 0; q < Bt.n; ++q) max_rows = std::max<int64_t>(max_rows, Bt.j[q].rows ? Bt.j[q].n_rows : 0);
                    dim3 grid((unsigned)((max_rows + 127) / 128), (unsigned)Bt.n);
                    ProfScope ps("gemm_mfma_deep", stream, bytes);
                    int max_out = 0;
                    for (int q = 0; q < Bt.n; ++q) max_out = std::max(max_out, Bt.j[q].n_out);
                    const char *env = getenv("PEA_DEEP_STAGED");   // A/B switch: the per-column-tile staged kernel
                    int max_k = 0;
                    for (int q = 0; q < Bt.n; ++q) max_k = std::max(max_k, Bt.j[q].K1 + Bt.j[q].K2);
                    const int nct_r = max_out <= 32 ? 1 : max_out <= 64 ? 2 : 4;
                    const size_t lds_res = (size_t)max_k * 32 * nct_r * sizeof(float);
                    bool plain = true;   // one input block, no edge-less-row substitution: what the resident kernel's loader takes
                    for (int q = 0; q < Bt.n; ++q) plain = plain && Bt.j[q].K2 == 0 && Bt.j[q].a1_mask == nullptr;
                    if (plain && max_out <= 64 && lds_res <= kLdsBudget && !(env && atoi(env) != 0)) {   // (4 column tiles would spill)
                        // B resident in LDS, no barriers: persistent workgroups of 8 waves
                        static bool attr_res = false;
                        if (!attr_res) {
                            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_deep_resident_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
                            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_deep_resident_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
                            attr_res = true;
                        }
                        hipDeviceProp_t prop;
                        int dev = 0;
                        static int n_cu_deep = 0;
                        if (!n_cu_deep && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu_deep = prop.multiProcessorCount;
                        const int64_t tiles = (max_rows + 31) / 32;
                        dim3 grid_r((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_cu_deep > 0 ? n_cu_deep : 256, (tiles + 15) / 16)), (unsigned)Bt.n);
                        if (nct_r == 1) PEA_LAUNCH(gemm_deep_resident_kernel<1>, grid_r, dim3(1024), lds_res, stream, Bt, rows, n_rows);
                        else PEA_LAUNCH(gemm_deep_resident_kernel<2>, grid_r, dim3(1024), lds_res, stream, Bt, rows, n_rows);
                    } else if (max_out <= 128 && !(env && atoi(env) != 0)) {
                        const int nct = max_out <= 32 ? 1 : max_out <= 64 ? 2 : 4;
                        const size_t lds_deep = (size_t)2 * 128 * 32 * nct * sizeof(float);
                        static bool attr_set = false;
                        if (!attr_set) {
                            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_deep_kernel<2>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 64 * 4);
                            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_deep_kernel<4>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 128 * 4);
                            attr_set = true;
                        }
                        if (nct == 1) PEA_LAUNCH(gemm_deep_kernel<1>, grid, dim3(256), lds_deep, stream, Bt, rows, n_rows);
                        else if (nct == 2) PEA_LAUNCH(gemm_deep_kernel<2>, grid, dim3(256), lds_deep, stream, Bt, rows, n_rows);
                        else PEA_LAUNCH(gemm_deep_kernel<4>, grid, dim3(256), lds_deep, stream, Bt, rows, n_rows);
                    } else {
                        PEA_LAUNCH(gemm_mfma_kernel<64>, grid, dim3(256), 0, stream, Bt, rows, n_rows);
                    }
                    if (hipGetLastError() != hipSuccess) rc = PEA_ERR_HIP;
                }
                Bt.n = 0;
                bytes = 0.0;
                lds = 0;
                return rc;
            };
            for (int i = 0; i < n_jobs; ++i) {
                const int K = jobs[i].K1 + jobs[i].K2;
                const int cls = K <= 32 ? 16 : K <= 64 ? 32 : 64;
                if (cls != classes[ci] || (K > 128) != (deep == 1)) continue;
                const size_t need = (size_t)(2 * cls + 1) * ((jobs[i].n_out + 31) / 32 * 32) * sizeof(float);
                if (!deep && Bt.n > 0 && lds + need > kLdsBudget) PEA_TRY(flush());
                Bt.j[Bt.n++] = jobs[i];
                lds += need;
                bytes += 4.0 * (double)(jobs[i].rows ? jobs[i].n_rows : n_rows) * (K + jobs[i].n_out);
                if (Bt.n == kMaxBatch) PEA_TRY(flush());
            }
            PEA_TRY(flush());
        }
    }
    return PEA_OK;
}

int launch_gemm(const GemmJob &job, const int *rows, int64_t n_rows, hipStream_t stream) {
    return launch_gemm_batch(&job, 1, rows, n_rows, stream);
}

int launch_pack(const PackJob *jobs, int n_jobs, hipStream_t stream) {
    for (int base = 0; base < n_jobs; base += 24) {
        PackLaunch L;
        L.n = n_jobs - base < 24 ? n_jobs - base : 24;
        for (int i = 0; i < L.n; ++i) L.j[i] = jobs[base + i];
        ProfScope ps("pack_weights", stream);
        PEA_LAUNCH(pack_kernel, dim3(L.n), dim3(256), 0, stream, L);
        PEA_HIP(hipGetLastError());
    }
    return PEA_OK;
}

}  // namespace pea
