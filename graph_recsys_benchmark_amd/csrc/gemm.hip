// Dense per-layer transforms of the PEA path for gfx950: h = x.W^T (GAT), x.W (GCN),
// lin_rel(mean) + lin_root(x) (SAGE) -- the torch.nn.Linear / matmul calls inside the PyG convs
// (reference call site graph_recsys_benchmark/models/base.py:138-139, SURVEY.md 2.1 "dense transform").
//
// The weights of every layer are re-packed each forward (they change every optimizer step) into
// k-major blocks  B[k][col]  so one GEMM job can serve several channels that share the same input
// (all P first-layer channels read the same x, models/base.py:192-193).
//
// f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32, bitwise a k-ordered fmaf chain): one wave owns a 32-row
// tile, keeps its A fragment in registers (lane (r, h) holds A[row r][k in h*KH .. h*KH+KH) -- the k order is
// permuted identically on the B side, which a sum over k does not care about) and walks the job's 32-column tiles,
// streaming the k-major B columns from L2.  No LDS, no barriers; 4 independent waves per block.
#include <algorithm>

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
template <int KH>
__device__ __forceinline__ void load_a(const GemmJob &J, int64_t srow, bool rv, int kbase, float (&a)[KH]) {
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
        const int k = kbase + q * 4;
        const bool ok = k < K;
        const float *p = (k < J.K1 || !ok) ? p1 + (ok ? k : 0) : p2 + (k - J.K1);
        v[q] = ld4(p);
        bv[q] = (J.a1_mask != nullptr && k < J.K1) ? ld4(J.a1_bias + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
        const int k = kbase + q * 4;
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
        orow[reg] = (rows && g < n_rows) ? rows[g] : 0;
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

// Persistent variant (the default): every workgroup first copies the WHOLE k-major B of all its jobs into LDS
// (the 9 first-layer transforms of the MovieLens model are one 64 x 596 block = 149 KiB of the CU's 160 KiB), then its
// 16 waves walk (job, 32-row tile, column group) items with no barrier at all: A fragment from global, B fragment
// from LDS, 32 MFMAs per 32x32 output tile, stores straight from the accumulators.
constexpr int kColGroup = 5;       // 32-column tiles per item
struct PersistArgs {
    int n_items;                   // all jobs
    int item_start[kMaxBatch + 1]; // first item of job j
    int lds_off[kMaxBatch];        // float offset of job j's B image, row stride lds_ld[j]
    int lds_ld[kMaxBatch];
    int n_tiles[kMaxBatch];        // 32-row tiles of job j (jobs may carry their own row lists)
};

extern __shared__ float g_lds[];

template <int KH>
__global__ __launch_bounds__(KH > 32 ? 512 : 1024) void gemm_persist_kernel(const GemmBatch Bt, const PersistArgs Pa, const int *__restrict__ rows,
                                                            int64_t n_rows) {
    constexpr int NT = KH > 32 ? 512 : 1024, NW = NT / 64;  // deeper k needs more registers per lane
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int j = 0; j < Bt.n; ++j) {
        const GemmJob &J = Bt.j[j];
        const int K = J.K1 + J.K2, ld = Pa.lds_ld[j], q4 = ld / 4;
        float *dst = g_lds + Pa.lds_off[j];
        for (int idx = tid; idx < 2 * KH * q4; idx += NT) {
            const int k = idx / q4, c = (idx % q4) * 4;
            const float4 v = (k < K && c < J.ldb) ? ld4(J.B + (size_t)k * J.ldb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(dst + k * ld + c) = v;
        }
    }
    __syncthreads();
    const int stride = gridDim.x * NW;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // item, job and tile are wave-uniform: scalar loads
    for (int item = blockIdx.x * NW + wave; item < Pa.n_items; item += stride) {
        int j = 0;
        while (j + 1 < Bt.n && item >= Pa.item_start[j + 1]) ++j;
        const GemmJob &J = Bt.j[j];
        const int *jrows = J.rows ? J.rows : rows;
        const int64_t jn = J.rows ? J.n_rows : n_rows;
        const int local = item - Pa.item_start[j];
        const int n_tiles = Pa.n_tiles[j];
        const int tile = local % n_tiles, grp = local / n_tiles;  // consecutive waves -> consecutive row tiles
        const int64_t row0 = (int64_t)tile * 32, grow = row0 + r;
        const bool rv = grow < jn;
        const int64_t srow = rv ? (jrows ? (int64_t)jrows[grow] : grow) : 0;
        float a[KH];
        load_a<KH>(J, srow, rv, h * KH, a);
        int orow[16];
        const unsigned valid = load_orow(jrows, row0, h, jn, orow);
        const int ld = Pa.lds_ld[j];
        const float *bimg = g_lds + Pa.lds_off[j] + (h * KH) * ld + r;
        const int nct = (J.n_out + 31) / 32;
        const int ct_end = min(nct, (grp + 1) * kColGroup);
        for (int ct = grp * kColGroup; ct < ct_end; ++ct) {
            const int col0 = ct * 32, c = col0 + r;
            const OutCol o = find_out(J, c);
            const float bias = (J.bias && c < J.n_out) ? J.bias[c] : 0.f;  // in flight during the MFMAs
            float b[KH];
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) b[kk] = bimg[kk * ld + col0];
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
            store_tile(acc, o, bias, jrows != nullptr, orow, valid, row0, h);
        }
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
        for (int o = tid; o < J.HF; o += 256) {
            // log2(e) folded in: the aggregation kernels keep logits in the log2 domain (leaky_relu commutes with a
            // positive scale), so every softmax weight is a single v_exp_f32
            J.att_dst[o] = J.w1[o] * 1.44269504088896340736f;  // att_i multiplies the TARGET row
            J.att_src[o] = J.w2[o] * 1.44269504088896340736f;  // att_j multiplies the SOURCE row
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

template <int KH>
int launch_persist(const GemmBatch &Bt, const int *rows, int64_t n_rows, double bytes, hipStream_t stream) {
    PersistArgs Pa;
    int off = 0, items = 0;
    for (int j = 0; j < Bt.n; ++j) {
        const int nct = (Bt.j[j].n_out + 31) / 32;
        const int n_tiles = (int)(((Bt.j[j].rows ? Bt.j[j].n_rows : n_rows) + 31) / 32);
        Pa.n_tiles[j] = n_tiles;
        Pa.lds_ld[j] = nct * 32;
        Pa.lds_off[j] = off;
        off += 2 * KH * Pa.lds_ld[j];
        Pa.item_start[j] = items;
        items += n_tiles * ((nct + kColGroup - 1) / kColGroup);
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
    static bool attr_set = false;
    if (!attr_set) {
        PEA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_persist_kernel<KH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        attr_set = true;
    }
    const int per_cu = lds * 2 <= kLdsBudget ? 2 : 1;  // two workgroups share a CU when their B images both fit
    constexpr int NT = KH > 32 ? 512 : 1024;
    const int grid = std::min(n_cu * per_cu, (items + NT / 64 - 1) / (NT / 64));
    ProfScope ps(Bt.n == 1 ? "gemm_mfma_shared" : "gemm_mfma_batch", stream, bytes);
    hipLaunchKernelGGL(gemm_persist_kernel<KH>, dim3((unsigned)grid), dim3(NT), lds, stream, Bt, Pa, rows, n_rows);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

// Jobs of one call share the row set; jobs with the same k-depth class go out as one launch.  A job whose B does
// not fit the LDS budget is cut into column chunks; k deeper than 128 falls back to the staged kernel.
int launch_gemm_batch(const GemmJob *jobs_in, int n_jobs_in, const int *rows, int64_t n_rows, hipStream_t stream) {
    if (n_jobs_in <= 0) return PEA_OK;
    std::vector<GemmJob> jobs;
    for (int i = 0; i < n_jobs_in; ++i) {
        PEA_TRY(check_job(jobs_in[i]));
        const GemmJob &J = jobs_in[i];
        if ((J.rows ? J.n_rows : n_rows) <= 0) continue;
        const int K = J.K1 + J.K2;
        const int KH = K <= 32 ? 16 : K <= 64 ? 32 : 64;
        const int max_cols = (int)(kLdsBudget / sizeof(float) / (size_t)(2 * KH)) / 32 * 32;
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
                    hipLaunchKernelGGL(gemm_mfma_kernel<64>, grid, dim3(256), 0, stream, Bt, rows, n_rows);
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
                const size_t need = (size_t)2 * cls * ((jobs[i].n_out + 31) / 32 * 32) * sizeof(float);
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
        hipLaunchKernelGGL(pack_kernel, dim3(L.n), dim3(256), 0, stream, L);
        PEA_HIP(hipGetLastError());
    }
    return PEA_OK;
}

}  // namespace pea
