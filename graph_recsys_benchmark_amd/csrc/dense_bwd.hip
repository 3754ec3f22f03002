// Dense half of the PEA backward for gfx950 (reference solvers.py:213-216 loss.backward()):
//   pea_grad_weight   dW = A^T B over the node dimension (torch.nn.Linear / matmul weight gradients of every conv layer:
//                     [HF, N] x [N, in]) -- a few thousand outputs, a reduction over 10^5..10^7 rows.  BLAS treats it as a
//                     tall-k GEMM and is 10-25x off its memory bound on these shapes (rocBLAS: 0.52 ms per [16,N]x[N,64]
//                     on 273,744 rows, 87 MB of input); here the rows are cut into parts, every workgroup streams its part
//                     once through v_mfma_f32_16x16x4_f32 (the reduction index IS the MFMA k, so both operands are read
//                     row-contiguous with no transpose), and the per-part tiles are summed in a fixed order (no atomics).
//   pea_dense_batch   out = A W for a batch of independent jobs (the input gradients dIn = dT W of one level in one
//                     launch), on the forward transform kernels (gemm.hip).
#include <algorithm>
#include <vector>

#include <type_traits>

#include "common.h"

namespace pea {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGwRecords = 4096;  // (job, part) records of 16 x 256 floats in the workspace
constexpr int kGwMaxJobs = 40;    // 64x64 blocks per launch (the job table travels as a kernel argument: < 4 KB)
constexpr int kGwMaxParts = 512;  // row parts per job (grid.x), chosen per launch: gw_parts()
struct GwJob {
    const float *a, *b;
    const unsigned char *b_mask;   // optional [rows]: rows flagged here take their b operand from b_alt (pea_gw_job)
    const float *b_alt;
    const float *b_alt_scale;      // optional per-row factor on the alternative row
    int64_t ldb_alt;
    int64_t lda, ldb;
    int ma, nb;       // valid columns of this block (<= 16 * MT, <= 16 * NT)
    float *out;       // block origin, row stride ldo
    int64_t ldo;
};
struct GwBatch {
    int n;
    GwJob j[kGwMaxJobs];
};

// stage 1: partial[job][part][MT*NT tiles][256] = sum over the part's rows of a[n][i] * b[n][j]
template <int MT, int NT>
__global__ __launch_bounds__(256) void gw_stage1(const GwBatch Jb, const RowMap M, float *__restrict__ partial) {
    const int64_t n_rows = M.size();   // sharded plans reduce over the rows this rank owns (RowMap), else over all rows
    __shared__ float red[3][MT * NT * 256];
    // grid (jobs, parts): the workgroups of one row part -- they read neighbouring column blocks of the SAME rows -- are
    // dispatched together
    const int job = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
    const GwJob &J = Jb.j[job];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int64_t chunk = ((n_rows + parts - 1) / parts + 15) / 16 * 16;
    const int64_t r0 = (int64_t)part * chunk, r1 = min(n_rows, r0 + chunk);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool am[MT], bm[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) am[mt] = 16 * mt + i < J.ma;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bm[nt] = 16 * nt + i < J.nb;
    constexpr int U = 4;  // row groups in flight per wave (each: 4 rows, one dword per operand tile and lane)
    for (int64_t n0 = r0 + 4 * wave; n0 < r1; n0 += 16 * U) {
        float av[U][MT], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t n = n0 + 16 * u + kq;
            const int64_t nr = n < r1 ? M.row(n) : 0;
            const bool ok = n < r1 && nr < M.N;
            const int64_t nc = ok ? nr : 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[u][mt] = (ok && am[mt]) ? J.a[nc * J.lda + 16 * mt + i] : 0.f;
            const bool alt = J.b_mask && J.b_mask[nc] != 0;
            const float *brow = alt ? J.b_alt + nc * J.ldb_alt : J.b + nc * J.ldb;
            const float bs = (alt && J.b_alt_scale) ? J.b_alt_scale[nc] : 1.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[u][nt] = (ok && bm[nt]) ? bs * brow[16 * nt + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt], bv[u][nt], acc[mt][nt], 0, 0, 0);
    }
    // waves 1..3 park their tiles in LDS, wave 0 adds them in wave order and writes the part's record
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[wave - 1][((mt * NT + nt) * 4 + v) * 64 + lane] = acc[mt][nt][v];
    }
    __syncthreads();
    if (wave == 0) {
        float *dst = partial + ((size_t)job * parts + part) * (MT * NT * 256);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int e = ((mt * NT + nt) * 4 + v) * 64 + lane;
                    dst[e] = ((acc[mt][nt][v] + red[0][e]) + red[1][e]) + red[2][e];
                }
    }
}

// stage 1 for full 64 x 64 blocks (the first layer's dW = dT_0^T x: nine of them): the dword loads above fetch 64-byte
// pieces (one column per lane) -- 4.9 M load instructions for 1.26 GB.  Here the workgroup streams its row part in chunks
// of 32 rows with float4 loads (a whole 256-byte row piece per 16 lanes) into a double-buffered LDS image, and wave w
// computes the four 16 x 16 tiles of tile row w from it (operands read back as dwords: row stride 80 floats = two lanes
// per bank, the minimum for 64 lanes); no cross-wave reduction, the record layout of stage 2 is unchanged.  Rows are
// added in the same order as above within a part (the k index of the MFMA steps runs over the rows in order), so the
// results agree with the dword version bit for bit per tile... up to which wave owned which rows there: that version
// summed four interleaved row subsets and added them; this one adds all rows of the part in order.
// LDS images: 64 floats per row, no padding; row r is stored ROTATED by 16 (r & 3) floats, so the four rows a wave's operand
// read touches (lanes (kq, i): row 4s + kq, column c0 + i) fall into four different 16-bank groups of the 64 banks.  A
// padded stride of 72 floats (round 3's first version) put them 8 banks apart: SQ_LDS_BANK_CONFLICT = 42 % of the LDS-active
// cycles (profiles/r03/sq_gw_r03.txt).  32 KB per workgroup.
constexpr int kGwLd = 64;
__device__ __forceinline__ int gw_swz(int r, int c) { return (c + 16 * (r & 3)) & 63; }
__global__ __launch_bounds__(256) void gw_stage1_lds(const GwBatch Jb, const RowMap M, float *__restrict__ partial) {
    const int64_t n_rows = M.size();
    __shared__ float As[2][32][kGwLd], Bs[2][32][kGwLd];
    const int job = blockIdx.x, part = blockIdx.y, parts = gridDim.y;   // (see gw_stage1)
    const GwJob &J = Jb.j[job];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int64_t chunk = ((n_rows + parts - 1) / parts + 15) / 16 * 16;
    const int64_t r0 = (int64_t)part * chunk, r1 = min(n_rows, r0 + chunk);
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (two accumulator sets -- even / odd k-steps, 8 independent MFMA chains per wave instead of 4 -- measured SLOWER: 0.107 ->
    // 0.151 ms on the nine first-layer blocks; so did whole 128 x 128 blocks; bank conflicts, prefetch depth and residency were
    // each ruled out by a measurement: profiles/r03/sq_gw_r03.txt, gw_bench_r03.txt)
    // loader role: thread t moves float4 column (t % 16) of rows (t / 16) and (t / 16) + 16 of a 32-row chunk, both operands.
    // Software pipeline, per chunk c: row ids + operand flags (c - 3 .. c - 2: one step ahead of the loads they steer -- a
    // flag or a listed row id read in the same step sat in front of every operand load: 0.56 -> 0.84 ms on the nine
    // first-layer blocks of the 25m-shaped graph), global loads into one of TWO register sets (issued while chunk c - 2 is
    // multiplied), LDS store after chunk c - 1's products, products.  (Two chunks of loads in flight and four workgroups per
    // CU instead of one and three: measured, no change -- 0.122 ms for nine blocks over 72 k listed rows either way: the
    // kernel is not waiting on latency; its operand pieces are 256 bytes of 2304-byte rows.)
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    float4 pa[2][2], pb[2][2];
    unsigned char fl[2] = {0, 0};
    int64_t rid[2] = {-1, -1};      // -1: past the end of the part
    auto flags = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t n = base + lr + 16 * u;
            const int64_t nr = n < r1 ? M.row(n) : -1;
            const bool ok = nr >= 0 && nr < M.N;
            rid[u] = ok ? nr : -1;
            fl[u] = (J.b_mask && ok) ? J.b_mask[nr] : (unsigned char)0;
        }
    };
    auto fetch = [&](int set) {     // the chunk whose ids / flags are current
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool ok = rid[u] >= 0;
            const int64_t nc = ok ? rid[u] : 0;
            const float4 va = *reinterpret_cast<const float4 *>(J.a + nc * J.lda + lc);
            float4 vb = *reinterpret_cast<const float4 *>((fl[u] ? J.b_alt + nc * J.ldb_alt : J.b + nc * J.ldb) + lc);
            if (fl[u] && J.b_alt_scale) {
                const float bs = J.b_alt_scale[nc];
                vb = make_float4(bs * vb.x, bs * vb.y, bs * vb.z, bs * vb.w);
            }
            pa[set][u] = ok ? va : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[set][u] = ok ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int set, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<float4 *>(&As[buf][lr + 16 * u][gw_swz(lr + 16 * u, lc)]) = pa[set][u];
            *reinterpret_cast<float4 *>(&Bs[buf][lr + 16 * u][gw_swz(lr + 16 * u, lc)]) = pb[set][u];
        }
    };
    if (r0 < r1) {
        flags(r0);
        fetch(0);                 // chunk 0
        flags(r0 + 32);
        stash(0, 0);
        fetch(1);                 // chunk 1 (zeros past the end)
        flags(r0 + 64);           // ids of chunk 2
    }
    // (two chunks per trip so that the register set / LDS image indices are compile-time constants: indexed by a
    // run-time parity the register sets went to scratch memory)
    auto step = [&](auto CUR, int64_t base) {
        constexpr int cur = decltype(CUR)::value;
        __syncthreads();          // LDS image `cur` is complete; image cur ^ 1 is free (its products ended an iteration ago)
        const bool more1 = base + 32 < r1, more2 = base + 64 < r1;
        if (more2) {
            fetch(cur);           // chunk it + 2 -> register set (it & 1): chunk it's registers went to LDS an iteration ago
            flags(base + 96);     // ids of chunk it + 3
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float av = As[cur][4 * s + kq][gw_swz(kq, 16 * wave + i)];     // (4 s + kq) & 3 == kq
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bs[cur][4 * s + kq][gw_swz(kq, 16 * nt + i)], acc[nt], 0, 0, 0);
        }
        if (more1) stash(cur ^ 1, cur ^ 1);   // chunk it + 1, loaded while chunks it - 1 and it were multiplied
    };
    for (int64_t base = r0; base < r1; base += 64) {
        step(std::integral_constant<int, 0>(), base);
        if (base + 32 < r1) step(std::integral_constant<int, 1>(), base + 32);
    }
    float *dst = partial + ((size_t)job * parts + part) * (16 * 256);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int v = 0; v < 4; ++v) dst[((wave * 4 + nt) * 4 + v) * 64 + lane] = acc[nt][v];
}

// stage 1 for full 128 x 128 blocks (width-128 models: Yelp presets; every [128, N] x [N, 128] job is four 64 x 64 blocks,
// i.e. both operands read twice): the same pipeline on 32-row chunks of 128 + 128 columns; wave w owns tile rows 2w, 2w + 1
// (sixteen 16 x 16 tiles).  70 KB of LDS per workgroup (dynamic), two workgroups per CU.  Measured SLOWER than four 64 x 64
// blocks (see grad_weight_impl): kept for the record, off by default.
constexpr int kGwLd128 = 136;
extern __shared__ float gw_lds128[];
__global__ __launch_bounds__(256) void gw_stage1_lds128(const GwBatch Jb, const RowMap M, float *__restrict__ partial) {
    const int64_t n_rows = M.size();
    float (*As)[32][kGwLd128] = reinterpret_cast<float (*)[32][kGwLd128]>(gw_lds128);
    float (*Bs)[32][kGwLd128] = reinterpret_cast<float (*)[32][kGwLd128]>(gw_lds128 + 2 * 32 * kGwLd128);
    const int job = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
    const GwJob &J = Jb.j[job];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int64_t chunk = ((n_rows + parts - 1) / parts + 15) / 16 * 16;
    const int64_t r0 = (int64_t)part * chunk, r1 = min(n_rows, r0 + chunk);
    f32x4 acc[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // loader role: thread t moves float4 column (t % 32) of rows (t / 32) + 8 u, u = 0 .. 3, of a chunk, both operands
    const int lr = tid >> 5, lc = (tid & 31) * 4;
    float4 pa[2][4], pb[2][4];
    unsigned char fl[4] = {0, 0, 0, 0};
    int64_t rid[4] = {-1, -1, -1, -1};
    auto flags = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t n = base + lr + 8 * u;
            const int64_t nr = n < r1 ? M.row(n) : -1;
            const bool ok = nr >= 0 && nr < M.N;
            rid[u] = ok ? nr : -1;
            fl[u] = (J.b_mask && ok) ? J.b_mask[nr] : (unsigned char)0;
        }
    };
    auto fetch = [&](auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = rid[u] >= 0;
            const int64_t nc = ok ? rid[u] : 0;
            const float4 va = *reinterpret_cast<const float4 *>(J.a + nc * J.lda + lc);
            float4 vb = *reinterpret_cast<const float4 *>((fl[u] ? J.b_alt + nc * J.ldb_alt : J.b + nc * J.ldb) + lc);
            if (fl[u] && J.b_alt_scale) {
                const float bs = J.b_alt_scale[nc];
                vb = make_float4(bs * vb.x, bs * vb.y, bs * vb.z, bs * vb.w);
            }
            pa[set][u] = ok ? va : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[set][u] = ok ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](auto SET) {           // register set s -> LDS image s
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<float4 *>(&As[set][lr + 8 * u][lc]) = pa[set][u];
            *reinterpret_cast<float4 *>(&Bs[set][lr + 8 * u][lc]) = pb[set][u];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (r0 < r1) {
        flags(r0);
        fetch(I0());
        flags(r0 + 32);
        stash(I0());
        fetch(I1());
        flags(r0 + 64);
    }
    auto step = [&](auto CUR, auto NXT, int64_t base) {
        constexpr int cur = decltype(CUR)::value;
        __syncthreads();
        const bool more1 = base + 32 < r1, more2 = base + 64 < r1;
        if (more2) {
            fetch(CUR);
            flags(base + 96);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float a0 = As[cur][4 * s + kq][32 * wave + i], a1 = As[cur][4 * s + kq][32 * wave + 16 + i];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const float bv = Bs[cur][4 * s + kq][16 * nt + i];
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][nt], 0, 0, 0);
            }
        }
        if (more1) stash(NXT);
    };
    for (int64_t base = r0; base < r1; base += 64) {
        step(I0(), I1(), base);
        if (base + 32 < r1) step(I1(), I0(), base + 32);
    }
    float *dst = partial + ((size_t)job * parts + part) * (64 * 256);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int v = 0; v < 4; ++v) dst[(((2 * wave + m) * 8 + nt) * 4 + v) * 64 + lane] = acc[m][nt][v];
}

// stage 2: out = sum over parts, part order.  Element e of a record: tile (mt, nt), register v, lane l ->
// row 16 mt + 4 (l / 16) + v, column 16 nt + l % 16.
template <int MT, int NT>
__global__ __launch_bounds__(256) void gw_stage2(const GwBatch Jb, const float *__restrict__ partial, int parts) {
    // 64 elements per workgroup; quarter q of the threads adds parts q, q + 4, ... of its element in order, then the four
    // sub-sums are added in order (fixed order; one chain of `parts` dependent loads per element took as long as stage 1
    // on the listed-row launches)
    __shared__ float sub[3][64];
    const GwJob &J = Jb.j[blockIdx.y];
    const int q = threadIdx.x >> 6, e = blockIdx.x * 64 + (threadIdx.x & 63);
    const float *src = partial + (size_t)blockIdx.y * parts * (MT * NT * 256) + e;
    float s = 0.f;
#pragma unroll 8                 // the loads of eight parts in flight; the adds stay in part order
    for (int p = q; p < parts; p += 4) s += src[(size_t)p * (MT * NT * 256)];
    if (q > 0) sub[q - 1][threadIdx.x & 63] = s;
    __syncthreads();
    if (q > 0) return;
    s = ((s + sub[0][threadIdx.x]) + sub[1][threadIdx.x]) + sub[2][threadIdx.x];
    const int lane = e & 63, v = (e >> 6) & 3, t = e >> 8, mt = t / NT, nt = t % NT;
    const int row = 16 * mt + 4 * (lane >> 4) + v, col = 16 * nt + (lane & 15);
    if (row < J.ma && col < J.nb) J.out[(int64_t)row * J.ldo + col] = s;
}

// Row parts per job of one launch: (jobs x parts) = one round of resident workgroups for the LDS-staged kernel (four per CU),
// two for the dword kernel, each part keeping at least ~256 rows where the row count allows.  Measured on the 25m-shaped
// shapes (profiles/tools/gw_bench.py, profiles/r03/gw_bench_r03.txt): between 43 and 227 parts stage 1 moves within
// +-15 % (2.7-3.9 TB/s of operand bytes: 256-byte pieces of 2304-byte rows), while stage 2 grows with the part count.
// PEA_GW_PARTS overrides (experiments).
int gw_parts(int n_jobs, int64_t n_rows, bool lds_kernel) {
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const char *env = getenv("PEA_GW_PARTS");
    int parts;
    if (env && atoi(env) > 0) {
        parts = atoi(env);
    } else {
        const int round = std::max(1, (lds_kernel ? 4 : 3) * n_cu / n_jobs);    // parts of one resident round
        const int64_t fit = std::max<int64_t>(1, n_rows / 256);                  // parts that still hold ~256 rows
        const int rounds = lds_kernel ? 1 : 2;
        parts = (int)std::min<int64_t>((int64_t)round * rounds, std::max<int64_t>(fit, 8));
    }
    return std::max(1, std::min(std::min(parts, kGwMaxParts), kGwRecords / n_jobs));
}

template <int MT, int NT>
int launch_gw(const GwBatch &Jb, const RowMap &rows, float *partial, double bytes, int64_t n_rows, hipStream_t stream) {
    bool full = MT == 4 && NT == 4;   // every block 64 x 64 with float4-addressable operands: the LDS-staged kernel
    for (int q = 0; q < Jb.n && full; ++q)
        full = Jb.j[q].ma == 64 && Jb.j[q].nb == 64 && Jb.j[q].lda % 4 == 0 && Jb.j[q].ldb % 4 == 0 &&
               (reinterpret_cast<uintptr_t>(Jb.j[q].a) | reinterpret_cast<uintptr_t>(Jb.j[q].b)) % 16 == 0 &&
               (!Jb.j[q].b_mask || (Jb.j[q].ldb_alt % 4 == 0 && reinterpret_cast<uintptr_t>(Jb.j[q].b_alt) % 16 == 0));
    const int parts = gw_parts(Jb.n, n_rows, full);
    {
        ProfScope ps("grad_weight", stream, bytes);
        if (full) {
            PEA_LAUNCH(gw_stage1_lds, dim3((unsigned)Jb.n, (unsigned)parts), dim3(256), 0, stream, Jb, rows, partial);
        } else {
            PEA_LAUNCH((gw_stage1<MT, NT>), dim3((unsigned)Jb.n, (unsigned)parts), dim3(256), 0, stream, Jb, rows, partial);
        }
        PEA_HIP(hipGetLastError());
    }
    ProfScope ps("grad_weight_sum", stream);
    PEA_LAUNCH((gw_stage2<MT, NT>), dim3(MT * NT * 4, (unsigned)Jb.n), dim3(256), 0, stream, Jb, partial, parts);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

int tiles_of(int w) { return w <= 16 ? 1 : w <= 32 ? 2 : 4; }

int launch_gw128(const GwBatch &Jb, const RowMap &rows, float *partial, double bytes, int64_t n_rows, hipStream_t stream) {
    constexpr size_t lds = (size_t)4 * 32 * kGwLd128 * sizeof(float);
    static bool lds_set = false;
    if (!lds_set) {
        PEA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gw_stage1_lds128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = true;
    }
    // a record is 64 tiles (four of the 64 x 64 kind); two workgroups per CU
    int parts = gw_parts(2 * Jb.n, n_rows, true);
    parts = std::max(1, std::min(parts, kGwRecords / (4 * Jb.n)));
    {
        ProfScope ps("grad_weight", stream, bytes);
        PEA_LAUNCH(gw_stage1_lds128, dim3((unsigned)Jb.n, (unsigned)parts), dim3(256), lds, stream, Jb, rows, partial);
        PEA_HIP(hipGetLastError());
    }
    ProfScope ps("grad_weight_sum", stream);
    PEA_LAUNCH((gw_stage2<8, 8>), dim3(64 * 4, (unsigned)Jb.n), dim3(256), 0, stream, Jb, partial, parts);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

}  // namespace
}  // namespace pea

using namespace pea;

extern "C" size_t pea_grad_weight_workspace_bytes(void) {
    return (size_t)kGwRecords * 16 * 256 * sizeof(float) + 256;
}

extern "C" int pea_grad_weight_sharded(int64_t n_rows, int shard_tile, int shard_world, int shard_rank, int n_jobs,
                                       const pea_gw_job *jobs_host, void *workspace, size_t workspace_bytes, void *stream);

static int grad_weight_impl(const pea::RowMap &rowmap, int64_t n_rows, int n_jobs, const pea_gw_job *jobs_host, void *workspace,
                            size_t workspace_bytes, void *stream);

extern "C" int pea_grad_weight_rows(int64_t num_rows, const int32_t *rows, const int32_t *count_dev, int64_t capacity, int n_jobs,
                                    const pea_gw_job *jobs_host, void *workspace, size_t workspace_bytes, void *stream) {
    PEA_REQUIRE(num_rows > 0 && rows && count_dev && capacity >= 0 && capacity <= num_rows, PEA_ERR_ARG, "grad_weight_rows: bad row list");
    return grad_weight_impl(pea::make_rowmap_list(num_rows, rows, count_dev, capacity), capacity, n_jobs, jobs_host, workspace,
                            workspace_bytes, stream);
}

extern "C" int pea_grad_weight(int64_t n_rows, int n_jobs, const pea_gw_job *jobs_host, void *workspace,
                               size_t workspace_bytes, void *stream) {
    return pea_grad_weight_sharded(n_rows, 1, 1, 0, n_jobs, jobs_host, workspace, workspace_bytes, stream);
}

// the same reduction over the rows ONE RANK owns (row i belongs to rank (i / tile) % world): its share of dW; the
// ranks' shares are summed by the host mirror (all-reduce)
extern "C" int pea_grad_weight_sharded(int64_t n_rows, int shard_tile, int shard_world, int shard_rank, int n_jobs,
                                       const pea_gw_job *jobs_host, void *workspace, size_t workspace_bytes, void *stream) {
    PEA_REQUIRE(n_rows >= 0 && n_jobs >= 0 && (jobs_host || n_jobs == 0), PEA_ERR_ARG, "grad_weight: bad arguments");
    PEA_REQUIRE(shard_world >= 1 && shard_rank >= 0 && shard_rank < shard_world && shard_tile > 0, PEA_ERR_ARG,
                "grad_weight: bad shard (rank %d of %d, tile %d)", shard_rank, shard_world, shard_tile);
    return grad_weight_impl(make_rowmap(n_rows, shard_tile, shard_world, shard_rank), n_rows, n_jobs, jobs_host, workspace,
                            workspace_bytes, stream);
}

static int grad_weight_impl(const pea::RowMap &rowmap, int64_t n_rows, int n_jobs, const pea_gw_job *jobs_host, void *workspace,
                            size_t workspace_bytes, void *stream) {
    PEA_REQUIRE(n_jobs >= 0 && (jobs_host || n_jobs == 0), PEA_ERR_ARG, "grad_weight: bad arguments");
    PEA_REQUIRE(workspace && workspace_bytes >= pea_grad_weight_workspace_bytes(), PEA_ERR_NOMEM, "grad_weight: workspace too small");
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
    // cut every job into <= 64 x 64 blocks, group the blocks by tile shape, one pair of launches per shape and batch
    std::vector<GwJob> blocks[3][3], blocks128;
    // OFF unless PEA_GW128=1: measured slower on the Yelp-shaped presets (0.84 vs 0.73 ms per step for SAGE, 0.50 vs 0.44 GAT,
    // 0.48 vs 0.42 GCN: profiles/r03/gw128_r03.txt) although both operands are read once instead of twice -- the reduction is
    // bound by its LDS -> MFMA dependency chain per workgroup, and the 70 KB image leaves 8 waves per CU where the 64 x 64
    // kernel keeps 16
    const char *env128 = getenv("PEA_GW128");
    const bool use128 = env128 && atoi(env128) == 1;
    for (int q = 0; q < n_jobs; ++q) {
        const pea_gw_job &S = jobs_host[q];
        PEA_REQUIRE(S.a && S.b && S.out && S.ma > 0 && S.nb > 0 && S.lda >= S.ma && S.ldb >= S.nb && S.ldo >= S.nb,
                    PEA_ERR_ARG, "grad_weight: job %d malformed", q);
        PEA_REQUIRE(!S.b_mask || (S.b_alt && S.ldb_alt >= S.nb), PEA_ERR_ARG, "grad_weight: job %d has a row mask but no alternative operand", q);
        // whole 128 x 128 blocks with float4-addressable operands: one block each (both operands read once)
        if (use128 && S.ma % 128 == 0 && S.nb % 128 == 0 && S.lda % 4 == 0 && S.ldb % 4 == 0 &&
            (reinterpret_cast<uintptr_t>(S.a) | reinterpret_cast<uintptr_t>(S.b)) % 16 == 0 &&
            (!S.b_mask || (S.ldb_alt % 4 == 0 && reinterpret_cast<uintptr_t>(S.b_alt) % 16 == 0))) {
            for (int i0 = 0; i0 < S.ma; i0 += 128)
                for (int j0 = 0; j0 < S.nb; j0 += 128) {
                    GwJob B;
                    B.a = S.a + i0;
                    B.b = S.b + j0;
                    B.b_mask = S.b_mask;
                    B.b_alt = S.b_mask ? S.b_alt + j0 : nullptr;
                    B.b_alt_scale = S.b_mask ? S.b_alt_scale : nullptr;
                    B.ldb_alt = S.ldb_alt;
                    B.lda = S.lda;
                    B.ldb = S.ldb;
                    B.ma = 128;
                    B.nb = 128;
                    B.out = S.out + (int64_t)i0 * S.ldo + j0;
                    B.ldo = S.ldo;
                    blocks128.push_back(B);
                }
            continue;
        }
        for (int i0 = 0; i0 < S.ma; i0 += 64)
            for (int j0 = 0; j0 < S.nb; j0 += 64) {
                GwJob B;
                B.a = S.a + i0;
                B.b = S.b + j0;
                B.b_mask = S.b_mask;
                B.b_alt = S.b_mask ? S.b_alt + j0 : nullptr;
                B.b_alt_scale = S.b_mask ? S.b_alt_scale : nullptr;
                B.ldb_alt = S.ldb_alt;
                B.lda = S.lda;
                B.ldb = S.ldb;
                B.ma = std::min(64, S.ma - i0);
                B.nb = std::min(64, S.nb - j0);
                B.out = S.out + (int64_t)i0 * S.ldo + j0;
                B.ldo = S.ldo;
                const int mt = tiles_of(B.ma), nt = tiles_of(B.nb);
                blocks[mt == 1 ? 0 : mt == 2 ? 1 : 2][nt == 1 ? 0 : nt == 2 ? 1 : 2].push_back(B);
            }
    }
    for (size_t base = 0; base < blocks128.size(); base += kGwMaxJobs) {
        GwBatch Jb;
        Jb.n = (int)std::min<size_t>(kGwMaxJobs, blocks128.size() - base);
        for (int q = 0; q < Jb.n; ++q) Jb.j[q] = blocks128[base + q];
        PEA_TRY(launch_gw128(Jb, rowmap, partial, 4.0 * (double)n_rows * 256.0 * Jb.n, n_rows, (hipStream_t)stream));
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const std::vector<GwJob> &L = blocks[a][b];
            for (size_t base = 0; base < L.size(); base += kGwMaxJobs) {
                GwBatch Jb;
                Jb.n = (int)std::min<size_t>(kGwMaxJobs, L.size() - base);
                double bytes = 0.0;
                for (int q = 0; q < Jb.n; ++q) {
                    Jb.j[q] = L[base + q];
                    bytes += 4.0 * (double)n_rows * (Jb.j[q].ma + Jb.j[q].nb);
                }
                hipStream_t st = (hipStream_t)stream;
                int rc = PEA_OK;
                switch (a * 3 + b) {
                    case 0: rc = launch_gw<1, 1>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    case 1: rc = launch_gw<1, 2>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    case 2: rc = launch_gw<1, 4>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    case 3: rc = launch_gw<2, 1>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    case 4: rc = launch_gw<2, 2>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    case 5: rc = launch_gw<2, 4>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    case 6: rc = launch_gw<4, 1>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    case 7: rc = launch_gw<4, 2>(Jb, rowmap, partial, bytes, n_rows, st); break;
                    default: rc = launch_gw<4, 4>(Jb, rowmap, partial, bytes, n_rows, st); break;
                }
                PEA_TRY(rc);
            }
        }
    return PEA_OK;
}

extern "C" int pea_dense_batch_rows(int64_t n_rows, const int32_t *rows, int n_jobs, const pea_dense_job *jobs_host,
                                    void *stream);

extern "C" int pea_dense_batch(int64_t n_rows, int n_jobs, const pea_dense_job *jobs_host, void *stream) {
    return pea_dense_batch_rows(n_rows, nullptr, n_jobs, jobs_host, stream);
}

// rows != null: only the listed rows (device int32 [n_rows], e.g. the rows a rank owns) of every operand are read / written
extern "C" int pea_dense_batch_rows(int64_t n_rows, const int32_t *rows, int n_jobs, const pea_dense_job *jobs_host,
                                    void *stream) {
    PEA_REQUIRE(n_rows >= 0 && n_jobs >= 0 && (jobs_host || n_jobs == 0), PEA_ERR_ARG, "dense_batch: bad arguments");
    std::vector<GemmJob> jobs((size_t)n_jobs);
    for (int q = 0; q < n_jobs; ++q) {
        const pea_dense_job &S = jobs_host[q];
        PEA_REQUIRE(S.a && S.w && S.out && S.k > 0 && S.n_out > 0 && S.k % 4 == 0 && S.n_out % 4 == 0 && S.lda % 4 == 0 &&
                        S.ldw >= S.n_out && S.ldo >= S.n_out && S.lda >= S.k,
                    PEA_ERR_ARG, "dense_batch: job %d malformed (widths and the input stride must be multiples of 4)", q);
        GemmJob J{};
        J.A1 = S.a;
        J.lda1 = (int)S.lda;
        J.K1 = S.k;
        J.B = S.w;
        J.ldb = (int)S.ldw;
        J.n_out = S.n_out;
        J.n_seg = 1;
        J.seg[0].c0 = 0;
        J.seg[0].c1 = S.n_out;
        J.seg[0].dst = S.out;
        J.seg[0].ld = (int)S.ldo;
        J.seg[0].relu = 0;
        PEA_REQUIRE(S.gate == nullptr || (S.ld_gate >= S.n_out && S.k <= 128), PEA_ERR_ARG,
                    "dense_batch: job %d: a gate needs a row stride covering the columns and k <= 128", q);
        J.seg[0].gate = S.gate;
        J.seg[0].ld_gate = (int)S.ld_gate;
        jobs[(size_t)q] = J;
    }
    return launch_gemm_batch(jobs.data(), n_jobs, rows, n_rows, (hipStream_t)stream);
}


// out[n, 0:width] = sum_{b < n_blocks} src[n, b * width : (b + 1) * width], blocks added in order b = 0, 1, ... (fixed order:
// bitwise reproducible).  Two-step training schedule: the first layer's x-space backward leaves one dx part per channel side
// by side ([N, P * emb]); their sum is dx (reference: autograd sums the gradients of the P uses of self.x, models/base.py:193).
namespace pea {
namespace {
__global__ __launch_bounds__(256) void block_sum_kernel(int64_t n_rows, int n_blocks, int w4, const float *__restrict__ src,
                                                        int64_t ld, float *__restrict__ dst, int64_t ld_dst) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_rows * w4) return;
    const int64_t row = idx / w4;
    const int c = (int)(idx - row * w4) * 4;
    const float *p = src + row * ld + c;
    float4 acc = *reinterpret_cast<const float4 *>(p);
    for (int b = 1; b < n_blocks; ++b) {
        const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)b * w4 * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(dst + row * ld_dst + c) = acc;
}
}  // namespace
}  // namespace pea

extern "C" int pea_block_sum(int64_t n_rows, int n_blocks, int width, const float *src, int64_t ld, float *dst, int64_t ld_dst,
                             void *stream) {
    PEA_REQUIRE(n_rows >= 0 && n_blocks > 0 && width > 0 && width % 4 == 0 && src && dst && ld % 4 == 0 && ld_dst % 4 == 0 &&
                    ld >= (int64_t)n_blocks * width && ld_dst >= width, PEA_ERR_ARG, "block_sum: bad argument");
    if (n_rows == 0) return PEA_OK;
    pea::ProfScope ps("block_sum", (hipStream_t)stream, 4.0 * (double)n_rows * width * (n_blocks + 1));
    const int64_t total = n_rows * (width / 4);
    PEA_LAUNCH(pea::block_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_rows, n_blocks,
               width / 4, src, ld, dst, ld_dst);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}
