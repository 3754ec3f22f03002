// Row packing / unpacking for the multi-GPU exchanges, on the launch stream: the host mirror
// (graph_recsys_benchmark_amd/sharding.py) used to do this with torch index arithmetic -- a dozen small launches, a
// zero-filled staging tensor and a clone per exchange, first-order on a step whose per-rank kernel time is a few
// hundred microseconds.  These three kernels move whole float4s between a node-major table and a rank-major
// exchange buffer; the collective itself (RCCL all-gather / all-reduce) stays with torch.distributed.
//   pack    dst[k, 0:w]                     = table[nodes[k], col:col+w]
//   unpack  table[nodes[k], col:col+w]      = src[src_rows ? src_rows[k] : k, 0:w]
//   select  out[k, 0:w]                     = owner(ids[k]) == rank ? table[ids[k], 0:w] : 0     (batch rows for the loss)
// The reference has no counterpart: its forward is single-process (models/base.py:191-206).
#include "common.h"

namespace pea {
namespace {

__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4g(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

__global__ __launch_bounds__(256) void pack_rows_kernel(int64_t n, int w4, const float *__restrict__ table, int64_t ld,
                                                        const int *__restrict__ nodes, float *__restrict__ dst, int64_t dst_ld) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * w4) return;
    const int64_t k = t / w4;
    const int c = (int)(t % w4) * 4;
    st4g(dst + k * dst_ld + c, ld4g(table + (int64_t)nodes[k] * ld + c));
}

__global__ __launch_bounds__(256) void unpack_rows_kernel(int64_t n, int w4, const float *__restrict__ src, int64_t src_ld,
                                                          const int *__restrict__ src_rows, const int *__restrict__ nodes,
                                                          float *__restrict__ table, int64_t ld) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * w4) return;
    const int64_t k = t / w4;
    const int c = (int)(t % w4) * 4;
    const int64_t s = src_rows ? (int64_t)src_rows[k] : k;
    st4g(table + (int64_t)nodes[k] * ld + c, ld4g(src + s * src_ld + c));
}

__global__ __launch_bounds__(256) void select_owned_kernel(int64_t n, int w4, int64_t N, const float *__restrict__ table, int64_t ld,
                                                           const int64_t *__restrict__ ids, int64_t id_stride, int rank, int world,
                                                           int tile, float *__restrict__ out, int *err) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * w4) return;
    const int64_t k = t / w4;
    const int c = (int)(t % w4) * 4;
    const int64_t v = ids[k * id_stride];
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < 0 || v >= N) {
        if (c == 0) atomicOr(err, 1);
    } else if ((int)((v / tile) % world) == rank) {
        r = ld4g(table + v * ld + c);
    }
    st4g(out + k * (int64_t)(w4 * 4) + c, r);
}

// Batched forms: several (table, column block, node list) jobs against ONE rank-major exchange buffer whose rank blocks
// are `rank_stride` floats apart; job q's rows sit at buf_off[q] inside a rank block, `width` floats per row.  One launch
// moves all jobs (a sharded backward level used one pack, one collective and one unpack PER relation and buffer).
constexpr int kMaxXchgJobs = 24;
struct XchgJob {
    float *table;
    int64_t ld;
    const int *nodes;     // table rows
    const int *slots;     // unpack: slot of each row in the exchange order (rank * slots_per_rank + local); pack: null (k)
    int64_t n;
    int64_t buf_off;      // floats from the start of a rank block
    int w4, slots_per_rank;
    int64_t first;        // first float4 work item of this job
};
struct XchgBatch {
    int n_jobs;
    int64_t total;        // float4 work items
    XchgJob j[kMaxXchgJobs];
};

template <bool PACK>
__global__ __launch_bounds__(256) void xchg_batch_kernel(const XchgBatch B, float *__restrict__ buf, int64_t rank_stride) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= B.total) return;
    int q = 0;
    while (q + 1 < B.n_jobs && t >= B.j[q + 1].first) ++q;
    const XchgJob &J = B.j[q];
    const int64_t e = t - J.first, k = e / J.w4;
    const int c = (int)(e % J.w4) * 4;
    float *row = J.table + (int64_t)J.nodes[k] * J.ld + c;
    if (PACK) {   // buf = this rank's block
        st4g(buf + J.buf_off + k * (int64_t)(J.w4 * 4) + c, ld4g(row));
    } else {      // buf = rank 0's block
        const int s = J.slots[k];
        st4g(row, ld4g(buf + (int64_t)(s / J.slots_per_rank) * rank_stride + J.buf_off + (int64_t)(s % J.slots_per_rank) * (J.w4 * 4) + c));
    }
}

int check_w(int width, int64_t ld, const char *what) {
    PEA_REQUIRE(width > 0 && width % 4 == 0 && ld % 4 == 0 && ld >= width, PEA_ERR_ARG,
                "%s: width %d / row stride %lld must be multiples of 4 floats, stride >= width", what, width, (long long)ld);
    return PEA_OK;
}

}  // namespace
}  // namespace pea

using namespace pea;

extern "C" int pea_rows_pack(const float *table, int64_t ld, int col, int width, const int32_t *nodes, int64_t n, float *dst,
                             int64_t dst_ld, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(n >= 0 && col >= 0 && col % 4 == 0, PEA_ERR_ARG, "rows_pack: n=%lld col=%d", (long long)n, col);
    if (n == 0) return PEA_OK;
    PEA_REQUIRE(table && nodes && dst, PEA_ERR_ARG, "rows_pack: null pointer");
    PEA_TRY(check_w(width, ld, "rows_pack"));
    PEA_TRY(check_w(width, dst_ld, "rows_pack"));
    ProfScope ps("xchg_pack", stream, 8.0 * (double)n * width);
    const int w4 = width / 4;
    PEA_LAUNCH(pack_rows_kernel, dim3((unsigned)((n * w4 + 255) / 256)), dim3(256), 0, stream, n, w4, table + col, ld,
                       nodes, dst, dst_ld);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

extern "C" int pea_rows_unpack(const float *src, int64_t src_ld, const int32_t *src_rows, int width, const int32_t *nodes,
                               int64_t n, float *table, int64_t ld, int col, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(n >= 0 && col >= 0 && col % 4 == 0, PEA_ERR_ARG, "rows_unpack: n=%lld col=%d", (long long)n, col);
    if (n == 0) return PEA_OK;
    PEA_REQUIRE(table && nodes && src, PEA_ERR_ARG, "rows_unpack: null pointer");
    PEA_TRY(check_w(width, ld, "rows_unpack"));
    PEA_TRY(check_w(width, src_ld, "rows_unpack"));
    ProfScope ps("xchg_unpack", stream, 8.0 * (double)n * width);
    const int w4 = width / 4;
    PEA_LAUNCH(unpack_rows_kernel, dim3((unsigned)((n * w4 + 255) / 256)), dim3(256), 0, stream, n, w4, src, src_ld,
                       src_rows, nodes, table + col, ld);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

extern "C" int pea_rows_select_owned(const float *table, int64_t ld, int width, int64_t num_nodes, const int64_t *ids,
                                     int64_t id_stride, int64_t n, int rank, int world, int tile, float *out, int32_t *err_flag,
                                     void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(n >= 0 && world >= 1 && rank >= 0 && rank < world && tile > 0 && id_stride >= 1, PEA_ERR_ARG,
                "rows_select_owned: bad argument");
    if (n == 0) return PEA_OK;
    PEA_REQUIRE(table && ids && out && err_flag, PEA_ERR_ARG, "rows_select_owned: null pointer");
    PEA_TRY(check_w(width, ld, "rows_select_owned"));
    ProfScope ps("xchg_select", stream, 8.0 * (double)n * width);
    const int w4 = width / 4;
    PEA_LAUNCH(select_owned_kernel, dim3((unsigned)((n * w4 + 255) / 256)), dim3(256), 0, stream, n, w4, num_nodes, table,
                       ld, ids, id_stride, rank, world, tile, out, err_flag);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

static int xchg_batch(bool pack, int n_jobs, const pea_xchg_job *jobs, float *buf, int64_t rank_stride, hipStream_t stream) {
    PEA_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0) && buf, PEA_ERR_ARG, "rows_%s_batch: bad argument", pack ? "pack" : "unpack");
    for (int base = 0; base < n_jobs; base += kMaxXchgJobs) {
        XchgBatch B;
        B.n_jobs = 0;
        B.total = 0;
        double bytes = 0.0;
        for (int q = base; q < n_jobs && q < base + kMaxXchgJobs; ++q) {
            const pea_xchg_job &S = jobs[q];
            if (S.n <= 0) continue;
            PEA_REQUIRE(S.table && S.nodes && (pack || S.slots) && S.col >= 0 && S.col % 4 == 0 && S.buf_off >= 0 && S.buf_off % 4 == 0 &&
                            (pack || S.slots_per_rank > 0),
                        PEA_ERR_ARG, "rows_%s_batch: job %d malformed", pack ? "pack" : "unpack", q);
            PEA_TRY(check_w(S.width, S.ld, "rows_batch"));
            XchgJob &J = B.j[B.n_jobs++];
            J.table = S.table + S.col;
            J.ld = S.ld;
            J.nodes = S.nodes;
            J.slots = S.slots;
            J.n = S.n;
            J.buf_off = S.buf_off;
            J.w4 = S.width / 4;
            J.slots_per_rank = S.slots_per_rank;
            J.first = B.total;
            B.total += S.n * J.w4;
            bytes += 8.0 * (double)S.n * S.width;
        }
        if (B.total == 0) continue;
        ProfScope ps(pack ? "xchg_pack" : "xchg_unpack", stream, bytes);
        const unsigned blocks = (unsigned)((B.total + 255) / 256);
        if (pack) PEA_LAUNCH(xchg_batch_kernel<true>, dim3(blocks), dim3(256), 0, stream, B, buf, rank_stride);
        else PEA_LAUNCH(xchg_batch_kernel<false>, dim3(blocks), dim3(256), 0, stream, B, buf, rank_stride);
        PEA_HIP(hipGetLastError());
    }
    return PEA_OK;
}

extern "C" int pea_rows_pack_batch(int n_jobs, const pea_xchg_job *jobs_host, float *rank_block, void *stream) {
    return xchg_batch(true, n_jobs, jobs_host, rank_block, 0, (hipStream_t)stream);
}

extern "C" int pea_rows_unpack_batch(int n_jobs, const pea_xchg_job *jobs_host, const float *buffer, int64_t rank_stride,
                                     void *stream) {
    PEA_REQUIRE(rank_stride > 0 && rank_stride % 4 == 0, PEA_ERR_ARG, "rows_unpack_batch: rank stride %lld", (long long)rank_stride);
    return xchg_batch(false, n_jobs, jobs_host, const_cast<float *>(buffer), rank_stride, (hipStream_t)stream);
}
