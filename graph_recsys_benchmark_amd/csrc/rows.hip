// Device-side row sets for the training backward.  Only the BPR batch's rows carry an output gradient (reference
// models/base.py:46-48 reads cached_repr at the batch's ids), so after the last layer's gradient gathers the input gradient of
// that layer, dT_1, is identically zero on every node that is neither a batch row nor an in-neighbour of one (on the 25m-shaped
// graph: ~3/4 of the nodes).  pea_rows_nonzero marks the rows of a table that hold a non-zero and compacts their ids, count and
// all, in device memory -- the host never reads the count, nothing synchronises; the dense half of the first layer's backward
// then walks that list instead of all N rows.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace pea {
namespace {

// 8 lanes per row, float4 each: a lane group ORs its chunks, lane 0 of the group writes the flag
__global__ __launch_bounds__(256) void row_nonzero_kernel(int64_t n_rows, int w4, const float *__restrict__ src, int64_t ld,
                                                          const unsigned char *__restrict__ or_flags,
                                                          unsigned char *__restrict__ flags) {
    const int64_t row = (int64_t)blockIdx.x * 32 + threadIdx.x / 8;
    const int sl = threadIdx.x % 8;
    bool nz = false;
    if (row < n_rows) {
        const float *p = src + row * ld;
        for (int c = sl; c < w4; c += 8) {
            const float4 v = *reinterpret_cast<const float4 *>(p + 4 * c);
            nz = nz || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
        }
    }
    int any = nz ? 1 : 0;
    any |= __shfl_xor(any, 1);
    any |= __shfl_xor(any, 2);
    any |= __shfl_xor(any, 4);
    if (row < n_rows && sl == 0) flags[row] = (unsigned char)((any || (or_flags && or_flags[row])) ? 1 : 0);
}

// table[list[q], 0:4*w4] = 0 for q < *count: 16 lanes x float4 per row piece, a fixed grid walking the list
__global__ __launch_bounds__(256) void rows_zero_kernel(const int *__restrict__ list, const int *__restrict__ count, int w4,
                                                        float *__restrict__ table, int64_t ld) {
    const int n = *count;
    const int sl = threadIdx.x % 16;
    for (int q = (int)blockIdx.x * 16 + threadIdx.x / 16; q < n; q += (int)gridDim.x * 16) {
        float *p = table + (int64_t)list[q] * ld;
        for (int c = sl; c < w4; c += 16) *reinterpret_cast<float4 *>(p + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

size_t select_temp_bytes(int64_t n_rows) {
    size_t bytes = 0;
    (void)rocprim::select(nullptr, bytes, rocprim::counting_iterator<int>(0), (const unsigned char *)nullptr, (int *)nullptr,
                          (int *)nullptr, (size_t)n_rows, (hipStream_t)0);
    return bytes;
}

}  // namespace
}  // namespace pea

extern "C" size_t pea_rows_nonzero_workspace_bytes(int64_t n_rows) {
    return n_rows > 0 ? pea::select_temp_bytes(n_rows) + 256 : 0;
}

extern "C" int pea_rows_nonzero(int64_t n_rows, int width, const float *src, int64_t ld, unsigned char *flags, int32_t *list,
                                int32_t *count_dev, void *workspace, size_t workspace_bytes, void *stream_) {
    return pea_rows_nonzero_or(n_rows, width, src, ld, nullptr, flags, list, count_dev, workspace, workspace_bytes, stream_);
}

extern "C" int pea_rows_nonzero_or(int64_t n_rows, int width, const float *src, int64_t ld, const unsigned char *or_flags,
                                   unsigned char *flags, int32_t *list, int32_t *count_dev, void *workspace, size_t workspace_bytes,
                                   void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(n_rows > 0 && n_rows < (int64_t)INT32_MAX && width > 0 && width % 4 == 0 && ld % 4 == 0 && ld >= width && src &&
                    flags && list && count_dev && workspace && or_flags != flags, PEA_ERR_ARG, "rows_nonzero: bad argument");
    size_t temp = pea::select_temp_bytes(n_rows);
    PEA_REQUIRE(workspace_bytes >= temp + 256, PEA_ERR_NOMEM, "rows_nonzero: workspace too small");
    pea::ProfScope ps("rows_nonzero", stream, 4.0 * (double)n_rows * width);
    PEA_LAUNCH(pea::row_nonzero_kernel, dim3((unsigned)((n_rows + 31) / 32)), dim3(256), 0, stream, n_rows, width / 4, src, ld, or_flags, flags);
    PEA_HIP(hipGetLastError());
    void *tmp = aligned_ws(workspace);
    PEA_HIP(rocprim::select(tmp, temp, rocprim::counting_iterator<int>(0), (const unsigned char *)flags, list, count_dev,
                            (size_t)n_rows, stream));
    return PEA_OK;
}

// table[list[q], 0:width] = 0 for the *count_dev listed rows (the rows a previous training step left non-zero in a table whose
// other rows are zero by invariant: see autograd.py, two-step training schedule)
extern "C" int pea_rows_zero(float *table, int64_t ld, int width, const int32_t *list, const int32_t *count_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(table && list && count_dev && width > 0 && width % 4 == 0 && ld % 4 == 0 && ld >= width, PEA_ERR_ARG, "rows_zero: bad argument");
    pea::ProfScope ps("rows_zero", stream);
    PEA_LAUNCH(pea::rows_zero_kernel, dim3(2048), dim3(256), 0, stream, list, count_dev, width / 4, table, ld);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}
