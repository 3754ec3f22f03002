// Which ingredient of the transform kernel's inner loop keeps the fp32 matrix pipe from its 154 TFLOP/s:
// 32-MFMA chains fed (a) from registers, (b) B operand from LDS (ds_read_b32 per MFMA, compiler-scheduled),
// (c) B operand read ahead one whole tile, (d) ds_read_b128-style wide reads.
// hipcc --offload-arch=gfx950 -O3 profiles/tools/mfma_lds.cpp -o profiles/tools/_bin/mfma_lds
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern __shared__ float lds[];
constexpr int KH = 32, LD = 576;

template <int MODE>
__global__ __launch_bounds__(1024) void kern(float *out, int tiles, float a0) {
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * KH * LD; i += 1024) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    float a[KH];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) a[kk] = a0 + kk + lane;
    const float *bimg = lds + (h * KH) * LD + r;
    float s = 0.f;
    if (MODE == 0) {  // registers only
        for (int t = 0; t < tiles; ++t) {
            f32x16 acc;
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], a[(kk + 1) % KH], acc, 0, 0, 0);
            s += acc[0] + acc[15];
        }
    } else if (MODE == 1) {  // B from LDS, compiler's schedule
        for (int t = 0; t < tiles; ++t) {
            const int col0 = (t % 18) * 32;
            f32x16 acc;
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            float b[KH];
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) b[kk] = bimg[kk * LD + col0];
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
            s += acc[0] + acc[15];
        }
    } else if (MODE == 2) {  // B of the NEXT tile read while this tile's MFMAs run
        float b[KH], bn[KH];
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) b[kk] = bimg[kk * LD];
        for (int t = 0; t < tiles; ++t) {
            const int coln = ((t + 1) % 18) * 32;
            f32x16 acc;
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                bn[kk] = bimg[kk * LD + coln];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) b[kk] = bn[kk];
            s += acc[0] + acc[15];
        }
    } else if (MODE == 3) {  // two tiles in flight (independent chains), B from LDS
        for (int t = 0; t < tiles; t += 2) {
            const int col0 = (t % 18) * 32;
            f32x16 acc, acc2;
            for (int i = 0; i < 16; ++i) acc[i] = 0.f, acc2[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bimg[kk * LD + col0], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bimg[kk * LD + col0 + 32], acc2, 0, 0, 0);
            }
            s += acc[0] + acc[15] + acc2[0] + acc2[15];
        }
    }
    out[blockIdx.x * 1024 + tid] = s;
}

// Bottom-up GEMM_0 ([N,64] x [64,576] -> [N,576]): STEP 0 = item loop + LDS B + MFMA, results discarded;
// 1 = + real stores; 2 = + real A loads; 3 = 2 with A of the next item prefetched.
template <int STEP, int PRIO = 0>
__global__ __launch_bounds__(1024) void gemm0(const float *__restrict__ x, const float *__restrict__ B, float *__restrict__ T,
                                              int n_rows, int ld, int nct, int col_group, float a0) {
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * KH * ld / 4; i += 1024) reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(B)[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (PRIO == 1) {   // the 4 waves of a SIMD get distinct priorities: they fall out of lockstep (one computes, one stores)
        switch ((wave >> 2) & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
    const int n_tiles = (n_rows + 31) / 32, n_grp = (nct + col_group - 1) / col_group, n_items = n_tiles * n_grp;
    const float *bimg = lds + (4 * h) * ld + r;
    float s = 0.f;
    float a[KH];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) a[kk] = a0 + kk + lane;
    for (int item = blockIdx.x * 16 + wave; item < n_items; item += gridDim.x * 16) {
        const int tile = item % n_tiles, grp = item / n_tiles;
        const int row0 = tile * 32;
        if (STEP >= 2) {
            const int row = min(row0 + r, n_rows - 1);
            const float4 *px = reinterpret_cast<const float4 *>(x + (size_t)row * 64 + 4 * h);
            float4 v[KH / 4];
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) v[q] = px[2 * q];
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) a[4 * q] = v[q].x, a[4 * q + 1] = v[q].y, a[4 * q + 2] = v[q].z, a[4 * q + 3] = v[q].w;
        }
        const int ct_end = min(nct, (grp + 1) * col_group);
        for (int ct = grp * col_group; ct < ct_end; ++ct) {
            const int col0 = ct * 32;
            f32x16 acc;
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            float b[KH];
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) b[kk] = bimg[(8 * (kk / 4) + kk % 4) * ld + col0];
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
            if (STEP >= 1) {
                float *p = T + (size_t)(row0 + 4 * h) * ld + col0 + r;
                if (row0 + 32 <= n_rows) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) p[(size_t)((reg & 3) + 8 * (reg >> 2)) * ld] = acc[reg];
                }
            } else {
                s += acc[0] + acc[15];
            }
        }
    }
    if (STEP == 0) T[blockIdx.x * 1024 + tid] = s;
}

// Same product, but the 12 waves of a workgroup write whole rows together: per step a workgroup takes two 32-row
// tiles, wave w -> row tile w / 6, column tiles 3 * (w % 6) .. + 3 (18 column tiles = 576 columns).  MF = 0: no MFMAs
// (store pattern alone).
template <int MF>
__global__ __launch_bounds__(768) void gemm0_rows(const float *__restrict__ x, const float *__restrict__ B, float *__restrict__ T,
                                                  int n_rows, int ld, float a0) {
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * KH * ld / 4; i += 768) reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(B)[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (n_rows + 31) / 32, n_steps = (n_tiles + 1) / 2;
    const float *bimg = lds + (4 * h) * ld + r;
    for (int step = blockIdx.x; step < n_steps; step += gridDim.x) {
        const int tile = 2 * step + wave / 6;
        if (tile >= n_tiles) continue;
        const int row0 = tile * 32;
        float a[KH];
        {
            const int row = min(row0 + r, n_rows - 1);
            const float4 *px = reinterpret_cast<const float4 *>(x + (size_t)row * 64 + 4 * h);
            float4 v[KH / 4];
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) v[q] = px[2 * q];
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) a[4 * q] = v[q].x, a[4 * q + 1] = v[q].y, a[4 * q + 2] = v[q].z, a[4 * q + 3] = v[q].w;
        }
        for (int ct = 3 * (wave % 6); ct < 3 * (wave % 6) + 3; ++ct) {
            const int col0 = ct * 32;
            f32x16 acc;
            for (int i = 0; i < 16; ++i) acc[i] = a[i];
            if (MF) {
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                float b[KH];
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) b[kk] = bimg[(8 * (kk / 4) + kk % 4) * ld + col0];
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
            }
            float *p = T + (size_t)(row0 + 4 * h) * ld + col0 + r;
            if (row0 + 32 <= n_rows) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) p[(size_t)((reg & 3) + 8 * (reg >> 2)) * ld] = acc[reg];
            }
        }
    }
}

template <int MF>
static void run_rows(const char *name, const float *x, const float *B, float *T, int n_rows) {
    const int ld = 576;
    const size_t ldsb = 2 * KH * ld * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm0_rows<MF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) gemm0_rows<MF><<<256, 768, ldsb>>>(x, B, T, n_rows, ld, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) gemm0_rows<MF><<<256, 768, ldsb>>>(x, B, T, n_rows, ld, 1.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    printf("%-52s: %.4f ms  %.1f TFLOP/s  %.2f TB/s written (%s)\n", name, ms, 2.0 * n_rows * 64 * 576 / ms / 1e9,
           4.0 * n_rows * 576 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

template <int STEP, int PRIO = 0>
static void run_gemm0(const char *name, const float *x, const float *B, float *T, int n_rows, int col_group) {
    const int ld = 576, nct = 18;
    const size_t ldsb = 2 * KH * ld * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm0<STEP, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) gemm0<STEP, PRIO><<<256, 1024, ldsb>>>(x, B, T, n_rows, ld, nct, col_group, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) gemm0<STEP, PRIO><<<256, 1024, ldsb>>>(x, B, T, n_rows, ld, nct, col_group, 1.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    printf("%-40s colgroup %2d: %.4f ms  %.1f TFLOP/s  (%s)\n", name, col_group, ms, 2.0 * n_rows * 64 * 576 / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
}

template <int MODE>
static void run(const char *name, float *out) {
    const int tiles = 36 * 20;
    const size_t ldsb = 2 * KH * LD * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&kern<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kern<MODE><<<256, 1024, ldsb>>>(out, tiles, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<MODE><<<256, 1024, ldsb>>>(out, tiles, 1.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-44s %.3f ms  %.1f TFLOP/s  (%s)\n", name, ms, 256.0 * 16 * tiles * 32 * 4096.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    run<0>("registers only", out);
    run<1>("B from LDS, compiler schedule", out);
    run<2>("B from LDS, next tile prefetched", out);
    run<3>("B from LDS, two chains", out);
    const int N = 273744;
    float *x, *B, *T;
    hipMalloc(&x, (size_t)N * 64 * 4);
    hipMalloc(&B, 64 * 576 * 4);
    hipMalloc(&T, (size_t)N * 576 * 4);
    hipMemset(x, 0, (size_t)N * 64 * 4);
    hipMemset(B, 0, 64 * 576 * 4);
    run_rows<0>("rows-together mapping, stores + A loads only", x, B, T, N);
    run_rows<1>("rows-together mapping, full product", x, B, T, N);
    for (int cg : {5}) {
        run_gemm0<0>("gemm0: items + LDS B + MFMA", x, B, T, N, cg);
        run_gemm0<1>("gemm0: + stores", x, B, T, N, cg);
        run_gemm0<2>("gemm0: + A loads", x, B, T, N, cg);
        run_gemm0<2, 1>("gemm0: + A loads, wave priorities", x, B, T, N, cg);
        run_gemm0<1, 1>("gemm0: + stores, wave priorities", x, B, T, N, cg);
    }
    return 0;
}
