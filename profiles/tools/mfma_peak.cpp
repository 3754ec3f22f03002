// What the fp32 matrix pipe of this part actually sustains: chains of v_mfma_f32_32x32x2_f32 / 16x16x4_f32 with
// nothing else in the loop.  hipcc --offload-arch=gfx950 -O2 profiles/tools/mfma_peak.cpp -o profiles/tools/_bin/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int i = 0; i < 4; ++i) acc[n][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n)
        for (int i = 0; i < 4; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static void run(const char *name, F launch, double flops_per_wave_iter, int waves_per_cu) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 4000;
    launch(iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch(iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double waves = 256.0 * waves_per_cu;
    printf("%-34s waves/CU %2d: %.3f ms  %.1f TFLOP/s\n", name, waves_per_cu, ms, flops_per_wave_iter * iters * waves / ms / 1e9);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 16 * 64 * 4 * 4);
    for (int wpc : {4, 8, 16}) {  // waves per CU (1, 2, 4 per SIMD)
        const int blocks = 256 * wpc / 4;
        run("32x32x2 f32, 1 chain", [&](int it) { k32<1><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 8 * 4096.0, wpc);
        run("32x32x2 f32, 2 chains", [&](int it) { k32<2><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 16 * 4096.0, wpc);
        run("16x16x4 f32, 1 chain", [&](int it) { k16<1><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 8 * 2048.0, wpc);
        run("16x16x4 f32, 4 chains", [&](int it) { k16<4><<<blocks, 256>>>(out, it, 1.f, 2.f); }, 32 * 2048.0, wpc);
    }
    return 0;
}
