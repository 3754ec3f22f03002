// Do fp32 MFMA chains and a 630 MB store stream overlap on this part when they come from DIFFERENT kernels on two
// streams (no intra-wave ordering at all)?  If the pair takes max(a, b) the transform kernel's epilogue is to blame for
// its poor overlap; if it takes a + b the two pipes throttle each other and no kernel restructuring will help.
// hipcc --offload-arch=gfx950 -O3 profiles/tools/overlap_probe.cpp -o profiles/tools/_bin/overlap_probe
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_only(float *out, int iters, float a0) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = a0 + threadIdx.x, b = 2.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the transform's store pattern: a wave writes a 32-row x 32-column tile, 16 dword stores of 2 x 128 B
__global__ __launch_bounds__(256) void store_only(float *T, int n_rows, int ld, int reps) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int n_tiles = n_rows / 32, nct = ld / 32;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (int rep = 0; rep < reps; ++rep)
        for (int item = wave; item < n_tiles * nct; item += n_waves) {
            const int tile = item / nct, ct = item % nct;
            float *p = T + (size_t)(tile * 32 + 4 * h) * ld + ct * 32 + r;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) p[(size_t)((reg & 3) + 8 * (reg >> 2)) * ld] = (float)(item + reg);
        }
}

// one kernel, specialised waves: waves 0..7 of every workgroup run MFMA chains, waves 8..15 stream the stores
template <int X4>
__global__ __launch_bounds__(1024) void both_roles(float *out, float *T, int iters, int n_rows, int ld, int reps, int roles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 8) {
        if (!(roles & 1)) return;
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float a = 1.f + threadIdx.x, b = 2.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i];
        out[blockIdx.x * 1024 + threadIdx.x] = s;
    } else {
        if (!(roles & 2)) return;
        const int r = lane & 31, h = lane >> 5;
        const int n_tiles = n_rows / 32, nct = ld / 32;
        const int w = blockIdx.x * 8 + (wave - 8), n_waves = gridDim.x * 8;
        for (int rep = 0; rep < reps; ++rep)
            for (int item = w; item < n_tiles * nct; item += n_waves) {
                const int tile = item / nct, ct = item % nct;
                if (X4 == 0) {
                    float *p = T + (size_t)(tile * 32 + 4 * h) * ld + ct * 32 + r;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) p[(size_t)((reg & 3) + 8 * (reg >> 2)) * ld] = (float)(item + reg);
                } else {  // transposed accumulators: lane (r, h) owns row r, columns 8g + 4h .. + 3
                    float *p = T + (size_t)(tile * 32 + r) * ld + ct * 32 + 4 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4 *>(p + 8 * g) = make_float4((float)item, (float)g, 1.f, 2.f);
                }
            }
    }
}

int main() {
    const int N = 273728, ld = 576;
    float *out, *T;
    hipMalloc(&out, 1 << 24);
    hipMalloc(&T, (size_t)N * ld * 4);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 1000;  // 256 CUs x 8 waves x 16000 MFMAs x 64 cycles / 4 SIMDs / 2 waves per SIMD
    auto time = [&](int which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(a, s1);
            hipStreamWaitEvent(s2, a, 0);
            if (which & 1) mfma_only<<<512, 256, 0, s1>>>(out, iters, 1.f);       // 2 waves per SIMD, ~20 VGPRs
            if (which & 2) store_only<<<2048, 256, 0, s2>>>(T, N, ld, 4);          // 4 x 630 MB
            hipEventRecord(b, s2);
            hipStreamWaitEvent(s1, b, 0);
            hipEventRecord(b, s1);
            hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        return ms;
    };
    const float tm = time(1), ts = time(2), tb = time(3);
    printf("MFMA chains alone %.3f ms (%.1f TFLOP/s) | stores alone %.3f ms (%.2f TB/s) | both, two streams %.3f ms  (max %.3f, sum %.3f)\n",
           tm, 512.0 * 4 * iters * 16 * 4096 / tm / 1e9, ts, 4.0 * N * ld * 4 / ts / 1e9, tb, tm > ts ? tm : ts, tm + ts);
    for (int x4 = 0; x4 < 2; ++x4) {
        float t3[4];
        for (int roles = 1; roles <= 3; ++roles) {
            for (int rep = 0; rep < 2; ++rep) {
                hipDeviceSynchronize();
                hipEventRecord(a, s1);
                if (x4) both_roles<1><<<256, 1024, 0, s1>>>(out, T, iters, N, ld, 4, roles);
                else both_roles<0><<<256, 1024, 0, s1>>>(out, T, iters, N, ld, 4, roles);
                hipEventRecord(b, s1);
                hipEventSynchronize(b);
            }
            hipEventElapsedTime(&t3[roles], a, b);
        }
        printf("one kernel, specialised waves, %s stores: MFMA waves alone %.3f ms | store waves alone %.3f ms | both %.3f ms\n",
               x4 ? "float4 (32 rows x 32 B)" : "dword (2 rows x 128 B)", t3[1], t3[2], t3[3]);
    }
    return 0;
}
