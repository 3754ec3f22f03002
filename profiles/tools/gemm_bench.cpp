// Micro-benchmark of the dense-transform launches of the ml25m-shaped step, straight on pea::launch_gemm_batch
// (linked against graph_recsys_benchmark_amd/csrc/libpeahip.so).  Build + run: see profiles/tools/README in profiles/README.md.
//   hipcc --offload-arch=gfx950 -O2 -I graph_recsys_benchmark_amd/csrc -I include profiles/tools/gemm_bench.cpp \
//         -L graph_recsys_benchmark_amd/csrc -lpeahip -Wl,-rpath,'$ORIGIN/../../../graph_recsys_benchmark_amd/csrc' -o profiles/tools/_bin/gemm_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <functional>
#include <vector>

#include "common.h"

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);    \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static float time_ms(hipStream_t st, int iters, const std::function<void()> &f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 273744;
    const int P = 9, K = 64, HF = 64, R = 16;
    const int flags = argc > 2 ? atoi(argv[2]) : 0;  // 1: no edge-less-row mask on GEMM_1; 2: GEMM_1 inputs in separate dense [N,64] tables
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float *x, *T, *O, *X, *B0, *B1, *bias0, *bias1;
    unsigned char *mask;
    CK(hipMalloc(&x, N * K * 4));
    CK(hipMalloc(&T, N * P * HF * 4));
    CK(hipMalloc(&O, N * P * HF * 4));
    CK(hipMalloc(&X, N * P * R * 4));
    CK(hipMalloc(&B0, K * P * HF * 4));
    CK(hipMalloc(&B1, P * K * R * 4));
    CK(hipMalloc(&bias0, P * HF * 4));
    CK(hipMalloc(&bias1, P * R * 4));
    CK(hipMalloc(&mask, N));
    std::vector<float> h(N * K);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(x, h.data(), N * K * 4, hipMemcpyHostToDevice));
    std::vector<float> w(K * P * HF);
    for (auto &v : w) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(B0, w.data(), K * P * HF * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B1, w.data(), P * K * R * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias0, 0, P * HF * 4));
    CK(hipMemset(bias1, 0, P * R * 4));
    std::vector<unsigned char> hm(N);
    for (auto &v : hm) v = (rand() % 10) == 0;
    CK(hipMemcpy(mask, hm.data(), N, hipMemcpyHostToDevice));
    CK(hipMemset(O, 0, N * P * HF * 4));

    // GEMM_0: one job, shared x, 576 output columns
    pea::GemmJob J0{};
    J0.A1 = x;
    J0.lda1 = K;
    J0.K1 = K;
    J0.B = B0;
    J0.ldb = P * HF;
    J0.n_out = P * HF;
    J0.n_seg = 1;
    J0.seg[0] = {0, P * HF, T, P * HF, 0};
    // GEMM_1: nine jobs, A = relu outputs O[:, p*64 : p*64+64] (edge-less rows from T with bias/relu), 16 columns each
    std::vector<pea::GemmJob> J1(P);
    for (int p = 0; p < P; ++p) {
        pea::GemmJob J{};
        J.A1 = (flags & 2) ? O + (size_t)p * N * HF : O + p * HF;
        J.lda1 = (flags & 2) ? HF : P * HF;
        J.K1 = K;
        J.a1_mask = (flags & 1) ? nullptr : mask;
        J.a1_alt = T + p * HF;
        J.lda_alt = P * HF;
        J.a1_bias = bias0 + p * HF;
        J.B = B1 + p * K * R;
        J.ldb = R;
        J.n_out = R;
        J.bias = nullptr;
        J.n_seg = 1;
        J.seg[0] = {0, R, X + p * R, P * R, 0};
        J1[p] = J;
    }
    const float t0 = time_ms(st, 50, [&] { pea::launch_gemm_batch(&J0, 1, nullptr, N, st); });
    const float t1 = time_ms(st, 50, [&] { pea::launch_gemm_batch(J1.data(), P, nullptr, N, st); });
    const double fl0 = 2.0 * N * K * P * HF, fl1 = 2.0 * N * K * P * R;
    const double by0 = 4.0 * N * (K + P * HF), by1 = 4.0 * N * P * (K + R);
    printf("flags %d GEMM_0 %.4f ms  %.1f TFLOP/s  %.2f TB/s | GEMM_1 %.4f ms  %.1f TFLOP/s  %.2f TB/s\n", flags, t0, fl0 / t0 / 1e9,
           by0 / t0 / 1e9, t1, fl1 / t1 / 1e9, by1 / t1 / 1e9);
    {   // determinism: the same launch twice, compared bitwise
        float *T2;
        CK(hipMalloc(&T2, N * P * HF * 4));
        pea::GemmJob J2 = J0;
        J2.seg[0].dst = T2;
        pea::launch_gemm_batch(&J0, 1, nullptr, N, st);
        pea::launch_gemm_batch(&J2, 1, nullptr, N, st);
        CK(hipStreamSynchronize(st));
        std::vector<float> a((size_t)N * P * HF), b((size_t)N * P * HF);
        CK(hipMemcpy(a.data(), T, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), T2, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < a.size(); ++i)
            if (memcmp(&a[i], &b[i], 4) != 0 && bad++ == 0) first = i;
        // host reference of a few rows
        double maxerr = 0;
        for (int64_t row : {(int64_t)0, (int64_t)12345, N - 1}) {
            for (int c = 0; c < P * HF; ++c) {
                float acc = 0.f;
                for (int k = 0; k < K; ++k) acc = fmaf(h[row * K + k], w[(size_t)k * P * HF + c], acc);
                maxerr = std::max(maxerr, (double)fabsf(acc - a[row * P * HF + c]));
            }
        }
        printf("determinism: %zu mismatching floats (first at %zu); max |err| vs host fmaf chain on 3 rows: %.3g\n", bad, first, maxerr);
    }
    // checksum so variants can be compared
    std::vector<float> ht(4096), hx(4096);
    CK(hipMemcpy(ht.data(), T + 12345 * 576, 4096 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), X + 12345 * 144, 4096 * 4, hipMemcpyDeviceToHost));
    double s0 = 0, s1 = 0;
    for (int i = 0; i < 4096; ++i) {
        s0 += ht[i] * (1 + i % 7);
        s1 += hx[i] * (1 + i % 5);
    }
    printf("checksum T %.9g X %.9g\n", s0, s1);
    return 0;
}
