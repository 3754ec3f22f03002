// Device helpers shared by the forward (agg.hip) and backward (agg_bwd.hip) aggregation kernels.
#pragma once
#include "common.h"

namespace pea {
namespace {

constexpr int kBlock = 256;
constexpr float kNegBig = -3.0e38f;  // finite "minus infinity" for the running max (no inf-inf NaNs)
constexpr float kLog2e = 1.44269504088896340736f;

struct AggLaunch {
    int n_groups;
    int blk_start[kMaxAggGroups + 1];
    // agg_rows_kernel: workgroups [0, n_long_blocks) walk blk_start (long items), the rest blk_short (short rows)
    int n_long_blocks;
    int blk_short[kMaxAggGroups + 1];
    AggGroup g[kMaxAggGroups];
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 fma4(float w, float4 h, float4 a) {
    return make_float4(fmaf(w, h.x, a.x), fmaf(w, h.y, a.y), fmaf(w, h.z, a.z), fmaf(w, h.w, a.w));
}
__device__ __forceinline__ float4 scale4(float4 a, float f) { return make_float4(a.x * f, a.y * f, a.z * f, a.w * f); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m));
}

// Running softmax-weighted sum for one attention group: value = acc / s with weights exp(e - m).
struct Soft {
    float m, s;
    float4 acc;
    __device__ __forceinline__ void init() { m = kNegBig; s = 0.f; acc = make_float4(0.f, 0.f, 0.f, 0.f); }
    // The running max m lives in the log2 domain, the logits e in NATURAL units exactly as the reference forms them
    // (leaky_relu(x_i . att_i + x_j . att_j), reference GATConv.message): a weight is 2^(e * log2(e) - m), one v_fma_f32 (the
    // product exact inside it) + one v_exp_f32.  Every weight of a state is relative to the same m, whatever m is, so the
    // rounding of m itself cancels in acc / s; only the small difference e * log2(e) - m is rounded.  (Round 2 folded
    // log2(e) into the packed attention vectors instead: one more rounding per vector component, and the exponent inherited
    // the rounding of the scaled logit -- up to 4x the sequential oracle's error on deep multi-head stacks.)
    __device__ __forceinline__ void push(float e, float4 h) {
        const float mn = fmaxf(m, e * kLog2e);
        const float fs = __builtin_amdgcn_exp2f(m - mn);                    // factor on the old state (1 when the max stays)
        const float p = __builtin_amdgcn_exp2f(fmaf(e, kLog2e, -mn));       // weight of the new edge
        m = mn;
        s = fmaf(s, fs, p);
        acc.x = fmaf(acc.x, fs, p * h.x);
        acc.y = fmaf(acc.y, fs, p * h.y);
        acc.z = fmaf(acc.z, fs, p * h.z);
        acc.w = fmaf(acc.w, fs, p * h.w);
    }
    __device__ __forceinline__ void merge(float m2, float s2, float4 a2) {
        const float mn = fmaxf(m, m2);
        const float f1 = __builtin_amdgcn_exp2f(m - mn), f2 = __builtin_amdgcn_exp2f(m2 - mn);
        s = s * f1 + s2 * f2;
        acc = add4(scale4(acc, f1), scale4(a2, f2));
        m = mn;
    }
};

// slope in [0, 1] (checked on the host): leaky_relu(a) = max(a, slope * a)
__device__ __forceinline__ float leaky(float a, float slope) { return fmaxf(a, a * slope); }

// row j of a [rows, ld] fp32 matrix: one 32x32->64 multiply-add, no sign extension
__device__ __forceinline__ const float *row_at(const float *base, int j, int ld) {
    return base + (unsigned long long)(unsigned)j * (unsigned)ld;
}

__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

// Sum of v over the F4 = F/4 consecutive lanes that hold one attention head's columns (every lane of the
// head gets the total).  The logit (x_i.att_i).sum(-1) + (x_j.att_j).sum(-1) of GATConv.message is thus
// computed from the gathered row itself: no per-node attention scalars are stored or gathered.
// F4T > 0: head width known at compile time (a power of two): constant-offset xor shuffles (DPP / swizzle).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {  // v from the lane the DPP control selects (all rows, all banks)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

template <int F4T>
__device__ __forceinline__ float head_sum(float v, int lane, int pos, int F4, bool pow2) {
    if (F4T > 0) {  // all-reduce inside aligned power-of-two lane groups with data-parallel primitives (no LDS crossbar)
        if (F4T >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
        if (F4T >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
        if (F4T >= 8) v += dpp_f<0x141>(v);   // row_half_mirror: the other quad of the 8-lane half
        if (F4T >= 16) v += dpp_f<0x140>(v);  // row_mirror: the other half of the 16-lane row
        if (F4T >= 32) v += __shfl_xor(v, 16);
        if (F4T >= 64) v += __shfl_xor(v, 32);
        return v;
    }
    if (pow2) {
        for (int off = 1; off < F4; off <<= 1) v += __shfl_xor(v, off);
        return v;
    }
    for (int off = 1; off < F4; off <<= 1) {
        const float o = __shfl_down(v, off);
        if (pos + off < F4) v += o;
    }
    return __shfl(v, lane - pos);
}

__device__ __forceinline__ int find_group(const AggLaunch &L) {
    int g = 0;
    while (g + 1 < L.n_groups && (int)blockIdx.x >= L.blk_start[g + 1]) ++g;
    return g;
}


inline int lanes_for(int W) {
    int g = 4;
    while (g * 4 < W) g <<= 1;
    return g;
}

}  // namespace
}  // namespace pea
