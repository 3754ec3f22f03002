// Neighbour aggregation over destination-sorted CSR for gfx950 (wave64).
//
// Replaces, per conv layer, PyG 1.5.0's  index_select -> message -> scatter  pipeline
// (reference call site graph_recsys_benchmark/models/base.py:138-139):
//   AGG_GAT  : leaky_relu logits, per-target softmax (online, single pass), weighted sum, +bias, relu
//   AGG_GCN  : sum_j dinv[j]*dinv[i]*h_j, +bias, relu
//   AGG_MEAN : mean_j x_j (SAGE; the linear terms follow in the GEMM)
// No [M,F] temporaries, no float atomics: every output row is produced by one wave (or by a
// fixed-order merge of its hub chunks), so results are bitwise reproducible run to run.
//
// Thread mapping: a row chunk of W <= 256 fp32 columns is covered by G = 4..64 lanes x float4.
//   short rows  : one G-lane subgroup per destination row (64/G rows per wave), edges in sequence
//   long rows   : one wave per row (or per <=512-edge chunk of a hub row); the wave loads 64 source
//                 ids coalesced, the 64/G subgroups take them round-robin, partial states are merged
//                 with xor shuffles
//   hub rows    : chunks write (m, s, acc) records; a merge kernel folds them in chunk order
#include <algorithm>
#include <cstdlib>

#include "agg_common.h"

namespace pea {

namespace {

template <int MODE>
__device__ __forceinline__ void finish_row(const AggGroup &P, int row, int c4, int deg, Soft st, float4 sum) {
    float4 o;
    if (MODE == AGG_GAT) {
        const float inv = 1.0f / (st.s + 1e-16f);
        o = scale4(st.acc, inv);
        if (P.stats && c4 % P.F == 0) {  // training: keep (max, denominator) of this (row, head) for the backward
            float *sp = P.stats + (size_t)row * P.ld_stats + 2 * (c4 / P.F);
            sp[0] = st.m;
            sp[1] = st.s;
        }
    } else if (MODE == AGG_GCN || MODE == AGG_WSUM) {
        o = sum;
    } else {
        const float inv = 1.0f / (float)(deg < 1 ? 1 : deg);
        o = scale4(sum, inv);
        if (P.accum) o = add4(o, ld4(P.out + (size_t)row * P.ld_out + c4));  // + the root term already in the row
    }
    if (P.bias) o = add4(o, ld4(P.bias + c4));
    if (P.relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    st4(P.out + (size_t)row * P.ld_out + c4, o);
}

// ------------------------------------------------------------------------------------------------
// short rows: one G-lane subgroup per destination row
// ------------------------------------------------------------------------------------------------
template <int G, int MODE, int F4T>
__device__ __forceinline__ void short_rows(const AggGroup &P, const int blk) {
    const int item = blk * (kBlock / G) + (int)threadIdx.x / G;
    const int sl = (int)threadIdx.x % G;
    const bool valid = item < P.n_short;
    const int row = valid ? P.short_rows[item] : 0;
    const bool active = valid && sl * 4 < P.W;
    const int c4 = active ? sl * 4 : 0;
    const int beg = P.rowptr[row];
    const int end = valid ? P.rowptr[row + 1] : beg;
    const float *feat = P.feat + c4;
    const float *feat_self = P.feat_self + c4;

    Soft st;
    st.init();
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int lane = (int)threadIdx.x % kWave;
    const int F4 = P.F / 4, pos = sl % F4;
    const bool pow2 = (F4 & (F4 - 1)) == 0;
    float a_d = 0.f, di = 0.f;
    const float slope = P.neg_slope;  // kept in a register: read inside the edge loop it became a scalar load + wait per edge
    float4 att_s = make_float4(0.f, 0.f, 0.f, 0.f), h_self = att_s;
    if (MODE == AGG_GAT) {
        att_s = ld4(P.att_src + c4);
        h_self = ld4(row_at(feat_self, row, P.ld_self));
        a_d = head_sum<F4T>(dot4(h_self, ld4(P.att_dst + c4)), lane, pos, F4, pow2);
    } else if (MODE == AGG_GCN) {
        di = P.dinv_self[row];
    }
    // the subgroups of a wave walk rows of different length: keep every lane in the loop (the head sums are
    // cross-lane) and mask finished rows instead
    int len = end - beg;
    if (MODE == AGG_GAT) {
        for (int off = G; off < kWave; off <<= 1) len = max(len, __shfl_xor(len, off));
    }
    constexpr int U = 4;  // edges in flight per subgroup: U ids, then U gathered rows, then one softmax update for the U
    for (int t = 0; t < len; t += U) {
        int jj[U];
        bool ok[U];
        float4 h[U];
        float a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = beg + t + u;
            ok[u] = e < end;
            jj[u] = ok[u] ? P.col[e] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            h[u] = ld4(row_at(feat, jj[u], P.ld_feat));
            if (MODE == AGG_GCN) a[u] = P.dinv[jj[u]];
            if (MODE == AGG_WSUM) a[u] = ok[u] ? P.edge_w[P.eid[beg + t + u]] : 0.f;
        }
        if (MODE == AGG_GAT) {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = head_sum<F4T>(dot4(h[u], att_s), lane, pos, F4, pow2);
            float e[U];
            float mx = -INFINITY;                                            // largest logit of the batch, natural units
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float z = leaky(a[u] + a_d, slope);                   // computed for every slot, then selected:
                e[u] = ok[u] ? z : -INFINITY;                                // no branch, 2^(-inf) = 0
                mx = fmaxf(mx, e[u]);
            }
            const float mn = fmaxf(st.m, mx * kLog2e);                       // running max, log2 domain (finite)
            const float fs = __builtin_amdgcn_exp2f(st.m - mn);
            st.m = mn;
            st.s *= fs;
            st.acc = scale4(st.acc, fs);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = __builtin_amdgcn_exp2f(fmaf(e[u], kLog2e, -mn));  // 0 for the masked slots
                st.s += p;
                st.acc = fma4(p, h[u], st.acc);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == AGG_GCN) {
                    sum = fma4(ok[u] ? a[u] * di : 0.f, h[u], sum);
                } else if (MODE == AGG_WSUM) {
                    sum = fma4(a[u], h[u], sum);
                } else {
                    if (ok[u]) sum = add4(sum, h[u]);
                }
            }
        }
    }
    if (P.self_loop) {
        if (MODE == AGG_GAT) {
            const float a = head_sum<F4T>(dot4(h_self, att_s), lane, pos, F4, pow2);
            st.push(leaky(a + a_d, slope), h_self);
        } else if (MODE == AGG_GCN) {
            sum = fma4(di * di, ld4(row_at(feat_self, row, P.ld_self)), sum);
        }
    }
    if (active) finish_row<MODE>(P, row, c4, end - beg, st, sum);
}

// ------------------------------------------------------------------------------------------------
// long rows and hub chunks: one wave per item
// ------------------------------------------------------------------------------------------------
// HOT: the K most frequent sources of the relation sit in the workgroup's LDS image (K rows x W columns, filled once per
// workgroup); a CSR entry <= -2 names image row -(j + 2), anything >= 0 is read from memory as before.  Same edges, same
// order, same values: results are bitwise those of the plain kernel.
template <int G, int MODE, int F4T, int U, bool HOT>
__device__ __forceinline__ void long_item(const AggGroup &P, const LongItem it, const int lane, const float4 *img,
                                          const float *hot_dinv) {
    constexpr int NSG = kWave / G;
    const int sub = lane / G, sl = lane % G;
    const bool active = sl * 4 < P.W;
    const int c4 = active ? sl * 4 : 0;
    const float *feat = P.feat + c4;
    const float *feat_self = P.feat_self + c4;
    const int row = it.row;
    const int *col = HOT ? P.hot_col : P.col;
    const int W4 = P.W / 4;
    const int lc = active ? sl : 0;   // this lane's float4 column inside an image row

    Soft st;
    st.init();
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int kk = 2 * (c4 / P.F);
    const int F4 = P.F / 4, pos = sl % F4;
    const bool pow2 = (F4 & (F4 - 1)) == 0;
    float a_d = 0.f, di = 0.f;
    const float slope = P.neg_slope;  // kept in a register: read inside the edge loop it became a scalar load + wait per edge
    float4 att_s = make_float4(0.f, 0.f, 0.f, 0.f), h_self = att_s;
    if (MODE == AGG_GAT) {
        att_s = ld4(P.att_src + c4);
        h_self = ld4(row_at(feat_self, row, P.ld_self));
        a_d = head_sum<F4T>(dot4(h_self, ld4(P.att_dst + c4)), lane, pos, F4, pow2);
    } else if (MODE == AGG_GCN) {
        di = P.dinv_self[row];
    }
    // edges in flight per subgroup: U*NSG gathered rows per wave before the first use
    int src = it.beg + lane < it.end ? col[it.beg + lane] : -1;
    for (int base = it.beg; base < it.end; base += kWave) {
        const int nxt = base + kWave + lane;
        const int src_next = nxt < it.end ? col[nxt] : -1;  // next batch of ids is in flight during this one
        const int cnt = min(kWave, it.end - base);
        for (int t = 0; t < cnt; t += NSG * U) {
            int jj[U];
            bool ok[U];
            float4 h[U];
            float a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = t + u * NSG + sub;
                const int j = __shfl(src, idx & (kWave - 1));
                ok[u] = idx < cnt && j != -1;
                jj[u] = ok[u] ? j : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (HOT && jj[u] < 0) {  // image row (ds_read_b128), no memory traffic
                    const int r = -(jj[u] + 2);
                    h[u] = img[r * W4 + lc];
                    if (MODE == AGG_GCN) a[u] = hot_dinv[r];
                } else {
                    h[u] = ld4(row_at(feat, jj[u], P.ld_feat));
                    if (MODE == AGG_GCN) a[u] = P.dinv[jj[u]];
                }
                if (MODE == AGG_WSUM) a[u] = ok[u] ? P.edge_w[P.eid[base + t + u * NSG + sub]] : 0.f;
            }
            if (MODE == AGG_GAT) {
#pragma unroll
                for (int u = 0; u < U; ++u) a[u] = head_sum<F4T>(dot4(h[u], att_s), lane, pos, F4, pow2);
            }
            if (MODE == AGG_GAT) {
                // softmax updates in batches of 4 edges (the batch size is part of the arithmetic: it stays 4 whatever
                // U is, so every variant of the kernel produces the same bits): shared new max, one rescale, 4 weights
#pragma unroll
                for (int u0 = 0; u0 < U; u0 += 4) {
                    float e[4];
                    float mx = -INFINITY;                                        // largest logit of the batch, natural units
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float z = leaky(a[u0 + u] + a_d, slope);          // computed for every slot, then selected:
                        e[u] = ok[u0 + u] ? z : -INFINITY;                       // no branch, 2^(-inf) = 0
                        mx = fmaxf(mx, e[u]);
                    }
                    const float mn = fmaxf(st.m, mx * kLog2e);                   // running max, log2 domain (finite)
                    const float fs = __builtin_amdgcn_exp2f(st.m - mn);
                    st.m = mn;
                    st.s *= fs;
                    st.acc = scale4(st.acc, fs);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(e[u], kLog2e, -mn));  // 0 for the masked slots
                        st.s += p;
                        st.acc = fma4(p, h[u0 + u], st.acc);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (MODE == AGG_GCN) {
                        sum = fma4(ok[u] ? a[u] * di : 0.f, h[u], sum);
                    } else if (MODE == AGG_WSUM) {
                        sum = fma4(a[u], h[u], sum);
                    } else {
                        if (ok[u]) sum = add4(sum, h[u]);
                    }
                }
            }
        }
        src = src_next;
    }
    // fold the NSG subgroup states (xor butterfly: every lane ends with the same value)
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
        if (MODE == AGG_GAT) {
            const float m2 = __shfl_xor(st.m, off), s2 = __shfl_xor(st.s, off);
            const float4 a2 = shfl_xor4(st.acc, off);
            st.merge(m2, s2, a2);
        } else {
            sum = add4(sum, shfl_xor4(sum, off));
        }
    }
    if (it.slot >= 0) {  // hub chunk: write the partial record, the merge kernel finishes the row
        if (sub == 0 && active) {
            const int nk = P.W / P.F;
            float *rec = P.partial + (size_t)it.slot * (size_t)(P.W + 2 * nk);
            if (MODE == AGG_GAT) {
                st4(rec + c4, st.acc);
                if (c4 % P.F == 0) {
                    rec[P.W + kk] = st.m;
                    rec[P.W + kk + 1] = st.s;
                }
            } else {
                st4(rec + c4, sum);
            }
        }
        return;
    }
    if (P.self_loop) {
        if (MODE == AGG_GAT) {
            const float a = head_sum<F4T>(dot4(h_self, att_s), lane, pos, F4, pow2);
            st.push(leaky(a + a_d, slope), h_self);
        } else if (MODE == AGG_GCN) {
            sum = fma4(di * di, ld4(row_at(feat_self, row, P.ld_self)), sum);
        }
    }
    if (sub == 0 && active) finish_row<MODE>(P, row, c4, it.end - it.beg, st, sum);
}

// One launch per (lane width, head class) of a level: workgroups [0, n_long_blocks) take the long rows and hub chunks
// (one wave per item; heaviest work first), the workgroups after them the short rows (one G-lane subgroup per row), which
// fill the machine while the last long items drain.  Round 2: two launches before (agg_long_* / agg_short_*); on a rank of
// 8 the short-row launches were 12 us each for 2 us of work.
template <int G, int MODE, int F4T>
__global__ __launch_bounds__(kBlock) void agg_rows_kernel(const AggLaunch L) {
    if ((int)blockIdx.x >= L.n_long_blocks) {
        const int b = (int)blockIdx.x - L.n_long_blocks;
        int gi = 0;
        while (gi + 1 < L.n_groups && b >= L.blk_short[gi + 1]) ++gi;
        short_rows<G, MODE, F4T>(L.g[gi], b - L.blk_short[gi]);
        return;
    }
    const int gi = find_group(L);
    const AggGroup &P = L.g[gi];
    const int wave = (int)threadIdx.x / kWave;
    const int lane = (int)threadIdx.x % kWave;
    const int item = ((int)blockIdx.x - L.blk_start[gi]) * (kBlock / kWave) + wave;
    if (item >= P.n_long) return;  // wave-uniform
    const LongItem it = P.long_items[item];
    if (it.slot == -2) return;  // padding of the XCD-affine item layout (plan.hip)
    long_item<G, MODE, F4T, 4, false>(P, it, lane, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// "fat lanes": the same item with V = 4 * V4 consecutive columns per lane instead of 4.
// The thin kernel spends 12 (W = 112) / 8 (W = 64) vector instructions per edge, most of them per-edge bookkeeping that
// every lane of a row repeats (addressing, logit pipeline, mask, exp), and sits at 86 % / 68 % VALU-busy on the
// 25m-shaped graph (profiles/r02/sq_counters_agg_long_plain_r02.json).  With 16 columns per lane a row of W columns needs
// W / 16 lanes, a wave works on 64 / G edges per instruction instead of 64 / (4 G), the head sum of the logit shrinks
// (HL = F / V lanes per head; one lane per 16-wide channel: no cross-lane step at all), and every lane reads 64 contiguous
// bytes of its row.  Same edge order and the same 4-edge softmax batches as the thin kernel; the per-lane dot products
// group the columns differently, so logits differ in the last bits between the two (each row is always handled by the
// same one of them: the choice depends on the group's W and F only).
template <int G, int MODE, int HL, int V4>
__device__ __forceinline__ void long_item_fat(const AggGroup &P, const LongItem it, const int lane) {
    constexpr int NSG = kWave / G, U = 4, V = 4 * V4;
    const int sub = lane / G, sl = lane % G;
    const bool active = sl * V < P.W;
    // the HL lanes of a head interleave its float4 chunks: lane (head h, position q) holds chunks q, q + HL, q + 2 HL, ...
    // of the head, so ONE load instruction reads HL * 16 contiguous bytes per edge (64+ bytes for HL >= 4); with 16
    // contiguous columns per lane every instruction touched every cache line of the row (measured 2x slower)
    const int pos = sl % HL;
    const int c0 = active ? (sl / HL) * P.F + 4 * pos : 0;   // column of chunk 0; chunk v sits at c0 + 4 * HL * v
    constexpr int CS = 4 * HL;
    const float *feat = P.feat + c0;
    const float *feat_self = P.feat_self + c0;
    const int row = it.row;
    const int kk = 2 * (c0 / P.F);

    float sm = kNegBig, ss = 0.f;   // running softmax state of this lane's head (GAT)
    float4 acc[V4];                 // weighted sum (GAT) / plain sum (GCN, MEAN)
#pragma unroll
    for (int v = 0; v < V4; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    float a_d = 0.f, di = 0.f;
    const float slope = P.neg_slope;
    float4 att_s[V4], h_self[V4];
    if (MODE == AGG_GAT) {
        float d = 0.f;
#pragma unroll
        for (int v = 0; v < V4; ++v) {
            att_s[v] = ld4(P.att_src + c0 + CS * v);
            h_self[v] = ld4(row_at(feat_self, row, P.ld_self) + CS * v);
            d += dot4(h_self[v], ld4(P.att_dst + c0 + CS * v));
        }
        a_d = head_sum<HL>(d, lane, pos, HL, true);
    } else if (MODE == AGG_GCN) {
        di = P.dinv_self[row];
    }
    int src = it.beg + lane < it.end ? P.col[it.beg + lane] : -1;
    for (int base = it.beg; base < it.end; base += kWave) {
        const int nxt = base + kWave + lane;
        const int src_next = nxt < it.end ? P.col[nxt] : -1;  // next batch of ids is in flight during this one
        const int cnt = min(kWave, it.end - base);
        for (int t = 0; t < cnt; t += NSG * U) {
            int jj[U];
            bool ok[U];
            float4 h[U][V4];
            float a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = t + u * NSG + sub;
                const int j = __shfl(src, idx & (kWave - 1));
                ok[u] = idx < cnt && j >= 0;
                jj[u] = ok[u] ? j : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float *p = row_at(feat, jj[u], P.ld_feat);
#pragma unroll
                for (int v = 0; v < V4; ++v) h[u][v] = ld4(p + CS * v);
                if (MODE == AGG_GCN) a[u] = P.dinv[jj[u]];
            }
            if (MODE == AGG_GAT) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float d = 0.f;
#pragma unroll
                    for (int v = 0; v < V4; ++v) d += dot4(h[u][v], att_s[v]);
                    a[u] = head_sum<HL>(d, lane, pos, HL, true);
                }
                // one softmax update per 4 edges, as in the thin kernel
                float e[U];
                float mx = -INFINITY;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float z = leaky(a[u] + a_d, slope);
                    e[u] = ok[u] ? z : -INFINITY;
                    mx = fmaxf(mx, e[u]);
                }
                const float mn = fmaxf(sm, mx * kLog2e);    // log2-domain running max, natural logits (agg_common.h: Soft)
                const float fs = __builtin_amdgcn_exp2f(sm - mn);
                sm = mn;
                ss *= fs;
#pragma unroll
                for (int v = 0; v < V4; ++v) acc[v] = scale4(acc[v], fs);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(e[u], kLog2e, -mn));
                    ss += p;
#pragma unroll
                    for (int v = 0; v < V4; ++v) acc[v] = fma4(p, h[u][v], acc[v]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (MODE == AGG_GCN) {
                        const float w = ok[u] ? a[u] * di : 0.f;
#pragma unroll
                        for (int v = 0; v < V4; ++v) acc[v] = fma4(w, h[u][v], acc[v]);
                    } else if (ok[u]) {
#pragma unroll
                        for (int v = 0; v < V4; ++v) acc[v] = add4(acc[v], h[u][v]);
                    }
                }
            }
        }
        src = src_next;
    }
    // fold the NSG subgroup states (xor butterfly: every lane ends with the same value)
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
        if (MODE == AGG_GAT) {
            const float m2 = __shfl_xor(sm, off), s2 = __shfl_xor(ss, off);
            const float mn = fmaxf(sm, m2);
            const float f1 = __builtin_amdgcn_exp2f(sm - mn), f2 = __builtin_amdgcn_exp2f(m2 - mn);
            ss = ss * f1 + s2 * f2;
#pragma unroll
            for (int v = 0; v < V4; ++v) acc[v] = add4(scale4(acc[v], f1), scale4(shfl_xor4(acc[v], off), f2));
            sm = mn;
        } else {
#pragma unroll
            for (int v = 0; v < V4; ++v) acc[v] = add4(acc[v], shfl_xor4(acc[v], off));
        }
    }
    if (it.slot >= 0) {  // hub chunk: partial record in plain column order (the merge kernel is layout-agnostic)
        if (sub == 0 && active) {
            const int nk = P.W / P.F;
            float *rec = P.partial + (size_t)it.slot * (size_t)(P.W + 2 * nk);
#pragma unroll
            for (int v = 0; v < V4; ++v) st4(rec + c0 + CS * v, acc[v]);
            if (MODE == AGG_GAT && pos == 0) {
                rec[P.W + kk] = sm;
                rec[P.W + kk + 1] = ss;
            }
        }
        return;
    }
    if (P.self_loop) {
        if (MODE == AGG_GAT) {
            float d = 0.f;
#pragma unroll
            for (int v = 0; v < V4; ++v) d += dot4(h_self[v], att_s[v]);
            const float e = leaky(head_sum<HL>(d, lane, pos, HL, true) + a_d, slope);
            // Soft::push
            const float mn = fmaxf(sm, e * kLog2e);
            const float fs = __builtin_amdgcn_exp2f(sm - mn), p = __builtin_amdgcn_exp2f(fmaf(e, kLog2e, -mn));
            sm = mn;
            ss = fmaf(ss, fs, p);
#pragma unroll
            for (int v = 0; v < V4; ++v) {
                acc[v].x = fmaf(acc[v].x, fs, p * h_self[v].x);
                acc[v].y = fmaf(acc[v].y, fs, p * h_self[v].y);
                acc[v].z = fmaf(acc[v].z, fs, p * h_self[v].z);
                acc[v].w = fmaf(acc[v].w, fs, p * h_self[v].w);
            }
        } else if (MODE == AGG_GCN) {
#pragma unroll
            for (int v = 0; v < V4; ++v) acc[v] = fma4(di * di, ld4(row_at(feat_self, row, P.ld_self) + CS * v), acc[v]);
        }
    }
    if (sub == 0 && active) {
        Soft st;
        st.m = sm;
        st.s = ss;
#pragma unroll
        for (int v = 0; v < V4; ++v) {   // finish_row per float4 chunk (stats are written by the head's first chunk)
            st.acc = acc[v];
            finish_row<MODE>(P, row, c0 + CS * v, it.end - it.beg, st, acc[v]);
        }
    }
}

template <int G, int MODE, int HL, int V4>
__global__ __launch_bounds__(kBlock) void agg_long_fat_kernel(const AggLaunch L) {
    const int gi = find_group(L);
    const AggGroup &P = L.g[gi];
    const int wave = (int)threadIdx.x / kWave;
    const int lane = (int)threadIdx.x % kWave;
    const int item = ((int)blockIdx.x - L.blk_start[gi]) * (kBlock / kWave) + wave;
    if (item >= P.n_long) return;  // wave-uniform
    const LongItem it = P.long_items[item];
    if (it.slot == -2) return;  // padding of the XCD-affine item layout (plan.hip)
    long_item_fat<G, MODE, HL, V4>(P, it, lane);
}

// One 1024-thread workgroup per CU, persistent: it fills the LDS image once, then its 16 waves walk the item list.  The
// walk keeps the launch order's placement: the plain kernel hands items [4b, 4b + 4) to workgroup b, workgroups are dealt
// round-robin over the 8 XCDs, and the plan lays sliced hub rows out so that workgroup b works on source slice b % 8.  Here
// physical workgroup B plays the virtual workgroups  v = B % 8 + 8 * (4 * (B / 8) + wave / 4) + k * 4 * gridDim
// (v % 8 == B % 8: same XCD as before; the grid is a multiple of 8).
constexpr int kHotBlock = 1024;
constexpr int kHotU = 8;
template <int G, int MODE, int F4T>
__global__ __launch_bounds__(kHotBlock) void agg_long_hot_kernel(const AggLaunch L) {
    extern __shared__ float4 hot_img[];
    const AggGroup &P = L.g[0];
    const int W4 = P.W / 4;
    for (int idx = (int)threadIdx.x; idx < P.hot_K * W4; idx += kHotBlock) {
        const int r = idx / W4, c = idx - r * W4;
        hot_img[idx] = ld4(row_at(P.feat, P.hot_nodes[r], P.ld_feat) + 4 * c);
    }
    float *hot_dinv = reinterpret_cast<float *>(hot_img + P.hot_K * W4);
    if (MODE == AGG_GCN)
        for (int r = (int)threadIdx.x; r < P.hot_K; r += kHotBlock) hot_dinv[r] = P.dinv[P.hot_nodes[r]];
    __syncthreads();
    const int wave = (int)threadIdx.x / kWave;
    const int lane = (int)threadIdx.x % kWave;
    const int n_virtual = (P.n_long + 3) / 4;
    const int stride = (int)gridDim.x * 4;
    for (int v = (int)blockIdx.x % 8 + 8 * (4 * ((int)blockIdx.x / 8) + wave / 4); v < n_virtual; v += stride) {
        const int item = v * 4 + wave % 4;
        if (item >= P.n_long) continue;
        const LongItem it = P.long_items[item];
        if (it.slot == -2) continue;
        long_item<G, MODE, F4T, kHotU, true>(P, it, lane, hot_img, hot_dinv);
    }
}

// ------------------------------------------------------------------------------------------------
// hub rows: fold the chunk records in chunk order, add the self loop, finish
// ------------------------------------------------------------------------------------------------
template <int G, int MODE, int F4T>
__global__ __launch_bounds__(kBlock) void agg_merge_kernel(const AggLaunch L) {
    constexpr int NSG = kWave / G;
    const int gi = find_group(L);
    const AggGroup &P = L.g[gi];
    const int wave = (int)threadIdx.x / kWave;
    const int lane = (int)threadIdx.x % kWave;
    const int item = ((int)blockIdx.x - L.blk_start[gi]) * (kBlock / kWave) + wave;
    if (item >= P.n_hub) return;
    const int row = P.hub_rows[item];
    const int first = P.hub_first[item], count = P.hub_count[item];
    const int sub = lane / G, sl = lane % G;
    const bool active = sl * 4 < P.W;
    const int c4 = active ? sl * 4 : 0;
    const int nk = P.W / P.F;
    const size_t rec_sz = (size_t)(P.W + 2 * nk);
    const int kk = 2 * (c4 / P.F);

    Soft st;
    st.init();
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    // a hub row of the 25m-shaped graph has ~300 chunk records: 8 of them are in flight per subgroup before the first
    // merge (folded in the same chunk order as one at a time: the kernel was one dependent load per record, 30 us)
    constexpr int UM = 8;
    for (int c0 = sub; c0 < count; c0 += NSG * UM) {
        float4 a[UM];
        float m2[UM], s2[UM];
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            const int c = c0 + u * NSG;
            const bool ok = c < count;
            const float *rec = P.partial + (size_t)(first + (ok ? c : c0)) * rec_sz;
            a[u] = ld4(rec + c4);
            if (MODE == AGG_GAT) {
                m2[u] = rec[P.W + kk];
                s2[u] = rec[P.W + kk + 1];
            }
            // a slot past the end becomes the neutral record (weight 2^(kNegBig - m) = 0, state x 1): the merges below
            // stay unconditional, so the compiler keeps the UM loads ahead of them instead of sinking each into its branch
            if (!ok) {
                a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                m2[u] = kNegBig;
                s2[u] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            if (MODE == AGG_GAT) {
                st.merge(m2[u], s2[u], a[u]);
            } else {
                sum = add4(sum, a[u]);
            }
        }
    }
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
        if (MODE == AGG_GAT) {
            const float m2 = __shfl_xor(st.m, off), s2 = __shfl_xor(st.s, off);
            const float4 a2 = shfl_xor4(st.acc, off);
            st.merge(m2, s2, a2);
        } else {
            sum = add4(sum, shfl_xor4(sum, off));
        }
    }
    const float *feat_self = P.feat_self + c4;
    if (P.self_loop) {
        const float4 h = ld4(row_at(feat_self, row, P.ld_self));
        if (MODE == AGG_GAT) {
            const int F4 = P.F / 4, pos = sl % F4;
            const bool pow2 = (F4 & (F4 - 1)) == 0;
            const float a_d = head_sum<F4T>(dot4(h, ld4(P.att_dst + c4)), lane, pos, F4, pow2);
            const float a = head_sum<F4T>(dot4(h, ld4(P.att_src + c4)), lane, pos, F4, pow2);
            st.push(leaky(a + a_d, P.neg_slope), h);
        } else if (MODE == AGG_GCN) {
            const float di = P.dinv_self[row];
            sum = fma4(di * di, h, sum);
        }
    }
    const int deg = P.rowptr[row + 1] - P.rowptr[row];
    if (sub == 0 && active) finish_row<MODE>(P, row, c4, deg, st, sum);
}


// profile names, as profiles/summarize.py derives them from rocprofv3's kernel names: agg_rows_g<G>_<mode> = the level
// launch (long rows, hub chunks and short rows), agg_merge_* / agg_longhot_*
template <int G, int MODE>
const char *kname(int which) {
    static char names[4][32];
    static bool init = false;
    if (!init) {
        const char *w[4] = {"short", "rows", "merge", "longhot"};
        const char *m = MODE == AGG_GAT ? "gat" : MODE == AGG_GCN ? "gcn" : MODE == AGG_WSUM ? "wsum" : "mean";
        for (int i = 0; i < 4; ++i) snprintf(names[i], sizeof(names[i]), "agg_%s_g%d_%s", w[i], G, m);
        init = true;
    }
    return names[which];
}

template <int G, int MODE, int F4T>
int launch_for_g(const AggLaunch &base, const int *sel, int n_sel, hipStream_t stream) {
    AggLaunch L;
    // algorithmic bytes of SURVEY.md 8(d) attributed to a launch: per message 4 B per gathered feature column
    // + 4 B source index + 4 B per attention scalar (GAT) / norm scalar (GCN), times the messages it reduces
    double bytes_short = 0.0;
    // ... and what the launch itself pulls through the memory system: every message's row chunk + its source index
    // (GCN: + the norm scalar), each once; `table`: the largest gather footprint among the launch's groups
    double pull_short = 0.0, table = 0.0;
    for (int i = 0; i < n_sel; ++i) {
        const AggGroup &g = base.g[sel[i]];
        const double per_msg = 4.0 * g.W + 4.0 * g.idx_share + (MODE == AGG_GAT ? 4.0 * (g.W / g.F) : MODE == AGG_GCN ? 4.0 * g.idx_share : 0.0);
        bytes_short += per_msg * g.msgs_short;
        const double pull = 4.0 * g.W + 4.0 + (MODE == AGG_GCN ? 4.0 : 0.0);
        pull_short += pull * g.msgs_short;
        table = table > g.table_rows * 4.0 * g.W ? table : g.table_rows * 4.0 * g.W;
    }
    // long rows + hub chunks first, short rows behind them, ONE launch (groups with an LDS image or already served by the
    // fat-lane kernel keep their short rows here, their long items go elsewhere)
    L.n_groups = 0;
    int blocks = 0, sblocks = 0;
    double pull_plain = 0.0, alg_plain = 0.0;
    for (int i = 0; i < n_sel; ++i) {
        const AggGroup &g = base.g[sel[i]];
        const bool long_here = g.n_long > 0 && !g.hot_col && !g.skip_long;
        if (g.n_short <= 0 && !long_here) continue;
        L.blk_start[L.n_groups] = blocks;
        L.blk_short[L.n_groups] = sblocks;
        AggGroup &c = L.g[L.n_groups++];
        c = g;
        if (!long_here) c.n_long = 0;
        if (g.n_short > 0) sblocks += (g.n_short + (kBlock / G) - 1) / (kBlock / G);
        if (!long_here) continue;
        blocks += ((g.n_long + 3) / 4 + 7) / 8 * 8;  // groups start on a multiple of 8 workgroups (XCD round-robin)
        const double per_msg = 4.0 * g.W + 4.0 * g.idx_share + (MODE == AGG_GAT ? 4.0 * (g.W / g.F) : MODE == AGG_GCN ? 4.0 * g.idx_share : 0.0);
        alg_plain += per_msg * g.msgs_long;
        pull_plain += (4.0 * g.W + 4.0 + (MODE == AGG_GCN ? 4.0 : 0.0)) * g.msgs_long;
    }
    L.blk_start[L.n_groups] = blocks;
    L.blk_short[L.n_groups] = sblocks;
    L.n_long_blocks = blocks;
    if (blocks + sblocks > 0) {
        ProfScope ps(kname<G, MODE>(1), stream, alg_plain + bytes_short, pull_plain + pull_short, table);
        PEA_LAUNCH((agg_rows_kernel<G, MODE, F4T>), dim3(blocks + sblocks), dim3(kBlock), 0, stream, L);
        PEA_HIP(hipGetLastError());
    }
    // groups with an LDS image of their hottest sources: one persistent launch each
    for (int i = 0; i < n_sel; ++i) {
        const AggGroup &g = base.g[sel[i]];
        if (g.n_long <= 0 || !g.hot_col || g.skip_long) continue;
        if (MODE == AGG_WSUM) return PEA_ERR_ARG;  // never planned (per-edge weights are indexed by CSR slot)
        L.n_groups = 1;
        L.blk_start[0] = 0;
        L.g[0] = g;
        const size_t lds = (size_t)g.hot_K * g.W * sizeof(float) + (MODE == AGG_GCN ? (size_t)g.hot_K * sizeof(float) : 0);
        static size_t lds_set = 0;  // per instantiation: raise the dynamic-LDS cap once (160 KiB per workgroup on gfx950)
        if (lds > lds_set) {
            PEA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_long_hot_kernel<G, MODE, F4T>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_set = lds;
        }
        const int n_virtual = (g.n_long + 3) / 4;
        const int grid = std::max(8, std::min(256, (n_virtual + 3) / 4 + 7 & ~7));
        const double per_msg = 4.0 * g.W + 4.0 * g.idx_share + (MODE == AGG_GAT ? 4.0 * (g.W / g.F) : MODE == AGG_GCN ? 4.0 * g.idx_share : 0.0);
        // pulled through the memory system: the cold rows, every index, and one image fill per workgroup
        const double pull = (4.0 * g.W + (MODE == AGG_GCN ? 4.0 : 0.0)) * g.msgs_long * (1.0 - g.hot_frac) + 4.0 * g.msgs_long +
                            (double)grid * (double)lds;
        ProfScope ps(kname<G, MODE>(3), stream, per_msg * g.msgs_long, pull, g.table_rows * 4.0 * g.W);
        PEA_LAUNCH((agg_long_hot_kernel<G, MODE, F4T>), dim3(grid), dim3(kHotBlock), lds, stream, L);
        PEA_HIP(hipGetLastError());
    }
    // hub merge
    L.n_groups = 0;
    blocks = 0;
    for (int i = 0; i < n_sel; ++i) {
        const AggGroup &g = base.g[sel[i]];
        if (g.n_hub <= 0) continue;
        L.blk_start[L.n_groups] = blocks;
        L.g[L.n_groups++] = g;
        blocks += (g.n_hub + 3) / 4;
    }
    L.blk_start[L.n_groups] = blocks;
    if (blocks > 0) {
        ProfScope ps(kname<G, MODE>(2), stream, 0.0);
        PEA_LAUNCH((agg_merge_kernel<G, MODE, F4T>), dim3(blocks), dim3(kBlock), 0, stream, L);
        PEA_HIP(hipGetLastError());
    }
    return PEA_OK;
}

// head-width class of a group: the common widths get compile-time shuffles
int f4_class(int mode, int G, int F) {
    if (mode != AGG_GAT) return 0;
    const int f4 = F / 4;
    return f4 == G ? G : f4 == 4 ? 4 : 0;
}

template <int G, int MODE>
int launch_g(const AggLaunch &base, int cls, const int *sel, int n, hipStream_t stream) {
    if (MODE == AGG_GAT && cls == G) return launch_for_g<G, MODE, G>(base, sel, n, stream);
    if (MODE == AGG_GAT && cls == 4) return launch_for_g<G, MODE, (G >= 4 ? 4 : 0)>(base, sel, n, stream);
    return launch_for_g<G, MODE, 0>(base, sel, n, stream);
}

// ---- fat-lane long-row launches: groups whose heads are 64, 128 or 256 columns wide.  A head is spread over HL >= 8
// lanes so that one load instruction reads >= 128 contiguous bytes per edge (a whole cache line: with 4 lanes per head,
// 64-byte pieces, the kernel touched every line twice and lost to the thin kernel): 8 columns per lane for F = 64,
// 16 for F = 128 / 256.
template <int MODE>
const char *fat_name(int G) {
    static char names[4][32];
    const int i = G == 8 ? 0 : G == 16 ? 1 : G == 32 ? 2 : 3;
    if (!names[i][0]) {
        const char *m = MODE == AGG_GAT ? "gat" : MODE == AGG_GCN ? "gcn" : "mean";
        snprintf(names[i], sizeof(names[i]), "agg_long_fat_g%d_%s", G, m);
    }
    return names[i];
}

template <int G, int MODE, int HL, int V4>
int launch_fat(const AggLaunch &base, const int *sel, int n_sel, hipStream_t stream) {
    AggLaunch L;
    L.n_groups = 0;
    int blocks = 0;
    double alg = 0.0, pull = 0.0, table = 0.0;
    for (int i = 0; i < n_sel; ++i) {
        const AggGroup &g = base.g[sel[i]];
        L.blk_start[L.n_groups] = blocks;
        L.g[L.n_groups++] = g;
        blocks += ((g.n_long + 3) / 4 + 7) / 8 * 8;  // same item -> workgroup -> XCD placement as the thin kernel
        const double per_msg = 4.0 * g.W + 4.0 * g.idx_share + (MODE == AGG_GAT ? 4.0 * (g.W / g.F) : MODE == AGG_GCN ? 4.0 * g.idx_share : 0.0);
        alg += per_msg * g.msgs_long;
        pull += (4.0 * g.W + 4.0 + (MODE == AGG_GCN ? 4.0 : 0.0)) * g.msgs_long;
        table = std::max(table, g.table_rows * 4.0 * g.W);
    }
    L.blk_start[L.n_groups] = blocks;
    if (blocks <= 0) return PEA_OK;
    ProfScope ps(fat_name<MODE>(G), stream, alg, pull, table);
    PEA_LAUNCH((agg_long_fat_kernel<G, MODE, HL, V4>), dim3(blocks), dim3(kBlock), 0, stream, L);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

// Launches the long items of every eligible group with the fat-lane kernel and marks those groups (skip_long).
template <int MODE>
int launch_fat_groups(AggLaunch &base, hipStream_t stream) {
    if (MODE == AGG_WSUM) return PEA_OK;
    // OFF unless PEA_FAT=1: measured neutral to slower (round 2): 25m-shaped first-layer gather
    // 0.534 vs 0.537 ms, Yelp SAGE long rows 0.075 vs 0.092 ms, stress 9-channel gather 3.98 vs 3.35 ms -- the thin kernel's
    // 68-86 % VALU-busy overlaps its memory time, it is not the limiter.  Kept (and tested) as the record of that finding.
    const char *env = getenv("PEA_FAT");
    if (!(env && atoi(env) != 0)) return PEA_OK;
    // (lanes per edge G, lanes per head HL, float4 chunks per lane V4) classes
    const int classes[6][3] = {{8, 8, 2}, {16, 8, 2}, {32, 8, 2}, {8, 8, 4}, {16, 8, 4}, {16, 16, 4}};
    int cls[kMaxAggGroups];
    for (int i = 0; i < base.n_groups; ++i) {
        const AggGroup &g = base.g[i];
        cls[i] = -1;
        if (g.n_long <= 0 || g.hot_col) continue;
        const int F = MODE == AGG_GAT ? g.F : g.W;   // GCN / MEAN: the whole row is one "head"
        if (g.W % F != 0 || g.W > 256) continue;
        const int v4 = F == 64 ? 2 : (F == 128 || F == 256) ? 4 : 0;
        if (!v4) continue;
        const int hl = F / (4 * v4), G = g.W / (4 * v4);
        for (int c = 0; c < 6; ++c)
            if (classes[c][0] == G && classes[c][1] == hl && classes[c][2] == v4) cls[i] = c;
    }
    for (int c = 0; c < 6; ++c) {
        int sel[kMaxAggGroups], n = 0;
        for (int i = 0; i < base.n_groups; ++i)
            if (cls[i] == c) sel[n++] = i;
        if (!n) continue;
        switch (c) {
            case 0: PEA_TRY((launch_fat<8, MODE, 8, 2>(base, sel, n, stream))); break;
            case 1: PEA_TRY((launch_fat<16, MODE, 8, 2>(base, sel, n, stream))); break;
            case 2: PEA_TRY((launch_fat<32, MODE, 8, 2>(base, sel, n, stream))); break;
            case 3: PEA_TRY((launch_fat<8, MODE, 8, 4>(base, sel, n, stream))); break;
            case 4: PEA_TRY((launch_fat<16, MODE, 8, 4>(base, sel, n, stream))); break;
            default: PEA_TRY((launch_fat<16, MODE, 16, 4>(base, sel, n, stream))); break;
        }
        for (int q = 0; q < n; ++q) base.g[sel[q]].skip_long = 1;
    }
    return PEA_OK;
}

template <int MODE>
int launch_mode(const AggLaunch &base_in, hipStream_t stream) {
    AggLaunch base = base_in;
    PEA_TRY(launch_fat_groups<MODE>(base, stream));
    const int classes[3] = {0, 4, -1};  // -1 stands for "== G"
    for (int G = 4; G <= 64; G <<= 1) {
        for (int ci = 0; ci < 3; ++ci) {
            const int cls = classes[ci] < 0 ? G : classes[ci];
            if (ci == 1 && G == 4) continue;  // class 4 == class G there
            int sel[kMaxAggGroups], n = 0;
            for (int i = 0; i < base.n_groups; ++i)
                if (lanes_for(base.g[i].W) == G && f4_class(MODE, G, base.g[i].F) == cls) sel[n++] = i;
            if (!n) continue;
            switch (G) {
                case 4: PEA_TRY((launch_g<4, MODE>(base, cls, sel, n, stream))); break;
                case 8: PEA_TRY((launch_g<8, MODE>(base, cls, sel, n, stream))); break;
                case 16: PEA_TRY((launch_g<16, MODE>(base, cls, sel, n, stream))); break;
                case 32: PEA_TRY((launch_g<32, MODE>(base, cls, sel, n, stream))); break;
                default: PEA_TRY((launch_g<64, MODE>(base, cls, sel, n, stream))); break;
            }
        }
    }
    return PEA_OK;
}

}  // namespace

size_t partial_record_floats(int W, int F) { return (size_t)W + 2 * (size_t)(W / F); }

int launch_aggregate(AggMode mode, const AggGroup *groups, int n_groups, hipStream_t stream) {
    PEA_REQUIRE(n_groups >= 0 && n_groups <= kMaxAggGroups, PEA_ERR_ARG, "aggregate: %d groups (max %d)",
                n_groups, kMaxAggGroups);
    AggLaunch base;
    base.n_groups = n_groups;
    for (int i = 0; i < n_groups; ++i) {
        const AggGroup &g = groups[i];
        PEA_REQUIRE(g.W > 0 && g.W <= 256 && g.W % 4 == 0, PEA_ERR_ARG,
                    "aggregate: group width %d must be a multiple of 4 in (0, 256]", g.W);
        PEA_REQUIRE(g.F > 0 && g.F % 4 == 0 && g.W % g.F == 0, PEA_ERR_ARG,
                    "aggregate: head width %d must be a multiple of 4 dividing %d", g.F, g.W);
        PEA_REQUIRE(g.ld_feat % 4 == 0 && g.ld_out % 4 == 0, PEA_ERR_ARG,
                    "aggregate: row strides must be multiples of 4 floats");
        PEA_REQUIRE(g.n_hub == 0 || g.partial != nullptr, PEA_ERR_ARG, "aggregate: hub rows need a partial buffer");
        PEA_REQUIRE(mode != AGG_GAT || (g.neg_slope >= 0.f && g.neg_slope <= 1.f), PEA_ERR_ARG,
                    "aggregate: negative_slope %g outside [0, 1] is not supported", (double)g.neg_slope);
        base.g[i] = g;
    }
    switch (mode) {
        case AGG_GAT: return launch_mode<AGG_GAT>(base, stream);
        case AGG_GCN: return launch_mode<AGG_GCN>(base, stream);
        case AGG_WSUM: return launch_mode<AGG_WSUM>(base, stream);
        default: return launch_mode<AGG_MEAN>(base, stream);
    }
}

}  // namespace pea
