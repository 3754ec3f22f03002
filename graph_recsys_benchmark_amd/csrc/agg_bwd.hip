// Backward of the GAT aggregation for gfx950 (training; SURVEY.md section 8f rank 1: loss.backward() at reference
// solvers.py:215 runs autograd through GATConv's index_select / softmax / scatter).  Two gather passes, no atomics:
//
//   forward, per head:  z_ij = a_src_j + a_dst_i,  e_ij = leaky_relu(z_ij),  alpha_ij = exp(e_ij - m_i) / (S_i + 1e-16),
//                       out_i = sum_j alpha_ij T_j + b        (a_src_j = att_j . T_j,  a_dst_i = att_i . T_i)
//   given g_i = dL/d out_i:   c_i = g_i . (out_i - b)                      (= sum_j alpha_ij  g_i . T_j)
//                             d e_ij = alpha_ij (g_i . T_j - c_i),  d z_ij = d e_ij * leaky'(z_ij)
//   D pass (destination rows, forward CSR, gathers T_j):   d a_dst_i = sum_j d z_ij   (+ the side record
//                             [a_dst_i, m_i, 1/(S_i+eps), c_i] the S pass gathers)
//   S pass (source rows, REVERSED relation, gathers g_i):  d a_src_j = sum_i d z_ij
//                             dT_j = sum_i alpha_ij g_i + att_j d a_src_j + att_i d a_dst_j
// The softmax statistics (m_i in the log2 domain, S_i) were saved by the forward (AggGroup::stats); logits are
// recomputed from the rows exactly as in agg.hip (natural units; the weight is 2^(e log2(e) - m_i), agg_common.h: Soft).
// Same work decomposition as the forward: short rows per lane subgroup, long rows / hub chunks per wave, hub chunks
// folded in chunk order by a merge kernel -> bitwise reproducible gradients.
#include <algorithm>

#include "agg_common.h"

namespace pea {
namespace {

struct RowD {  // row-local state of the D pass
    float4 att_s, g, hself;
    float a_d, m, inv_s, c;
};

template <int F4T>
__device__ __forceinline__ RowD load_row_d(const AggGroup &P, int row, int c4, int lane, int pos, int F4, bool pow2) {
    RowD r;
    r.att_s = ld4(P.att_src + c4);
    r.hself = ld4(row_at(P.feat_self + c4, row, P.ld_self));
    r.a_d = head_sum<F4T>(dot4(r.hself, ld4(P.att_dst + c4)), lane, pos, F4, pow2);
    r.g = ld4(row_at(P.g_self + c4, row, P.ld_g));
    float4 o = ld4(row_at(P.o_self + c4, row, P.ld_g));
    if (P.bias) {
        const float4 b = ld4(P.bias + c4);
        o = make_float4(o.x - b.x, o.y - b.y, o.z - b.z, o.w - b.w);
    }
    r.c = head_sum<F4T>(dot4(r.g, o), lane, pos, F4, pow2);
    const float *sp = P.stats + (size_t)row * P.ld_stats + 2 * (c4 / P.F);
    r.m = sp[0];
    r.inv_s = 1.0f / (sp[1] + 1e-16f);
    return r;
}

// d z of one edge as seen from the destination row (h = gathered T_j)
template <int F4T>
__device__ __forceinline__ float dz_edge_d(const AggGroup &P, const RowD &r, float4 h, int lane, int pos, int F4, bool pow2) {
    const float zl = head_sum<F4T>(dot4(h, r.att_s), lane, pos, F4, pow2) + r.a_d;
    const float alpha = __builtin_amdgcn_exp2f(fmaf(leaky(zl, P.neg_slope), kLog2e, -r.m)) * r.inv_s;
    const float dal = head_sum<F4T>(dot4(h, r.g), lane, pos, F4, pow2);
    return alpha * (dal - r.c) * (zl > 0.f ? 1.f : P.neg_slope);
}

// is the gathered row i live?  (bitmap when the host built one, else the byte flags)
__device__ __forceinline__ bool row_live(const AggGroup &P, int i) {
    if (P.row_active_bits) return (P.row_active_bits[(unsigned)i >> 5] >> ((unsigned)i & 31u)) & 1u;
    return P.row_active[i] != 0;
}

__device__ __forceinline__ void finish_d(const AggGroup &P, const RowD &r, int row, int c4, float dsum) {
    if (c4 % P.F != 0) return;
    const int k = c4 / P.F;
    P.ksum[(size_t)row * P.ld_k + k] = dsum;
    st4(P.side_out + (size_t)row * P.ld_side + 4 * k, make_float4(r.a_d, r.m, r.inv_s, r.c));
}

struct RowS {  // row-local state of the S pass
    float4 att_s, t;
    float a_s;
};

template <int F4T>
__device__ __forceinline__ RowS load_row_s(const AggGroup &P, int row, int c4, int lane, int pos, int F4, bool pow2) {
    RowS r;
    r.att_s = ld4(P.att_src + c4);
    r.t = ld4(row_at(P.feat_self + c4, row, P.ld_self));
    r.a_s = head_sum<F4T>(dot4(r.t, r.att_s), lane, pos, F4, pow2);
    return r;
}

// one out-edge as seen from the source row: g = gathered output-gradient row of the destination, sd = its side record
template <int F4T>
__device__ __forceinline__ void edge_s(const AggGroup &P, const RowS &r, float4 g, float4 sd, bool ok, int lane, int pos,
                                       int F4, bool pow2, float4 &acc, float &dzs) {
    const float zl = r.a_s + sd.x;
    const float alpha = __builtin_amdgcn_exp2f(fmaf(leaky(zl, P.neg_slope), kLog2e, -sd.y)) * sd.z;
    const float dal = head_sum<F4T>(dot4(g, r.t), lane, pos, F4, pow2);
    const float dz = alpha * (dal - sd.w) * (zl > 0.f ? 1.f : P.neg_slope);
    if (ok) {
        acc = fma4(alpha, g, acc);
        dzs += dz;
    }
}

__device__ __forceinline__ void finish_s(const AggGroup &P, const RowS &r, int row, int c4, float4 acc, float dzs) {
    const int k = c4 / P.F;
    const float dad = P.da_dst[(size_t)row * P.ld_k + k];
    const float4 at_d = ld4(P.att_dst + c4);
    const float ws = dzs, wd = dad;   // natural-unit logits and attention vectors: d z / d T_j = att_j
    float4 o;
    o.x = acc.x + ws * r.att_s.x + wd * at_d.x;
    o.y = acc.y + ws * r.att_s.y + wd * at_d.y;
    o.z = acc.z + ws * r.att_s.z + wd * at_d.z;
    o.w = acc.w + ws * r.att_s.w + wd * at_d.w;
    st4(P.out + (size_t)row * P.ld_out + c4, o);
    if (c4 % P.F == 0) P.ksum[(size_t)row * P.ld_k + k] = dzs;
}

// ------------------------------------------------------------------------------------------------ short rows
// SP (here and in bwd_long_item): the launch has groups with row flags (the last layer of a training step).  The dense
// launches (SP = false: the first layer's x-space passes, 1.9 of the step's 5.6 ms) are compiled without the flag tests and
// the survivor queue: as run-time branches on a null pointer they cost the S pass 127 M scalar instructions per launch
// against 38 M of the D pass (SQ_INSTS_SALU, profiles/collect_sq_kind.sh gat train).
template <int G, int MODE, int F4T, bool SP>
__device__ __forceinline__ void bwd_short_rows(const AggGroup &P, const int blk) {
    const int item = blk * (kBlock / G) + (int)threadIdx.x / G;
    const int sl = (int)threadIdx.x % G, lane = (int)threadIdx.x % kWave;
    const bool valid = item < P.n_short;
    const int row = valid ? P.short_rows[item] : 0;
    const bool active = valid && sl * 4 < P.W;
    const int c4 = active ? sl * 4 : 0;
    // row_active (optional): rows whose output gradient is exactly zero (every row outside the BPR batch, for the last
    // layer) contribute nothing: the D pass writes d a_dst = 0 for them without gathering, the S pass skips their edges
    const bool row_on = !SP || !P.row_active || P.row_active[row] != 0;
    const int beg = P.rowptr[row];
    const int end = (valid && (MODE != AGG_GAT_BWD_D || row_on)) ? P.rowptr[row + 1] : beg;
    const int F4 = P.F / 4, pos = sl % F4;
    const bool pow2 = (F4 & (F4 - 1)) == 0;
    const int k = c4 / P.F;
    int len = end - beg;
    for (int off = G; off < kWave; off <<= 1) len = max(len, __shfl_xor(len, off));  // head sums are cross-lane
    if (MODE == AGG_SUM_BWD_S) {
        const float di = P.dinv_self[row];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int U = 4;
        for (int t = 0; t < len; t += U) {
            bool ok[U];
            int ii[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = beg + t + u < end;
                ii[u] = ok[u] ? P.col[beg + t + u] : 0;
                if (SP && P.row_active && ok[u] && !row_live(P, ii[u])) ok[u] = false, ii[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;                       // rows known to be zero are not fetched
                acc = fma4(P.dinv[ii[u]], ld4(row_at(P.feat + c4, ii[u], P.ld_feat)), acc);
            }
        }
        if (P.self_loop && row_on) acc = fma4(di, ld4(row_at(P.feat_self + c4, row, P.ld_self)), acc);
        if (active) st4(P.out + (size_t)row * P.ld_out + c4, scale4(acc, di));
        return;
    }
    if (MODE == AGG_GAT_BWD_D) {
        const RowD r = load_row_d<F4T>(P, row, c4, lane, pos, F4, pow2);
        float dsum = 0.f;
        constexpr int U = 4;  // edges in flight per subgroup (ids, then rows, then the arithmetic), as in agg_short_kernel
        for (int t = 0; t < len; t += U) {
            bool ok[U];
            float4 h[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = beg + t + u < end;
                const int j = ok[u] ? P.col[beg + t + u] : 0;
                h[u] = ld4(row_at(P.feat + c4, j, P.ld_feat));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float dz = dz_edge_d<F4T>(P, r, h[u], lane, pos, F4, pow2);
                if (ok[u]) dsum += dz;
            }
        }
        if (P.self_loop) {
            const float dz = dz_edge_d<F4T>(P, r, r.hself, lane, pos, F4, pow2);
            if (row_on) dsum += dz;
        }
        if (active) {
            if (row_on) finish_d(P, r, row, c4, dsum);
            else if (c4 % P.F == 0) P.ksum[(size_t)row * P.ld_k + k] = 0.f;
        }
    } else {
        const RowS r = load_row_s<F4T>(P, row, c4, lane, pos, F4, pow2);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float dzs = 0.f;
        constexpr int U = 4;
        for (int t = 0; t < len; t += U) {
            bool ok[U];
            float4 g[U], sd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = beg + t + u < end;
                int i = ok[u] ? P.col[beg + t + u] : 0;
                if (SP && P.row_active && ok[u] && !row_live(P, i)) ok[u] = false, i = 0;
                g[u] = ld4(row_at(P.feat + c4, i, P.ld_feat));
                sd[u] = ld4(row_at(P.side + 4 * k, i, P.ld_side));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) edge_s<F4T>(P, r, g[u], sd[u], ok[u], lane, pos, F4, pow2, acc, dzs);
        }
        if (P.self_loop) {
            const bool lone = P.deg0_self && P.deg0_self[row] != 0;   // wave-divergent per subgroup: both forms are lane-local
            const float4 gs = ld4(row_at(P.feat + c4, row, P.ld_feat));
            float4 sd = make_float4(0.f, 0.f, 1.f, 0.f);
            if (!lone) sd = ld4(row_at(P.side + 4 * k, row, P.ld_side));
            float4 acc2 = acc;
            float dz2 = dzs;
            edge_s<F4T>(P, r, gs, sd, row_on, lane, pos, F4, pow2, acc2, dz2);   // head sums are cross-lane: every lane runs it
            if (lone) {
                if (row_on) acc = add4(acc, gs);        // alpha = 1 exactly, d z = 0
            } else {
                acc = acc2;
                dzs = dz2;
            }
        }
        if (active) finish_s(P, r, row, c4, acc, dzs);
    }
}

// ------------------------------------------------------------------------------------------------ long rows / hub chunks
template <int G, int MODE, int F4T, bool SP>
__device__ __forceinline__ void bwd_long_item(const AggGroup &P, const int blk) {
    constexpr int NSG = kWave / G, U = 4;
    const int lane = (int)threadIdx.x % kWave;
    const int item = blk * (kBlock / kWave) + (int)threadIdx.x / kWave;
    if (item >= P.n_long) return;
    const LongItem it = P.long_items[item];
    if (it.slot == -2) return;
    const int sub = lane / G, sl = lane % G;
    const bool active = sl * 4 < P.W;
    const int c4 = active ? sl * 4 : 0;
    const int row = it.row;
    const int F4 = P.F / 4, pos = sl % F4;
    const bool pow2 = (F4 & (F4 - 1)) == 0;
    const int k = c4 / P.F, nk = P.W / P.F;
    const bool row_on = !SP || !P.row_active || P.row_active[row] != 0;  // see bwd_short_kernel
    if (MODE == AGG_GAT_BWD_D && !row_on) {
        if (it.slot < 0 && sub == 0 && active && c4 % P.F == 0) P.ksum[(size_t)row * P.ld_k + k] = 0.f;
        return;  // hub chunks of such a row leave their partial records alone: the merge kernel does not read them
    }
    constexpr bool kSrc = MODE == AGG_GAT_BWD_S || MODE == AGG_SUM_BWD_S;   // walks a source row's out-edges, filtered
    RowD rd;
    RowS rs;
    if (MODE == AGG_GAT_BWD_D) rd = load_row_d<F4T>(P, row, c4, lane, pos, F4, pow2);
    else if (MODE == AGG_GAT_BWD_S) rs = load_row_s<F4T>(P, row, c4, lane, pos, F4, pow2);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float dsum = 0.f;
    int src = it.beg + lane < it.end ? P.col[it.beg + lane] : -1;
    if (SP && kSrc && P.row_active && src >= 0 && !row_live(P, src)) src = -1;
    // Batch-sparse S pass (the last layer: 2.5 % of the gathered rows carry a gradient): the surviving edges of SEVERAL batches
    // of 64 are queued (edge order kept) and processed together.  Packing each batch on its own (round 2) still paid one
    // whole 4-edges-per-subgroup iteration for the 1-2 survivors of nearly every batch: 0.32 ms to find and process the
    // 0.6 M live edges among 24.8 M.
    __shared__ int live_q[kBlock / kWave][kWave];
    const int wq = (int)threadIdx.x / kWave;
    const bool queued = SP && kSrc && P.row_active != nullptr;
    int q_n = 0;
    for (int base = it.beg; base < it.end; base += kWave) {
        const int nxt = base + kWave + lane;
        int src_next = nxt < it.end ? P.col[nxt] : -1;
        if (SP && kSrc && P.row_active && src_next >= 0 && !row_live(P, src_next)) src_next = -1;
        int cnt = min(kWave, it.end - base);
        if (queued) {
            const unsigned long long live = __ballot(src >= 0);
            const int n_live = __popcll(live);
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0));
            const bool last_batch = base + kWave >= it.end;
            bool flush = false;
            if (q_n + n_live > kWave) {            // (wave-uniform) the queue cannot take this batch: process it first
                flush = true;
            }
            if (flush) {
                const int mine = lane < q_n ? live_q[wq][lane] : -1;
                // process the queue, then start a new one with this batch
                const int take = q_n;
                q_n = 0;
                if (src >= 0) live_q[wq][before] = src;
                q_n = n_live;
                src = mine;
                cnt = take;
            } else {
                if (src >= 0) live_q[wq][q_n + before] = src;
                q_n += n_live;
                if (!last_batch) {                  // keep collecting
                    src = src_next;
                    continue;
                }
                src = lane < q_n ? live_q[wq][lane] : -1;
                cnt = q_n;
                q_n = 0;
            }
        }
        for (int pass = 0; pass < 2; ++pass) {     // (queued, after a flush on the last batch: the new queue is processed too)
        for (int t = 0; t < cnt; t += NSG * U) {
            int jj[U];
            bool ok[U];
            float4 h[U], sd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = t + u * NSG + sub;
                const int j = __shfl(src, idx & (kWave - 1));
                ok[u] = idx < cnt && j >= 0;
                jj[u] = ok[u] ? j : 0;
            }
            if (MODE == AGG_SUM_BWD_S) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ok[u]) acc = fma4(P.dinv[jj[u]], ld4(row_at(P.feat + c4, jj[u], P.ld_feat)), acc);
                continue;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                h[u] = ld4(row_at(P.feat + c4, jj[u], P.ld_feat));
                if (MODE == AGG_GAT_BWD_S) sd[u] = ld4(row_at(P.side + 4 * k, jj[u], P.ld_side));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == AGG_GAT_BWD_D) {
                    const float dz = dz_edge_d<F4T>(P, rd, h[u], lane, pos, F4, pow2);
                    if (ok[u]) dsum += dz;
                } else {
                    edge_s<F4T>(P, rs, h[u], sd[u], ok[u], lane, pos, F4, pow2, acc, dsum);
                }
            }
        }
        if (!(queued && q_n > 0 && base + kWave >= it.end)) break;
        src = lane < q_n ? live_q[wq][lane] : -1;   // the last batch overflowed the queue: what it left behind
        cnt = q_n;
        q_n = 0;
        }
        src = src_next;
    }
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
        dsum += __shfl_xor(dsum, off);
        if (kSrc) acc = add4(acc, shfl_xor4(acc, off));
    }
    if (it.slot >= 0) {  // hub chunk: partial record [acc (W) | per head: unused, dsum]
        if (sub == 0 && active) {
            float *rec = P.partial + (size_t)it.slot * (size_t)(P.W + 2 * nk);
            if (kSrc) st4(rec + c4, acc);
            if (MODE != AGG_SUM_BWD_S && c4 % P.F == 0) rec[P.W + 2 * k + 1] = dsum;
        }
        return;
    }
    if (MODE == AGG_SUM_BWD_S) {
        const float di = P.dinv_self[row];
        if (P.self_loop && row_on) acc = fma4(di, ld4(row_at(P.feat_self + c4, row, P.ld_self)), acc);
        if (sub == 0 && active) st4(P.out + (size_t)row * P.ld_out + c4, scale4(acc, di));
        return;
    }
    if (P.self_loop) {
        if (MODE == AGG_GAT_BWD_D) {
            dsum += dz_edge_d<F4T>(P, rd, rd.hself, lane, pos, F4, pow2);
        } else if (P.deg0_self && P.deg0_self[row] != 0) {   // wave-uniform: one row per wave
            if (row_on) acc = add4(acc, ld4(row_at(P.feat + c4, row, P.ld_feat)));
        } else {
            edge_s<F4T>(P, rs, ld4(row_at(P.feat + c4, row, P.ld_feat)), ld4(row_at(P.side + 4 * k, row, P.ld_side)), row_on, lane,
                        pos, F4, pow2, acc, dsum);
        }
    }
    if (sub == 0 && active) {
        if (MODE == AGG_GAT_BWD_D) finish_d(P, rd, row, c4, dsum);
        else finish_s(P, rs, row, c4, acc, dsum);
    }
}

// One launch per pass and lane width, like the forward's agg_rows_kernel: workgroups [0, n_long_blocks) take the long rows
// and hub chunks, the rest the short rows (they fill the machine while the last long items drain).
template <int G, int MODE, int F4T, bool SP>
__global__ __launch_bounds__(kBlock) void bwd_rows_kernel(const AggLaunch L) {
    if ((int)blockIdx.x >= L.n_long_blocks) {
        const int b = (int)blockIdx.x - L.n_long_blocks;
        int gi = 0;
        while (gi + 1 < L.n_groups && b >= L.blk_short[gi + 1]) ++gi;
        bwd_short_rows<G, MODE, F4T, SP>(L.g[gi], b - L.blk_short[gi]);
        return;
    }
    const int gi = find_group(L);
    bwd_long_item<G, MODE, F4T, SP>(L.g[gi], (int)blockIdx.x - L.blk_start[gi]);
}

// ------------------------------------------------------------------------------------------------ hub rows
template <int G, int MODE, int F4T>
__global__ __launch_bounds__(kBlock) void bwd_merge_kernel(const AggLaunch L) {
    const int gi = find_group(L);
    const AggGroup &P = L.g[gi];
    const int lane = (int)threadIdx.x % kWave;
    const int item = ((int)blockIdx.x - L.blk_start[gi]) * (kBlock / kWave) + (int)threadIdx.x / kWave;
    if (item >= P.n_hub) return;
    const int row = P.hub_rows[item];
    const int first = P.hub_first[item], count = P.hub_count[item];
    const int sub = lane / G, sl = lane % G;
    const bool active = sl * 4 < P.W;
    const int c4 = active ? sl * 4 : 0;
    const int F4 = P.F / 4, pos = sl % F4;
    const bool pow2 = (F4 & (F4 - 1)) == 0;
    const int k = c4 / P.F, nk = P.W / P.F;
    const size_t rec_sz = (size_t)(P.W + 2 * nk);
    const bool row_on = !P.row_active || P.row_active[row] != 0;  // see bwd_short_kernel
    if (MODE == AGG_GAT_BWD_D && !row_on) {
        if (sub == 0 && active && c4 % P.F == 0) P.ksum[(size_t)row * P.ld_k + k] = 0.f;
        return;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float dsum = 0.f;
    // chunk order, every subgroup the same value; 8 records in flight per step (one dependent load per record made this
    // kernel 0.13 ms for a hub row of ~300 chunks): slots past the end re-read the last record and add zero, so the adds
    // stay unconditional and the compiler keeps the loads ahead of them
    constexpr int UM = 8;
    for (int c0 = 0; c0 < count; c0 += UM) {
        float4 a[UM];
        float dd[UM];
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            const int c = c0 + u < count ? c0 + u : count - 1;
            const float *rec = P.partial + (size_t)(first + c) * rec_sz;
            a[u] = MODE != AGG_GAT_BWD_D ? ld4(rec + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            dd[u] = MODE != AGG_SUM_BWD_S ? rec[P.W + 2 * k + 1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            const float f = c0 + u < count ? 1.f : 0.f;
            if (MODE != AGG_GAT_BWD_D) acc = fma4(f, a[u], acc);
            dsum = fmaf(f, dd[u], dsum);
        }
    }
    if (MODE == AGG_SUM_BWD_S) {
        const float di = P.dinv_self[row];
        if (P.self_loop && row_on) acc = fma4(di, ld4(row_at(P.feat_self + c4, row, P.ld_self)), acc);
        if (sub == 0 && active) st4(P.out + (size_t)row * P.ld_out + c4, scale4(acc, di));
        return;
    }
    if (MODE == AGG_GAT_BWD_D) {
        const RowD r = load_row_d<F4T>(P, row, c4, lane, pos, F4, pow2);
        if (P.self_loop) dsum += dz_edge_d<F4T>(P, r, r.hself, lane, pos, F4, pow2);
        if (sub == 0 && active) finish_d(P, r, row, c4, dsum);
    } else {
        const RowS r = load_row_s<F4T>(P, row, c4, lane, pos, F4, pow2);
        if (P.self_loop) {
            if (P.deg0_self && P.deg0_self[row] != 0) {   // wave-uniform (one hub row per wave): see bwd_short_kernel
                if (row_on) acc = add4(acc, ld4(row_at(P.feat + c4, row, P.ld_feat)));
            } else {
                edge_s<F4T>(P, r, ld4(row_at(P.feat + c4, row, P.ld_feat)), ld4(row_at(P.side + 4 * k, row, P.ld_side)), row_on,
                            lane, pos, F4, pow2, acc, dsum);
            }
        }
        if (sub == 0 && active) finish_s(P, r, row, c4, acc, dsum);
    }
}

template <int G, int MODE, int F4T>
int launch_bwd_g(const AggLaunch &base, const int *sel, int n_sel, hipStream_t stream) {
    // launches whose rows / gathered rows are filtered by row_active (the last layer: only the BPR batch's rows carry a
    // gradient) get their own name: the live messages are not known on the host, so no byte count is attributed to them
    bool sparse = false;
    for (int i = 0; i < n_sel; ++i) sparse = sparse || base.g[sel[i]].row_active != nullptr;
    // names as profiles/summarize.py derives them from rocprofv3's kernel names (bwd_rows_kernel<G, MODE, F4T>)
    static char names[2][32];
    if (!names[0][0]) {
        const char *fam = MODE == AGG_SUM_BWD_S ? "sum_bwd" : "gat_bwd";
        snprintf(names[0], sizeof(names[0]), "%s_%s_g%d", fam, MODE == AGG_GAT_BWD_D ? "dst" : "src", G);
        snprintf(names[1], sizeof(names[1]), "%s_%s_g%d_batch", fam, MODE == AGG_GAT_BWD_D ? "dst" : "src", G);
    }
    const char *nm = names[sparse ? 1 : 0];
    AggLaunch L;
    {   // long rows + hub chunks first, short rows behind them, one launch
        L.n_groups = 0;
        int blocks = 0, sblocks = 0;
        // bytes the launch pulls through the memory system, each once per message (bench.py's live roofline): the gathered
        // row chunk (D: T_j; S: the output-gradient row g_i) + the 4-byte index (+ S: the 16-byte side record per head)
        double pulled = 0.0, table = 0.0;
        for (int i = 0; i < n_sel; ++i) {
            const AggGroup &g = base.g[sel[i]];
            if (g.n_short <= 0 && g.n_long <= 0) continue;
            if (!sparse) {
                const double per = 4.0 * g.W + 4.0 + (MODE == AGG_GAT_BWD_S ? 16.0 * (g.W / g.F) : 0.0);
                pulled += per * (g.msgs_short + g.msgs_long);
                table = std::max(table, g.table_rows * 4.0 * g.W);
            }
            L.blk_start[L.n_groups] = blocks;
            L.blk_short[L.n_groups] = sblocks;
            L.g[L.n_groups++] = g;
            if (g.n_long > 0) blocks += ((g.n_long + 3) / 4 + 7) / 8 * 8;
            if (g.n_short > 0) sblocks += (g.n_short + (kBlock / G) - 1) / (kBlock / G);
        }
        L.blk_start[L.n_groups] = blocks;
        L.blk_short[L.n_groups] = sblocks;
        L.n_long_blocks = blocks;
        if (blocks + sblocks > 0) {
            ProfScope ps(nm, stream, pulled, pulled, table);
            if (sparse) PEA_LAUNCH((bwd_rows_kernel<G, MODE, F4T, true>), dim3(blocks + sblocks), dim3(kBlock), 0, stream, L);
            else PEA_LAUNCH((bwd_rows_kernel<G, MODE, F4T, false>), dim3(blocks + sblocks), dim3(kBlock), 0, stream, L);
            PEA_HIP(hipGetLastError());
        }
    }
    {   // hub merge
        L.n_groups = 0;
        int blocks = 0;
        for (int i = 0; i < n_sel; ++i) {
            const AggGroup &g = base.g[sel[i]];
            if (g.n_hub <= 0) continue;
            L.blk_start[L.n_groups] = blocks;
            L.g[L.n_groups++] = g;
            blocks += (g.n_hub + 3) / 4;
        }
        L.blk_start[L.n_groups] = blocks;
        if (blocks > 0) {
            ProfScope ps(MODE == AGG_GAT_BWD_D ? "gat_bwd_dst_merge" : MODE == AGG_SUM_BWD_S ? "sum_bwd_src_merge" : "gat_bwd_src_merge",
                         stream, 0.0);
            PEA_LAUNCH((bwd_merge_kernel<G, MODE, F4T>), dim3(blocks), dim3(kBlock), 0, stream, L);
            PEA_HIP(hipGetLastError());
        }
    }
    return PEA_OK;
}

template <int G, int MODE>
int launch_bwd_cls(const AggLaunch &base, int cls, const int *sel, int n, hipStream_t stream) {
    if (cls == G) return launch_bwd_g<G, MODE, G>(base, sel, n, stream);
    if (cls == 4) return launch_bwd_g<G, MODE, (G >= 4 ? 4 : 0)>(base, sel, n, stream);
    return launch_bwd_g<G, MODE, 0>(base, sel, n, stream);
}

template <int MODE>
int launch_bwd_mode(const AggLaunch &base, hipStream_t stream) {
    const int classes[3] = {0, 4, -1};
    for (int G = 4; G <= 64; G <<= 1)
        for (int ci = 0; ci < 3; ++ci) {
            const int cls = classes[ci] < 0 ? G : classes[ci];
            if (ci == 1 && G == 4) continue;
            int sel[kMaxAggGroups], n = 0;
            for (int i = 0; i < base.n_groups; ++i) {
                const int f4 = base.g[i].F / 4;
                const int c = f4 == G ? G : f4 == 4 ? 4 : 0;
                if (lanes_for(base.g[i].W) == G && c == cls) sel[n++] = i;
            }
            if (!n) continue;
            switch (G) {
                case 4: PEA_TRY((launch_bwd_cls<4, MODE>(base, cls, sel, n, stream))); break;
                case 8: PEA_TRY((launch_bwd_cls<8, MODE>(base, cls, sel, n, stream))); break;
                case 16: PEA_TRY((launch_bwd_cls<16, MODE>(base, cls, sel, n, stream))); break;
                case 32: PEA_TRY((launch_bwd_cls<32, MODE>(base, cls, sel, n, stream))); break;
                default: PEA_TRY((launch_bwd_cls<64, MODE>(base, cls, sel, n, stream))); break;
            }
        }
    return PEA_OK;
}

// out_k[c] = sum_n A[n, c] * (S_k ? S_k[n, c / F] : 1) for c < W (k = 0 and, optionally, 1: two scaled sums over ONE read
// of A: the att_j and att_i gradients of a GAT level both weight T_s), rows in a fixed order (two stages, no atomics).
// Stage 1: block b owns a contiguous row chunk; each of its 4 waves streams WHOLE rows -- lane l reads the float4
// chunks l, l + 64, l + 128, l + 192 of a row, 4 rows in flight -- and the 4 waves' sums are folded through LDS in wave
// order.  (Round 1 read 64-column pieces, one dword per lane: nine strided passes over a 2304-byte-row table at
// 2.7 TB/s.)  MAPPED = false: rows 0..N-1 in place (one GPU); true: the rows a rank owns (RowMap).
constexpr int kColsumT = 4;   // float4 chunks per lane: column blocks of up to 1024 columns
// T = chunks per lane (W > 256: every lane of a wave works on ONE row); T == 1 and W <= 256: a row needs only
// C = pow2ceil(W / 4) lanes, so a wave streams 64 / C rows side by side (the sub-rows are folded through LDS with the waves)
template <bool MAPPED, bool TWO, int T>
__global__ __launch_bounds__(256) void colsum_stage1(const RowMap M, int W, int F, int C, const float *__restrict__ A, int lda,
                                                     const float *__restrict__ S0, const float *__restrict__ S1, int lds,
                                                     float *__restrict__ part, size_t part_stride) {
    extern __shared__ float red[];  // [4 waves][NK][64 lanes * T] float4 slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = 64 / C;                    // rows per wave step (1 when T > 1)
    const int rsub = lane / C, cl = lane % C;
    const int64_t n_all = M.size();
    const int64_t rows_per = (n_all + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = min(n_all, r0 + rows_per);
    constexpr int NK = TWO ? 2 : 1;
    for (int cb = 0; cb < W; cb += 256 * T) {
        const int wb = min(W - cb, 256 * T);
        float4 acc[NK][T];
        bool on[T];
        int hk[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int c = cb + 4 * (cl + 64 * t);
            on[t] = c < cb + wb;
            hk[t] = on[t] ? c / F : 0;
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[k][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        constexpr int U = 4;
        for (int64_t q = r0 + (int64_t)wave * R + rsub; q < r1; q += (int64_t)4 * R * U) {
            float4 v[U][T];
            float s0[U][T], s1[U][T];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t qq = q + (int64_t)4 * R * u;
                int64_t rr = MAPPED ? (qq < r1 ? M.row(qq) : 0) : qq;
                const bool rv = qq < r1 && (!MAPPED || rr < M.N);
                if (!rv) rr = 0;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const bool ok = rv && on[t];
                    v[u][t] = ok ? ld4(A + rr * lda + cb + 4 * (cl + 64 * t)) : make_float4(0.f, 0.f, 0.f, 0.f);
                    s0[u][t] = (ok && S0) ? S0[rr * lds + hk[t]] : 1.f;
                    if (TWO) s1[u][t] = ok ? S1[rr * lds + hk[t]] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)     // row order inside a lane: fixed
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    acc[0][t] = fma4(s0[u][t], v[u][t], acc[0][t]);
                    if (TWO) acc[1][t] = fma4(s1[u][t], v[u][t], acc[1][t]);
                }
        }
        // every (wave, sub-row) stream parks its sums; lanes of wave 0 / sub-row 0 fold them in (wave, sub-row) order
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int t = 0; t < T; ++t)
                st4(red + (((size_t)wave * NK + k) * T + t) * 256 + 4 * lane, acc[k][t]);
        __syncthreads();
        if (wave == 0 && rsub == 0) {
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if (!on[t]) continue;
                    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int w = 0; w < 4; ++w)
                        for (int rs = 0; rs < R; ++rs)
                            r = add4(r, ld4(red + (((size_t)w * NK + k) * T + t) * 256 + 4 * (rs * C + cl)));
                    st4(part + (size_t)k * part_stride + (size_t)blockIdx.x * W + cb + 4 * (cl + 64 * t), r);
                }
        }
        __syncthreads();
    }
}

// 16 columns per block; thread (g, cc) adds parts g, g + 16, ... of its column in order, then one thread per column adds
// the 16 sub-sums in order: fixed summation order, 32 dependent loads per thread instead of 512
__global__ __launch_bounds__(256) void colsum_stage2(int nparts, int W, const float *__restrict__ part, float scale,
                                                     float *__restrict__ out) {
    __shared__ float sub[16][17];
    const int g = threadIdx.x >> 4, cc = threadIdx.x & 15, c = blockIdx.x * 16 + cc;
    float s = 0.f;
    if (c < W)
        for (int p = g; p < nparts; p += 16) s += part[(size_t)p * W + c];
    sub[g][cc] = s;
    __syncthreads();
    if (g == 0 && c < W) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sub[q][cc];
        out[c] = t * scale;
    }
}

// in place: G[n, c] = O[n, c] > 0 ? G[n, c] : 0   (relu between steps, reference models/base.py:138)
__global__ __launch_bounds__(256) void relu_mask_kernel(const RowMap M, int W4, float *__restrict__ G, int ldg,
                                                        const float *__restrict__ O, int ldo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M.n * W4) return;
    int64_t q;
    int c;
    if (M.n * W4 < (int64_t)1 << 31) {   // 32-bit division (the usual case): the 64-bit one costs ~40 instructions per thread
        const unsigned qi = (unsigned)idx / (unsigned)W4;
        q = qi;
        c = (int)((unsigned)idx - qi * (unsigned)W4) * 4;
    } else {
        q = idx / W4;
        c = (int)(idx % W4) * 4;
    }
    const int64_t r = M.row(q);
    if (r >= M.N) return;
    float4 g = ld4(G + r * ldg + c);
    const float4 o = ld4(O + r * ldo + c);
    g.x = o.x > 0.f ? g.x : 0.f;
    g.y = o.y > 0.f ? g.y : 0.f;
    g.z = o.z > 0.f ? g.z : 0.f;
    g.w = o.w > 0.f ? g.w : 0.f;
    st4(G + r * ldg + c, g);
}

}  // namespace

int launch_gat_backward(AggMode mode, const AggGroup *groups, int n_groups, hipStream_t stream) {
    PEA_REQUIRE(n_groups >= 0 && n_groups <= kMaxAggGroups, PEA_ERR_ARG, "gat backward: %d groups", n_groups);
    AggLaunch base;
    base.n_groups = n_groups;
    for (int i = 0; i < n_groups; ++i) base.g[i] = groups[i];
    if (mode == AGG_SUM_BWD_S) return launch_bwd_mode<AGG_SUM_BWD_S>(base, stream);
    return mode == AGG_GAT_BWD_D ? launch_bwd_mode<AGG_GAT_BWD_D>(base, stream) : launch_bwd_mode<AGG_GAT_BWD_S>(base, stream);
}

template <bool MP, bool TW>
static void colsum_dispatch(int T, size_t lds_bytes, hipStream_t stream, const RowMap &rows, int W, int F, int C, const float *A,
                            int lda, const float *S0, const float *S1, int lds, float *part, size_t part_stride) {
#define PEA_COLSUM(TT)                                                                                                        \
    PEA_LAUNCH((colsum_stage1<MP, TW, TT>), dim3(kColsumParts), dim3(256), lds_bytes, stream, rows, W, F, C, A, lda, S0, \
                       S1, lds, part, part_stride)
    if (T == 1) PEA_COLSUM(1);
    else if (T == 2) PEA_COLSUM(2);
    else if (T == 3) PEA_COLSUM(3);
    else PEA_COLSUM(4);
#undef PEA_COLSUM
}

int launch_colsum2(const RowMap &rows, int W, int F, const float *A, int lda, const float *S0, const float *S1, int lds,
                   float scale, float *part, float *out0, float *out1, hipStream_t stream) {
    if (W <= 0) return PEA_OK;
    PEA_REQUIRE(W % 4 == 0 && lda % 4 == 0 && F % 4 == 0, PEA_ERR_ARG, "colsum: widths / strides must be multiples of 4");
    ProfScope ps("colsum", stream, 0.0);
    const bool two = S1 != nullptr;
    const size_t part_stride = (size_t)kColsumParts * W;   // second output's partials follow the first's
    const int w4 = W / 4;
    const int T = std::min(kColsumT, (w4 + 63) / 64);
    int C = 64;
    if (T == 1) {
        C = 1;
        while (C < w4) C <<= 1;
    }
    const size_t lds_bytes = (size_t)4 * (two ? 2 : 1) * T * 256 * sizeof(float);
    const bool mapped = rows.world > 1 || rows.list != nullptr;
    if (mapped && two) colsum_dispatch<true, true>(T, lds_bytes, stream, rows, W, F, C, A, lda, S0, S1, lds, part, part_stride);
    else if (mapped) colsum_dispatch<true, false>(T, lds_bytes, stream, rows, W, F, C, A, lda, S0, S1, lds, part, part_stride);
    else if (two) colsum_dispatch<false, true>(T, lds_bytes, stream, rows, W, F, C, A, lda, S0, S1, lds, part, part_stride);
    else colsum_dispatch<false, false>(T, lds_bytes, stream, rows, W, F, C, A, lda, S0, S1, lds, part, part_stride);
    PEA_LAUNCH(colsum_stage2, dim3((unsigned)((W + 15) / 16)), dim3(256), 0, stream, kColsumParts, W, part, scale, out0);
    if (two)
        PEA_LAUNCH(colsum_stage2, dim3((unsigned)((W + 15) / 16)), dim3(256), 0, stream, kColsumParts, W,
                           part + part_stride, scale, out1);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

int launch_colsum(const RowMap &rows, int W, int F, const float *A, int lda, const float *S, int lds, float scale, float *part,
                  float *out, hipStream_t stream) {
    return launch_colsum2(rows, W, F, A, lda, S, nullptr, lds, scale, part, out, nullptr, stream);
}

int launch_relu_mask(const RowMap &rows, int W, float *G, int ldg, const float *O, int ldo, hipStream_t stream) {
    if (W <= 0 || rows.n <= 0) return PEA_OK;
    ProfScope ps("relu_mask", stream, 0.0);
    const int64_t total = rows.n * (W / 4);
    PEA_LAUNCH(relu_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, rows, W / 4, G, ldg, O, ldo);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

}  // namespace pea
