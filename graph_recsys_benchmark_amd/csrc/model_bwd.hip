// Backward of one level of the PEA schedule (training; reference solvers.py:215 loss.backward()).
// The sparse half -- gradient gathers over the reversed relations, relu masks, bias / attention-vector gradient
// reductions -- runs here; the dense half (dW = In^T dT, dIn = dT W: plain GEMMs) is done by the host mirror with
// rocBLAS through torch.mm on views of the same workspace (graph_recsys_benchmark_amd/autograd.py).
//
// Workspace regions (model.h): dX [N, ld_x] gradient of the last-layer outputs (internal column order), per level
// dO_s [N, ld_o] gradient of the relu(conv) outputs, dT_s [N, ld_t] gradient of the transformed features (SAGE: of the
// neighbour means), side_s / dad_s / das_s GAT per-head records, gpack: bias / att gradients at the SAME offsets as
// their values in the weight pack.
#include "model.h"

namespace pea {
namespace {

__global__ void fill_kernel(int64_t n, float v, float *p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// bits[w] = bit b set where flags[32 w + b] != 0
__global__ void flags_to_bits_kernel(int64_t N, const unsigned char *__restrict__ flags, unsigned *__restrict__ bits) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = n < N && flags[n] != 0;
    const unsigned long long b = __ballot(on);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && n < N) bits[n >> 5] = (unsigned)b;
    if (lane == 32 && n < N) bits[n >> 5] = (unsigned)(b >> 32);
}

__global__ void invdeg_kernel(int64_t N, const int *__restrict__ rowptr, float *__restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int d = rowptr[v + 1] - rowptr[v];
    out[v] = 1.0f / (float)(d < 1 ? 1 : d);
}

int ensure_sage_arrays(pea_plan *plan, int rel, hipStream_t stream) {
    const int64_t N = plan->N;
    if (!plan->ones) {
        PEA_HIP(hipMalloc((void **)&plan->ones, (size_t)N * sizeof(float)));
        PEA_LAUNCH(fill_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, 1.0f, plan->ones);
    }
    Relation &R = plan->rels[(size_t)rel];
    if (!R.invdeg) {
        PEA_HIP(hipMalloc((void **)&R.invdeg, (size_t)N * sizeof(float)));
        PEA_LAUNCH(invdeg_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, R.rowptr, R.invdeg);
    }
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}

void fill_lists(AggGroup &a, const Relation &R) {
    a.rowptr = R.rowptr;
    a.col = R.col;
    a.short_rows = R.short_rows;
    a.long_items = R.long_items;
    a.hub_rows = R.hub_rows;
    a.hub_first = R.hub_first;
    a.hub_count = R.hub_count;
    a.n_short = R.n_short;
    a.n_long = R.n_long;
    a.n_hub = R.n_hub;
}

}  // namespace
}  // namespace pea

using namespace pea;

// Level 0 of the two-step training schedule: the rows whose input gradient of the first transform is not identically zero
// (pea_rows_nonzero of dT_1): the softmax passes skip the others (their dA_0 rows are never written), the bias gradient walks
// the list.  NULL flags: every row.
extern "C" int pea_model_set_active_rows0(pea_model *m, const unsigned char *flags, const int32_t *list, const int32_t *count_dev) {
    PEA_REQUIRE(m, PEA_ERR_ARG, "set_active_rows0: null model");
    PEA_REQUIRE((list == nullptr) == (count_dev == nullptr) && (flags == nullptr || list != nullptr), PEA_ERR_ARG,
                "set_active_rows0: a list comes with its count, flags with the list");
    m->active0 = flags;              // NULL with a list: the gathers run over every row (dA_0 is zero outside the list)
    m->active0_list = list;
    m->active0_count = count_dev;
    return PEA_OK;
}

extern "C" int pea_model_set_active_rows(pea_model *m, const unsigned char *row_active) {
    PEA_REQUIRE(m, PEA_ERR_ARG, "set_active_rows: null model");
    m->active_rows = row_active;
    return PEA_OK;
}

// phase 0: relu masks + bias gradients (all kinds); GAT/GCN: the aggregation backward -> dT_s (+ att gradients)
// phase 1: SAGE only: reverse mean aggregation of dM_s (written by the host into the dT_s region) -> side_s region
// Sharded plans (one rank's rows): the gradient gathers over the REVERSED relation read output-gradient rows (and GAT
// side records) of destination nodes other ranks own, so GAT/GCN phase 0 stops before them -- relu masks, GAT D pass,
// bias / att_i reductions over own rows -- the host fills those rows in from their owners (sharding.fill_in_rows on
// the node-indexed dX / dO_s / side_s buffers), and
// phase 2 runs the rest: GAT S pass / GCN reverse aggregation -> dT_s on own rows, att_j reduction.  (SAGE: the host
// fills in dM_s rows between phase 0 and phase 1.)  Every row-wise reduction walks the rank's own rows (RowMap).
extern "C" int pea_model_backward_level(pea_model *m, int level, int phase, void *workspace, size_t workspace_bytes,
                                        void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(m && workspace && m->backward, PEA_ERR_ARG, "backward: model without enable_backward");
    const bool premasked = (phase & PEA_BWD_PREMASKED) != 0;   // the producer of this level's output gradients applied the relu mask
    phase &= ~PEA_BWD_PREMASKED;
    PEA_REQUIRE(level >= 0 && level < (int)m->levels.size() && phase >= 0 && phase <= 2, PEA_ERR_ARG, "backward: level %d phase %d", level, phase);
    PEA_REQUIRE(workspace_bytes >= pea_model_workspace_bytes(m), PEA_ERR_NOMEM, "backward: workspace too small");
    float *wsf = aligned_ws(workspace);
    pea_plan *plan = const_cast<pea_plan *>(m->plan);
    const pea_model_desc &d = m->d;
    const int64_t N = plan->N;
    Level &L = m->levels[(size_t)level];
    float *pack = wsf, *gpack = wsf + m->off_gpack, *colsum_part = wsf + m->off_colsum;
    float *T = wsf + L.off_t, *O = wsf + L.off_o, *X = wsf + m->off_x;
    float *dT = wsf + L.off_dt, *dO = wsf + L.off_do, *dX = wsf + m->off_dx;
    float *partial = wsf + m->off_partial;
    const bool loops = plan->flags & PEA_PLAN_SELF_LOOPS;
    const bool sharded = plan->shard_world > 1;
    const RowMap own = make_rowmap(N, plan->shard_tile, plan->shard_world, plan->shard_rank);

    if (d.kind == PEA_KIND_SAGE && m->fused2_train) {
        // Two-step training schedule, SAGE (forward: model.hip run_fused2_stage0 / mlp2_sage_kernel).  Layer 2 is transform
        // first -- out = mean_j T_1[j] + R_1[i], T_1 = H lin_rel1^T, R_1 = H lin_root1^T + bias1 -- so its backward is a
        // 16-wide gather:   level 1:  d bias1 = colsum dX;  dT_1[j] = sum_{i: j -> i} dX[i] / deg_i  (reversed relation).
        // Level 0 (x space; the host mirror has run csrc/mlp2_bwd.hip: dZ_0 in dO_0, dM_0 = dZ_0 lin_rel0 per channel in the
        // dT_0 region, the root term's gradient dZ_0 lin_root0 per channel in the side region):  d bias0 = colsum dZ_0 over
        // the live rows;  per channel  dXp_j = sum_{i: j -> i} dM_0[i] / deg_i + (dZ_0 lin_root0)[j]  written over A_0 (dead
        // once the weight gradients have read the means); the host sums the channel blocks into dx.
        PEA_REQUIRE(phase == 0 && !sharded && level <= 1, PEA_ERR_ARG, "backward: the two-step training schedule is single-GPU, phase 0");
        std::vector<AggGroup> gs;
        if (level == 1) {
            PEA_TRY(launch_colsum(own, L.n_cols, L.n_cols, dX, m->ld_x, nullptr, 0, 1.0f, colsum_part, gpack + L.bias_off, stream));
            const unsigned *active_bits = nullptr;
            if (m->active_rows) {   // only the batch's rows of dX are non-zero: the others are not fetched (csrc/agg.hip)
                if (!m->active_bits) PEA_HIP(hipMalloc((void **)&m->active_bits, (size_t)((N + 63) / 64) * 2 * sizeof(unsigned)));
                PEA_LAUNCH(flags_to_bits_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, m->active_rows, m->active_bits);
                PEA_HIP(hipGetLastError());
                active_bits = m->active_bits;
            }
            for (const GroupPlan &g : L.groups) {
                const int rr = m->reverse_of[(size_t)g.rel];
                PEA_REQUIRE(rr >= 0, PEA_ERR_ARG, "backward: relation %d has no reversed relation in the plan", g.rel);
                PEA_REQUIRE(g.last && g.out_col == g.col, PEA_ERR_ARG, "backward: two-step SAGE expects X in the T_1 column order");
                Relation &R = plan->rels[(size_t)g.rel], &Rr = plan->rels[(size_t)rr];
                PEA_REQUIRE(g.partial_off + (size_t)std::max(R.n_slots, Rr.n_slots) * partial_record_floats(g.W, g.F) <= m->partial_floats,
                            PEA_ERR_NOMEM, "backward: hub partial buffer too small for relation %d and its reverse", g.rel);
                PEA_TRY(ensure_sage_arrays(plan, g.rel, stream));
                AggGroup a{};
                fill_lists(a, Rr);
                a.W = g.W;
                a.F = g.W;
                a.feat = dX + g.out_col;
                a.ld_feat = m->ld_x;
                a.feat_self = a.feat;
                a.ld_self = m->ld_x;
                a.dinv = R.invdeg;           // 1 / max(deg_i, 1) of the gathered (forward destination) node
                a.dinv_self = plan->ones;
                a.row_active = m->active_rows;
                a.row_active_bits = active_bits;
                a.out = dT + g.col;
                a.ld_out = L.ld_t;
                a.partial = partial + g.partial_off;
                a.msgs_short = (double)Rr.edges_short;
                a.msgs_long = (double)Rr.edges_long;
                a.idx_share = 1.0;
                a.table_rows = (double)Rr.src_span;
                gs.push_back(a);
            }
        } else {
            const int E0 = d.emb_dim;
            const RowMap live = m->active0_list ? make_rowmap_list(N, m->active0_list, m->active0_count, N) : own;
            PEA_TRY(launch_colsum(live, L.n_cols, L.n_cols, dO, L.ld_o, nullptr, 0, 1.0f, colsum_part, gpack + L.bias_off, stream));
            size_t part_off = 0;
            for (size_t ui = 0; ui < L.units.size(); ++ui) {
                const Unit &u = L.units[ui];
                const int rr = m->reverse_of[(size_t)u.rel];
                PEA_REQUIRE(rr >= 0, PEA_ERR_ARG, "backward: relation %d has no reversed relation in the plan", u.rel);
                Relation &R = plan->rels[(size_t)u.rel], &Rr = plan->rels[(size_t)rr];
                const size_t rec = partial_record_floats(E0, E0);
                PEA_REQUIRE(part_off + (size_t)std::max(R.n_slots, Rr.n_slots) * rec <= m->partial_floats, PEA_ERR_NOMEM,
                            "backward: hub partial buffer too small for relation %d and its reverse", u.rel);
                PEA_TRY(ensure_sage_arrays(plan, u.rel, stream));
                AggGroup a{};
                fill_lists(a, Rr);
                a.W = E0;
                a.F = E0;
                a.self_loop = 1;             // the root term's gradient rides as the row's own term (weight 1)
                a.partial = partial + part_off;
                part_off += (size_t)std::max(R.n_slots, Rr.n_slots) * rec;
                a.feat = dT + (size_t)ui * E0;
                a.ld_feat = L.ld_t;
                a.feat_self = wsf + L.off_side + (size_t)ui * E0;
                a.ld_self = L.ld_t;
                a.dinv = R.invdeg;
                a.dinv_self = plan->ones;
                a.out = T + (size_t)ui * E0;
                a.ld_out = L.ld_t;
                a.msgs_short = (double)Rr.edges_short;
                a.msgs_long = (double)Rr.edges_long;
                a.idx_share = 1.0;
                a.table_rows = (double)Rr.src_span;
                gs.push_back(a);
            }
        }
        // level 1 with the batch's row flags: the batch-sparse walk of the backward kernels (only flagged rows are fetched)
        const bool batch_walk = level == 1 && m->active_rows != nullptr;
        for (size_t b = 0; b < gs.size(); b += kMaxAggGroups) {
            const int nb = (int)std::min<size_t>(kMaxAggGroups, gs.size() - b);
            if (batch_walk) PEA_TRY(launch_gat_backward(AGG_SUM_BWD_S, gs.data() + b, nb, stream));
            else PEA_TRY(launch_aggregate(AGG_GCN, gs.data() + b, nb, stream));
        }
        return PEA_OK;
    }
    if (d.kind == PEA_KIND_SAGE) {
        PEA_REQUIRE(phase <= 1, PEA_ERR_ARG, "backward: SAGE levels have phases 0 and 1");
        if (phase == 0) {
            for (const Unit &u : L.units) {
                float *G = u.last ? dX + u.o_col : dO + u.o_col;
                const float *Out = u.last ? X + u.o_col : O + u.o_col;
                const int ldg = u.last ? m->ld_x : L.ld_o;
                if (!u.last) PEA_TRY(launch_relu_mask(own, u.HF, G, ldg, Out, ldg, stream));
                PEA_TRY(launch_colsum(own, u.HF, u.HF, G, ldg, nullptr, 0, 1.0f, colsum_part, gpack + u.bias_off, stream));
            }
            return PEA_OK;
        }
        std::vector<AggGroup> gs;
        for (const GroupPlan &g : L.groups) {
            const int rr = m->reverse_of[(size_t)g.rel];
            PEA_REQUIRE(rr >= 0, PEA_ERR_ARG, "backward: relation %d has no reversed relation in the plan", g.rel);
            PEA_TRY(ensure_sage_arrays(plan, g.rel, stream));
            AggGroup a{};
            fill_lists(a, plan->rels[(size_t)rr]);
            a.W = g.W;
            a.F = g.W;
            a.feat = dT + g.out_col;  // dM columns of this group
            a.ld_feat = L.ld_t;
            a.feat_self = a.feat;
            a.ld_self = L.ld_t;
            a.dinv = plan->rels[(size_t)g.rel].invdeg;  // 1 / max(deg_i, 1) of the gathered (forward destination) node
            a.dinv_self = plan->ones;
            a.out = wsf + L.off_side + g.out_col;       // gradient wrt the gathered input columns, M_s column layout
            a.ld_out = L.ld_t;
            a.partial = partial + g.partial_off;
            gs.push_back(a);
        }
        for (size_t b = 0; b < gs.size(); b += kMaxAggGroups)
            PEA_TRY(launch_aggregate(AGG_GCN, gs.data() + b, (int)std::min<size_t>(kMaxAggGroups, gs.size() - b), stream));
        return PEA_OK;
    }

    if (m->fused2_train && level == 0) {
        // Two-step training schedule, first layer, in x space.  Inputs (host mirror: autograd.py): dO_0 = dZ_0, the gradient of
        // the first transform's pre-activations (relu mask applied by the gated product that wrote it); dA_0 = dZ_0 W_0 in
        // the dT_0 region; A_0 (the aggregates of the rows with incoming edges) in the T_0 region; the forward's softmax statistics.  Here: the bias
        // gradient, the D pass (destination rows, gathers x rows, c_i = dA_i . A_i), the S pass (source rows over the reversed
        // relation, gathers dA_i and the side records; per channel  dXp_j = sum_i alpha_ij dA_i + ws d a_src_j + wd d a_dst_j
        // with ws / wd = W_0^T att_j / att_i) -- written over A_0, which is dead once the D pass has read it.  The host then
        // sums the channel blocks into dx and reduces d ws / d wd (d a_src / d a_dst against x).
        PEA_REQUIRE(phase == 0 && !sharded, PEA_ERR_ARG, "backward: the two-step training schedule is single-GPU, phase 0");
        PEA_REQUIRE(m->last_x != nullptr, PEA_ERR_ARG, "backward: no training forward ran on this model");
        const float *xin = m->last_x;
        const int ldx = (int)m->last_ldx;
        const int E0 = d.emb_dim;
        float *A0 = T, *dA0 = dT;
        PEA_MEMSET_ASYNC(wsf + L.off_dad, 0, (size_t)N * (size_t)L.ld_k * sizeof(float), stream);
        const RowMap live = m->active0_list ? make_rowmap_list(N, m->active0_list, m->active0_count, N) : own;
        PEA_TRY(launch_colsum(live, L.n_cols, L.n_cols, dO, L.ld_o, nullptr, 0, 1.0f, colsum_part, gpack + L.bias_off, stream));
        std::vector<AggGroup> gd, gs;
        size_t part_off = 0;
        for (size_t ui = 0; ui < L.units.size(); ++ui) {
            const Unit &u = L.units[ui];
            const int rr = m->reverse_of[(size_t)u.rel];
            PEA_REQUIRE(rr >= 0, PEA_ERR_ARG, "backward: relation %d has no reversed relation in the plan", u.rel);
            Relation &R = plan->rels[(size_t)u.rel], &Rr = plan->rels[(size_t)rr];
            const size_t rec = partial_record_floats(E0, E0);
            PEA_REQUIRE(part_off + (size_t)std::max(R.n_slots, Rr.n_slots) * rec <= m->partial_floats, PEA_ERR_NOMEM,
                        "backward: hub partial buffer too small for relation %d and its reverse", u.rel);
            if (d.kind == PEA_KIND_GCN) {
                // GCN: the aggregation is linear, its backward is the same weighted sum over the REVERSED relation, in x space:
                // dXp_j = sum_i dinv_j dinv_i dA_i + dinv_j^2 dA_j (norm of the forward relation: GCNConv)
                const bool fc = d.gcn_deg_from_col != 0;
                PEA_TRY(ensure_dinv(plan, u.rel, fc, stream));
                AggGroup a{};
                fill_lists(a, Rr);
                a.W = E0;
                a.F = E0;
                a.self_loop = 1;
                a.partial = partial + part_off;
                part_off += (size_t)std::max(R.n_slots, Rr.n_slots) * rec;
                a.feat = dA0 + (size_t)ui * E0;
                a.ld_feat = L.ld_t;
                a.feat_self = a.feat;
                a.ld_self = L.ld_t;
                a.dinv = fc ? R.dinv_col : R.dinv_row;
                a.dinv_self = a.dinv;
                a.out = A0 + (size_t)ui * E0;
                a.ld_out = L.ld_t;
                a.msgs_short = (double)Rr.edges_short;
                a.msgs_long = (double)Rr.edges_long;
                a.idx_share = 1.0;
                a.table_rows = (double)Rr.src_span;
                gs.push_back(a);
                continue;
            }
            AggGroup a{};
            a.W = E0;
            a.F = E0;
            a.neg_slope = d.negative_slope;
            a.self_loop = 1;
            a.partial = partial + part_off;
            part_off += (size_t)std::max(R.n_slots, Rr.n_slots) * rec;
            a.att_src = pack + m->mlp2_att_off + (size_t)2 * ui * E0;   // ws = W_0^T att_j (mlp2_pack_kernel)
            a.att_dst = a.att_src + E0;                                  // wd = W_0^T att_i
            a.bias = nullptr;
            a.ld_side = L.ld_side;
            a.ld_k = L.ld_k;
            a.ld_g = L.ld_t;
            a.row_active = m->active0;   // D: rows with a zero gradient are not gathered for; S: their rows are not gathered
            AggGroup D = a;
            fill_lists(D, R);
            D.feat = xin;
            D.ld_feat = ldx;
            D.feat_self = xin;
            D.ld_self = ldx;
            D.g_self = dA0 + (size_t)ui * E0;
            D.o_self = A0 + (size_t)ui * E0;
            D.stats = wsf + L.off_stats + 2 * (int)ui;
            D.ld_stats = L.ld_stats;
            D.side_out = wsf + L.off_side + 4 * (int)ui;
            D.ksum = wsf + L.off_dad + (int)ui;
            D.short_rows = R.short_rows + R.n_short0;   // edge-less rows: not visited (no side record, d a_dst = 0)
            D.n_short = R.n_short - R.n_short0;
            D.msgs_short = (double)R.edges_short;
            D.msgs_long = (double)R.edges_long;
            D.table_rows = (double)R.src_span;
            gd.push_back(D);
            AggGroup S = a;
            fill_lists(S, Rr);
            S.feat = dA0 + (size_t)ui * E0;
            S.ld_feat = L.ld_t;
            S.feat_self = xin;
            S.ld_self = ldx;
            S.side = wsf + L.off_side + 4 * (int)ui;
            S.da_dst = wsf + L.off_dad + (int)ui;
            S.ksum = wsf + L.off_das + (int)ui;
            S.out = A0 + (size_t)ui * E0;
            S.ld_out = L.ld_t;
            S.deg0_self = R.deg0;
            S.msgs_short = (double)Rr.edges_short;
            S.msgs_long = (double)Rr.edges_long;
            S.table_rows = (double)Rr.src_span;
            gs.push_back(S);
        }
        for (size_t b = 0; b < gd.size(); b += kMaxAggGroups)
            PEA_TRY(launch_gat_backward(AGG_GAT_BWD_D, gd.data() + b, (int)std::min<size_t>(kMaxAggGroups, gd.size() - b), stream));
        for (size_t b = 0; b < gs.size(); b += kMaxAggGroups) {
            const int nb = (int)std::min<size_t>(kMaxAggGroups, gs.size() - b);
            if (d.kind == PEA_KIND_GCN) PEA_TRY(launch_aggregate(AGG_GCN, gs.data() + b, nb, stream));
            else PEA_TRY(launch_gat_backward(AGG_GAT_BWD_S, gs.data() + b, nb, stream));
        }
        return PEA_OK;
    }
    PEA_REQUIRE(phase == 0 || phase == 2, PEA_ERR_ARG, "backward: GAT/GCN levels have phases 0 and (sharded) 2");
    PEA_REQUIRE(phase == 0 || sharded, PEA_ERR_ARG, "backward: phase 2 is the second half of a SHARDED level");
    const bool part_a = phase == 0;               // masks, D pass, reductions over what own rows already hold
    const bool part_b = phase == 2 || !sharded;   // gathers over the reversed relation, reductions of their results
    if (d.kind == PEA_KIND_GAT && part_a) {
        // rows whose softmax is their self loop alone are skipped by the D pass (alpha = 1, d z = 0): their d a_dst reads 0
        PEA_MEMSET_ASYNC(wsf + L.off_dad, 0, (size_t)N * (size_t)L.ld_k * sizeof(float), stream);
    }
    // relu between the steps (reference models/base.py:138): the output gradient of the channels that continue is masked
    // in place, one launch per run of groups whose columns are contiguous in dO (a 2-step model's first level: the whole row)
    if (part_a && !premasked) {
        size_t ri = 0;
        while (ri < L.groups.size()) {
            const GroupPlan &g0 = L.groups[ri];
            size_t rj = ri + 1;
            int W = g0.W;
            while (!g0.last && rj < L.groups.size() && !L.groups[rj].last && L.groups[rj].out_col == g0.out_col + W) W += L.groups[rj++].W;
            if (!g0.last) PEA_TRY(launch_relu_mask(own, W, dO + g0.out_col, L.ld_o, O + g0.out_col, L.ld_o, stream));
            ri = rj;
        }
    }
    const unsigned *active_bits = nullptr;
    if (m->active_rows && part_b) {   // (GCN: the reverse aggregation skips the gathered rows known to be zero, csrc/agg.hip)
        bool any_last = false;
        for (const GroupPlan &g : L.groups) any_last = any_last || g.last;
        if (any_last) {   // the S pass tests one flag per gathered row: as a bitmap the flags of all nodes fit a CU's L1
            if (!m->active_bits) PEA_HIP(hipMalloc((void **)&m->active_bits, (size_t)((N + 63) / 64) * 2 * sizeof(unsigned)));
            PEA_LAUNCH(flags_to_bits_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, m->active_rows, m->active_bits);
            PEA_HIP(hipGetLastError());
            active_bits = m->active_bits;
        }
    }
    std::vector<AggGroup> gd, gsrc, gsrc_batch;
    for (const GroupPlan &g : L.groups) {
        const int rr = m->reverse_of[(size_t)g.rel];
        PEA_REQUIRE(rr >= 0, PEA_ERR_ARG, "backward: relation %d has no reversed relation in the plan", g.rel);
        Relation &R = plan->rels[(size_t)g.rel], &Rr = plan->rels[(size_t)rr];
        PEA_REQUIRE(g.partial_off + (size_t)std::max(R.n_slots, Rr.n_slots) * partial_record_floats(g.W, g.F) <= m->partial_floats,
                    PEA_ERR_NOMEM, "backward: hub partial buffer too small for relation %d and its reverse", g.rel);
        float *G = g.last ? dX + g.out_col : dO + g.out_col;
        const float *Out = g.last ? X + g.out_col : O + g.out_col;
        const int ldg = g.last ? m->ld_x : L.ld_o;
        AggGroup a{};
        a.W = g.W;
        a.F = g.F;
        a.neg_slope = d.negative_slope;
        a.self_loop = loops ? 1 : 0;
        a.partial = partial + g.partial_off;  // every group its own region (sized for the relation and its reverse: slots_of)
        if (d.kind == PEA_KIND_GCN) {
            const bool fc = d.gcn_deg_from_col != 0;
            PEA_TRY(ensure_dinv(plan, g.rel, fc, stream));
            fill_lists(a, Rr);
            a.feat = G;
            a.ld_feat = ldg;
            a.feat_self = G;
            a.ld_self = ldg;
            a.dinv = fc ? R.dinv_col : R.dinv_row;
            a.dinv_self = a.dinv;
            a.out = dT + g.col;
            a.ld_out = L.ld_t;
            if (g.last && active_bits) {   // only the batch's rows of dX are non-zero: the batch-sparse walk (agg_bwd.hip)
                a.row_active = m->active_rows;
                a.row_active_bits = active_bits;
                if (part_b) gsrc_batch.push_back(a);
            } else if (part_b) {
                gsrc.push_back(a);
            }
            continue;
        }
        a.att_src = pack + L.att_src_off + g.col;
        a.att_dst = pack + L.att_dst_off + g.col;
        a.bias = pack + g.bias_off;
        a.ld_side = L.ld_side;
        a.ld_k = L.ld_k;
        a.ld_g = ldg;
        a.row_active = g.last ? m->active_rows : nullptr;  // only the final outputs' gradient is known to be batch-sparse
        a.row_active_bits = g.last ? active_bits : nullptr;
        // D pass: destination rows of the forward relation, gathers T_j
        AggGroup D = a;
        fill_lists(D, R);
        D.feat = T + g.col;
        D.ld_feat = L.ld_t;
        D.feat_self = D.feat;
        D.ld_self = L.ld_t;
        D.g_self = G;
        D.o_self = Out;
        D.stats = wsf + L.off_stats + 2 * g.a_k;
        D.ld_stats = L.ld_stats;
        D.side_out = wsf + L.off_side + 4 * g.a_k;
        D.ksum = wsf + L.off_dad + g.a_k;
        if (sharded && level > 0) {  // the gather sources of this level sit in the forward's exchange buffer (slot order)
            PEA_REQUIRE(R.col_slot != nullptr && g.xch_ld > 0, PEA_ERR_ARG, "backward: relation %d has no exchange layout", g.rel);
            D.col = R.col_slot;
            D.feat = wsf + g.xch_off;
            D.ld_feat = g.xch_ld;
        }
        if (loops) {  // edge-less rows come first in the short-row list: not visited (no side record, d a_dst = 0)
            D.short_rows = R.short_rows + R.n_short0;
            D.n_short = R.n_short - R.n_short0;
        }
        D.msgs_short = (double)R.edges_short;   // bookkeeping for the live roofline (agg_bwd.hip: launch_bwd_g)
        D.msgs_long = (double)R.edges_long;
        D.table_rows = (sharded && level > 0) ? (double)R.slots_per_rank * plan->shard_world : (double)R.src_span;
        if (part_a) gd.push_back(D);
        // S pass: source rows = destination rows of the reversed relation, gathers g_i and the side records
        AggGroup S = a;
        fill_lists(S, Rr);
        S.feat = G;
        S.ld_feat = ldg;
        S.feat_self = T + g.col;
        S.ld_self = L.ld_t;
        S.side = wsf + L.off_side + 4 * g.a_k;
        S.da_dst = wsf + L.off_dad + g.a_k;
        S.ksum = wsf + L.off_das + g.a_k;
        S.out = dT + g.col;
        S.ld_out = L.ld_t;
        S.deg0_self = loops ? R.deg0 : nullptr;
        S.msgs_short = (double)Rr.edges_short;
        S.msgs_long = (double)Rr.edges_long;
        S.table_rows = (double)Rr.src_span;
        if (part_b) gsrc.push_back(S);
    }
    // all groups of the level side by side in one set of launches per pass (the S pass of a group reads what the D pass
    // of the same group wrote: every D launch precedes every S launch on the stream)
    for (size_t b = 0; b < gd.size(); b += kMaxAggGroups)
        PEA_TRY(launch_gat_backward(AGG_GAT_BWD_D, gd.data() + b, (int)std::min<size_t>(kMaxAggGroups, gd.size() - b), stream));
    for (size_t b = 0; b < gsrc.size(); b += kMaxAggGroups) {
        const int nb = (int)std::min<size_t>(kMaxAggGroups, gsrc.size() - b);
        if (d.kind == PEA_KIND_GCN) PEA_TRY(launch_aggregate(AGG_GCN, gsrc.data() + b, nb, stream));
        else PEA_TRY(launch_gat_backward(AGG_GAT_BWD_S, gsrc.data() + b, nb, stream));
    }
    for (size_t b = 0; b < gsrc_batch.size(); b += kMaxAggGroups)   // GCN, last layer
        PEA_TRY(launch_gat_backward(AGG_SUM_BWD_S, gsrc_batch.data() + b, (int)std::min<size_t>(kMaxAggGroups, gsrc_batch.size() - b), stream));
    // Gradient reductions, one launch per run of groups whose columns (and heads) are contiguous:
    //   d bias[c] = sum_n g[n, c];   d att_j[c] = sum_n d a_src[n, head(c)] T[n, c];   d att_i likewise with d a_dst
    size_t gi = 0;
    while (gi < L.groups.size()) {
        const GroupPlan &g0 = L.groups[gi];
        size_t gj = gi + 1;
        int W = g0.W;
        while (gj < L.groups.size() && L.groups[gj].last == g0.last && L.groups[gj].F == g0.F &&
               L.groups[gj].col == g0.col + W && L.groups[gj].out_col == g0.out_col + W)
            W += L.groups[gj++].W;
        float *G = g0.last ? dX + g0.out_col : dO + g0.out_col;
        const int ldg = g0.last ? m->ld_x : L.ld_o;
        if (part_a) PEA_TRY(launch_colsum(own, W, W, G, ldg, nullptr, 0, 1.0f, colsum_part, gpack + g0.bias_off, stream));
        if (d.kind == PEA_KIND_GAT) {
            float *das = wsf + L.off_das + g0.a_k, *dad = wsf + L.off_dad + g0.a_k;
            if (part_a && part_b) {  // one GPU: both attention-vector gradients weight T_s: one pass over it
                PEA_TRY(launch_colsum2(own, W, g0.F, T + g0.col, L.ld_t, das, dad, L.ld_k, 1.0f, colsum_part,
                                       gpack + L.att_src_off + g0.col, gpack + L.att_dst_off + g0.col, stream));
            } else {
                if (part_b)
                    PEA_TRY(launch_colsum(own, W, g0.F, T + g0.col, L.ld_t, das, L.ld_k, 1.0f, colsum_part,
                                          gpack + L.att_src_off + g0.col, stream));
                if (part_a)
                    PEA_TRY(launch_colsum(own, W, g0.F, T + g0.col, L.ld_t, dad, L.ld_k, 1.0f, colsum_part,
                                          gpack + L.att_dst_off + g0.col, stream));
            }
        }
        gi = gj;
    }
    return PEA_OK;
}

// Flat description of the schedule for the host mirror (all offsets in floats from the 256-byte aligned workspace base):
//   [0] n_levels  [1] ld_x  [2] off_x  [3] off_dx  [4] off_gpack  [5] pack_floats  [6] two-step training schedule (0 / 1)
//   then per level  ld_t ld_o off_t off_o off_dt off_do off_side bias_off att_src_off att_dst_off off_dad off_das ld_k n_units,
//   then per unit
//   p s rel in_w heads F HF last in_col t_col o_col b_off ldb bias_off
extern "C" int pea_model_describe(const pea_model *m, int64_t *out, int max_len, int *needed) {
    PEA_REQUIRE(m && needed, PEA_ERR_ARG, "describe: null");
    std::vector<int64_t> v = {(int64_t)m->levels.size(), m->ld_x, (int64_t)m->off_x, (int64_t)m->off_dx,
                              (int64_t)m->off_gpack, (int64_t)m->pack_floats, m->fused2_train ? 1 : 0};
    for (const Level &L : m->levels) {
        const int64_t head[] = {L.ld_t, L.ld_o, (int64_t)L.off_t, (int64_t)L.off_o, (int64_t)L.off_dt, (int64_t)L.off_do,
                                (int64_t)L.off_side, (int64_t)L.bias_off, (int64_t)L.att_src_off, (int64_t)L.att_dst_off,
                                (int64_t)L.off_dad, (int64_t)L.off_das, (int64_t)L.ld_k, (int64_t)L.units.size()};
        v.insert(v.end(), head, head + 14);
        for (const Unit &u : L.units) {
            const int64_t un[] = {u.p, u.s, u.rel, u.in_w, u.heads, u.F, u.HF, u.last ? 1 : 0, u.in_col, u.t_col, u.o_col,
                                  (int64_t)u.b_off, u.ldb, (int64_t)u.bias_off};
            v.insert(v.end(), un, un + 14);
        }
    }
    *needed = (int)v.size();
    if (out && max_len >= (int)v.size()) std::copy(v.begin(), v.end(), out);
    return PEA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Weighted neighbour sum with caller-supplied per-edge weights (the message/aggregate half of the reference's own
// baseline convs: graph_recsys_benchmark/nn/kgat_conv.py:36-44, nn/kgcn_conv.py:32-37 `x_j * att_map`, and
// nn/ngcf_conv.py:42-45 with coff folded to a per-edge weight):   out_i = sum_{e: j -> i} w_e x_j.
// The plan must have been created with PEA_PLAN_EDGE_IDS (weights are in the caller's COO order).
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t pea_weighted_aggregate_workspace_bytes(const pea_plan *plan, int relation, int width) {
    if (!plan || relation < 0 || relation >= (int)plan->rels.size() || width <= 0) return 0;
    size_t fl = 0;
    for (int c = 0; c < width; c += 256) fl += (size_t)plan->rels[(size_t)relation].n_slots * partial_record_floats(std::min(256, width - c), std::min(256, width - c));
    return fl * sizeof(float) + 512;
}

extern "C" int pea_weighted_aggregate(const pea_plan *plan, int relation, int width, const float *x, int64_t ldx,
                                      const float *edge_weight, float *out, int64_t ldo, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    PEA_REQUIRE(plan && relation >= 0 && relation < (int)plan->rels.size(), PEA_ERR_ARG, "weighted_aggregate: bad relation");
    const Relation &R = plan->rels[(size_t)relation];
    PEA_REQUIRE(R.eid != nullptr || R.e_kept == 0, PEA_ERR_ARG, "weighted_aggregate: the plan was created without PEA_PLAN_EDGE_IDS");
    PEA_REQUIRE(x && out && (edge_weight || R.e_kept == 0), PEA_ERR_ARG, "weighted_aggregate: null argument");
    PEA_REQUIRE(width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= width && ldo >= width, PEA_ERR_ARG,
                "weighted_aggregate: width %d and row strides must be multiples of 4", width);
    PEA_REQUIRE(workspace_bytes >= pea_weighted_aggregate_workspace_bytes(plan, relation, width), PEA_ERR_NOMEM,
                "weighted_aggregate: workspace too small");
    float *partial = aligned_ws(workspace);
    std::vector<AggGroup> gs;
    for (int c = 0; c < width; c += 256) {
        AggGroup a{};
        fill_lists(a, R);
        a.eid = R.eid;
        a.edge_w = edge_weight;
        a.W = std::min(256, width - c);
        a.F = a.W;
        a.feat = x + c;
        a.ld_feat = (int)ldx;
        a.feat_self = a.feat;
        a.ld_self = (int)ldx;
        a.out = out + c;
        a.ld_out = (int)ldo;
        a.partial = partial;
        partial += (size_t)R.n_slots * partial_record_floats(a.W, a.F);
        gs.push_back(a);
    }
    for (size_t b = 0; b < gs.size(); b += kMaxAggGroups)
        PEA_TRY(launch_aggregate(AGG_WSUM, gs.data() + b, (int)std::min<size_t>(kMaxAggGroups, gs.size() - b), (hipStream_t)stream));
    return PEA_OK;
}
