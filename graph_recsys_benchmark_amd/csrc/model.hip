// Schedule of one PEA forward on gfx950: the per-metapath channel loop of
// graph_recsys_benchmark/models/base.py:129-140 (PEABaseChannel.forward) run for all P channels of
// models/base.py:191-193, then the fusion of models/base.py:196-203 -- as a fixed sequence of launches on
// one stream (capturable in a hipGraph by the caller; nothing here synchronises or allocates).
//
// The reference runs P x S independent conv calls.  Here the work is regrouped per LEVEL (step index):
//   - all channels of a level that read the same input share one GEMM job (level 0: every channel
//     reads x, models/base.py:192);
//   - channels of a level that aggregate over the same relation are fused HORIZONTALLY: their feature
//     columns sit side by side, so the relation's index array is read once and a gathered row is one
//     contiguous run (7 of the 9 MovieLens metapaths end in flip(user2item),
//     utils/general_utils.py:300-307 / 335-343);
// Buffers (row-major fp32, row strides multiples of 4 floats), per level s:
//   GAT/GCN : T_s [N, sum HF] transformed features (gather source; the GAT logits are computed from the
//             gathered rows, agg.hip), O_s [N, sum HF] relu(conv) output = input of level s+1
//   SAGE    : M_s [N, ...] neighbour means of the level's input, O_s [N, sum F]
//   X [N, P*R]: last-layer outputs of every channel (the "stack" the fusion reads).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "model.h"

namespace pea {
namespace {

int pad4(int v) { return (v + 3) & ~3; }
// row strides of the big node-major buffers are whole 128-byte cache lines and every buffer starts on a 256-byte
// boundary, so a gathered 256-byte row chunk touches 2 lines, never 3
int pad_ld(int v) { return (v + 31) & ~31; }
size_t pad_off(size_t v) { return (v + 63) & ~(size_t)63; }

int width_out(const pea_model_desc &d, int S, int s, int *heads_out) {
    // PEA{GAT,GCN,Sage}Channel (models/peagat.py:14-21): 1 step: emb -> repr with num_heads heads;
    // else emb -> hidden (num_heads) -> ... -> repr (1 head)
    const bool last = s == S - 1;
    int heads = d.kind == PEA_KIND_GAT ? d.heads : 1;
    if (S > 1 && last) heads = 1;
    *heads_out = heads;
    return last ? d.repr_dim : d.hidden_size;
}

// hub chunks a group may need: its relation's, or (training) the reversed relation's the backward walks
int slots_of(const pea_model *m, int rel) {
    int n = m->plan->rels[(size_t)rel].n_slots;
    if (m->backward && rel < (int)m->reverse_of.size() && m->reverse_of[(size_t)rel] >= 0)
        n = std::max(n, m->plan->rels[(size_t)m->reverse_of[(size_t)rel]].n_slots);
    return n;
}

int rel_at(const pea_model *m, int p, int s) { return m->relation_of[(size_t)(m->chan_first[(size_t)p] + s)]; }

int build_schedule(pea_model *m) {
    const pea_model_desc &d = m->d;
    const pea_plan *plan = m->plan;
    const int P = d.num_channels;
    const int64_t N = plan->N;
    // SAGE: a model with training buffers keeps the reference's order (mean of the INPUT rows, then lin_rel on the mean:
    // the backward is written for it); without them it runs on the GAT/GCN schedule, see pea_model::sage2
    // (two-step training schedule, fused2_train below: SAGE trains on the GAT/GCN schedule too -- mean and lin_rel commute)
    int Smax = 0;
    for (int p = 0; p < P; ++p) Smax = std::max(Smax, m->steps[(size_t)p]);
    {
        bool all2 = true;
        for (int p = 0; p < P; ++p) all2 = all2 && m->steps[(size_t)p] == 2;
        const char *env = getenv("PEA_FUSED2");
        // GAT / GCN: plans with self loops (a row without incoming edges aggregates to itself: mlp2 reads x for it);
        // SAGE: no self loops by definition, such a row's mean is 0
        const bool loops_ok = d.kind == PEA_KIND_SAGE ? !(plan->flags & PEA_PLAN_SELF_LOOPS) : (plan->flags & PEA_PLAN_SELF_LOOPS) != 0;
        m->fused2 = all2 && !m->backward && !m->single_conv && (d.heads == 1 || d.kind != PEA_KIND_GAT) && P <= kMaxMlp2Chan &&
                    mlp2_supported(d.kind, d.emb_dim, d.hidden_size, d.repr_dim) && loops_ok && !(env && atoi(env) == 0);
        const char *envt = getenv("PEA_FUSED2_TRAIN");
        m->fused2_train = all2 && m->backward && !m->single_conv && (d.kind != PEA_KIND_GAT || d.heads == 1) &&
                          P <= kMaxMlp2Chan &&
                          mlp2_supported(d.kind, d.emb_dim, d.hidden_size, d.repr_dim) && loops_ok && d.emb_dim == d.hidden_size &&
                          plan->shard_world == 1 && !(envt && atoi(envt) == 0);
        m->ld_a0 = pad_ld(P * d.emb_dim);
    }
    m->sage2 = d.kind == PEA_KIND_SAGE && (!m->backward || m->fused2_train);
    const bool sage = d.kind == PEA_KIND_SAGE && !m->sage2, gat = d.kind == PEA_KIND_GAT;
    m->levels.assign((size_t)Smax, Level());
    std::vector<int> in_w((size_t)P, d.emb_dim), in_col((size_t)P, 0);
    int x_cols = 0;
    size_t pack = 0, ws = 0, partial_max = 0, xch = 0;
    m->messages = 0;
    m->alg_bytes = 0.0;
    // Compulsory HBM bytes of THIS schedule (not of the reference's): x read once; per level T_s written once and read
    // once (however often its rows are gathered afterwards: re-reads can come from cache), O_s of the channels that
    // continue written once and read once, every group's index arrays read once; X written and read once; the fused
    // table written once.  step time x 8 TB/s over this = how far the whole step is from the DRAM floor.
    // A rank of a sharded plan writes and reads its own rows only (x is replicated: read whole).
    const double Nr = (plan->shard_world > 1 && plan->n_owned > 0) ? (double)plan->n_owned : (double)N;
    m->compulsory_bytes = 4.0 * (double)N * d.emb_dim;

    for (int s = 0; s < Smax; ++s) {
        Level &L = m->levels[(size_t)s];
        for (int p = 0; p < P; ++p) {
            const int S = m->steps[(size_t)p];
            if (S <= s) continue;
            Unit u;
            u.p = p;
            u.s = s;
            u.rel = rel_at(m, p, s);
            u.in_w = in_w[(size_t)p];
            u.F = width_out(d, S, s, &u.heads);
            u.HF = u.heads * u.F;
            u.last = s == S - 1;
            u.in_col = in_col[(size_t)p];
            PEA_REQUIRE(u.rel >= 0 && u.rel < (int)plan->rels.size(), PEA_ERR_ARG, "model: relation index %d out of range", u.rel);
            PEA_REQUIRE(u.in_w % 4 == 0 && u.F % 4 == 0 && u.F > 0, PEA_ERR_ARG,
                        "model: layer widths must be positive multiples of 4 (in %d, out %d)", u.in_w, u.F);
            PEA_REQUIRE(u.F <= 256, PEA_ERR_ARG, "model: per-head width %d > 256 is not supported", u.F);
            L.units.push_back(u);
        }
        // column order: channels that aggregate over the same relation side by side
        std::stable_sort(L.units.begin(), L.units.end(), [](const Unit &a, const Unit &b) {
            if (a.rel != b.rel) return a.rel < b.rel;
            if (a.last != b.last) return a.last < b.last;
            return a.F < b.F;
        });
        int col = 0, ak = 0;
        for (Unit &u : L.units) {
            u.t_col = col;
            u.a_k = ak;
            col += sage ? u.in_w : u.HF;
            ak += u.heads;
        }
        L.n_cols = col;
        L.n_heads = ak;
        L.ld_t = pad_ld(col);
        L.ld_a = 0;
        // outputs of channels that continue: GAT/GCN keep the T_s column order; SAGE lays O_s out in the order the
        // NEXT level aggregates it (same relation side by side)
        std::vector<Unit *> cont;
        for (Unit &u : L.units)
            if (!u.last) cont.push_back(&u);
        if (sage)
            std::stable_sort(cont.begin(), cont.end(), [&](const Unit *a, const Unit *b) {
                return rel_at(m, a->p, s + 1) < rel_at(m, b->p, s + 1);
            });
        int ocol = 0;
        for (Unit *u : cont) {
            u->o_col = sage ? ocol : u->t_col;
            ocol += u->HF;
            in_w[(size_t)u->p] = u->HF;
            in_col[(size_t)u->p] = u->o_col;
        }
        L.ld_o = sage ? pad_ld(ocol) : L.ld_t;
        for (Unit &u : L.units) {
            if (!u.last) continue;
            u.o_col = x_cols;
            m->x_col.c[u.p] = x_cols;
            x_cols += u.HF;
            // (a model of ONE channel has nothing to stack: its X is simply heads * repr_dim wide -- the autograd path of
            // the per-layer conv drop-ins builds such models, nn/conv.py)
            PEA_REQUIRE(m->single_conv || P == 1 || u.HF == d.repr_dim, PEA_ERR_ARG,
                        "model: channel %d ends %d columns wide but repr_dim is %d (a 1-step GAT channel with "
                        "num_heads > 1 cannot be stacked; the reference fails at models/base.py:196 too)",
                        u.p, u.HF, d.repr_dim);
        }
        // packed weights + biases
        L.shared_input = !sage && s == 0;
        if (L.shared_input) {
            L.n_out = pad4(L.n_cols);
            L.ldb = L.n_out;
            L.b_off = pack;
            pack += (size_t)d.emb_dim * (size_t)L.ldb;
            for (Unit &u : L.units) {
                u.b_off = L.b_off + (size_t)u.t_col;
                u.ldb = L.ldb;
            }
        } else {
            for (Unit &u : L.units) {
                const int K = sage ? 2 * u.in_w : u.in_w;
                u.ldb = pad4(u.HF);
                u.b_off = pack;
                pack += (size_t)K * (size_t)u.ldb;
            }
        }
        if (m->sage2)
            for (Unit &u : L.units) {
                u.ld_root = pad4(u.HF);
                u.root_off = pack;
                pack += (size_t)u.in_w * (size_t)u.ld_root;
            }
        if (sage) {
            for (Unit &u : L.units) {
                u.bias_off = pack;
                pack += (size_t)u.ldb;
            }
        } else {  // one bias row per level, in column order, so an aggregation group's bias is contiguous
            L.bias_off = pack;
            pack += (size_t)L.ld_t;
            for (Unit &u : L.units) u.bias_off = L.bias_off + (size_t)u.t_col;
            if (gat) {
                L.att_src_off = pack;
                pack += (size_t)L.ld_t;
                L.att_dst_off = pack;
                pack += (size_t)L.ld_t;
            }
        }
        // aggregation groups
        size_t partial = 0;
        size_t i = 0;
        if (sage) {
            int mcol = 0;
            while (i < L.units.size()) {
                size_t j = i;
                while (j < L.units.size() && L.units[j].rel == L.units[i].rel) ++j;
                int c_beg, c_end;
                if (s == 0) {  // every channel reads x: one mean per distinct relation, shared by its channels
                    c_beg = 0;
                    c_end = d.emb_dim;
                } else {
                    c_beg = INT32_MAX;
                    c_end = 0;
                    for (size_t k = i; k < j; ++k) {
                        c_beg = std::min(c_beg, L.units[k].in_col);
                        c_end = std::max(c_end, L.units[k].in_col + L.units[k].in_w);
                    }
                    int covered = 0;
                    for (size_t k = i; k < j; ++k) covered += L.units[k].in_w;
                    PEA_REQUIRE(covered == c_end - c_beg, PEA_ERR_ARG, "model: internal layout error (SAGE run not contiguous)");
                }
                for (size_t k = i; k < j; ++k) L.units[k].t_col = mcol + (s == 0 ? 0 : L.units[k].in_col - c_beg);
                for (int c = c_beg; c < c_end; c += 256) {
                    GroupPlan g;
                    g.rel = L.units[i].rel;
                    g.col = c;
                    g.W = std::min(256, c_end - c);
                    g.F = g.W;
                    g.out_col = mcol + (c - c_beg);
                    g.partial_off = partial;
                    partial += (size_t)slots_of(m, g.rel) * partial_record_floats(g.W, g.F);
                    L.groups.push_back(g);
                }
                mcol += c_end - c_beg;
                i = j;
            }
            L.ld_t = pad_ld(mcol);
        } else {
            while (i < L.units.size()) {
                const Unit &u0 = L.units[i];
                size_t j = i;
                while (j < L.units.size() && L.units[j].rel == u0.rel && L.units[j].last == u0.last && L.units[j].F == u0.F) ++j;
                const int c_beg = u0.t_col, c_end = L.units[j - 1].t_col + L.units[j - 1].HF;
                const int per = std::max(1, 256 / u0.F) * u0.F;  // whole heads per group, <= 256 columns
                for (int c = c_beg; c < c_end; c += per) {
                    GroupPlan g;
                    g.rel = u0.rel;
                    g.col = c;
                    g.W = std::min(per, c_end - c);
                    g.F = u0.F;
                    g.a_k = u0.a_k + (c - c_beg) / u0.F;
                    g.last = u0.last;
                    g.out_col = (u0.last ? u0.o_col : c_beg) + (c - c_beg);
                    g.bias_off = L.bias_off + (size_t)c;
                    g.n_convs = 0;
                    for (size_t k = i; k < j; ++k)
                        if (L.units[k].t_col < c + g.W && L.units[k].t_col + L.units[k].HF > c) ++g.n_convs;
                    g.partial_off = partial;
                    partial += (size_t)slots_of(m, g.rel) * partial_record_floats(g.W, g.F);
                    L.groups.push_back(g);
                }
                i = j;
            }
        }
        if ((m->fused2 || m->fused2_train) && s == 0) {  // one aggregation group per channel, emb columns wide, one "head"
            size_t p0 = 0;
            for (const Unit &u : L.units) p0 += (size_t)slots_of(m, u.rel) * partial_record_floats(d.emb_dim, d.emb_dim);
            partial = std::max(partial, p0);
        }
        partial_max = std::max(partial_max, partial);
        if (plan->shard_world > 1 && s > 0) {
            for (GroupPlan &g : L.groups) {
                g.xch_ld = pad_ld(g.W);
                g.xch_off = xch;  // relative; rebased below
                xch = pad_off(xch + (size_t)plan->shard_world * (size_t)plan->rels[(size_t)g.rel].slots_per_rank * (size_t)g.xch_ld);
            }
        }
        L.off_t = ws;
        ws = pad_off(ws + (size_t)N * (size_t)((m->fused2 && s == 0) ? std::max(L.ld_t, m->ld_a0) : L.ld_t));
        L.off_a = ws;
        L.off_o = ws;
        ws = pad_off(ws + (size_t)N * (size_t)L.ld_o);
        {
            double cont_cols = 0.0;
            for (const Unit &u : L.units)
                if (!u.last) cont_cols += u.HF;
            if (m->fused2 && s == 0) {
                // two-step schedule: no T_0 / O_0; per channel its index arrays and the aggregates A_0 of the rows that
                // have incoming edges (written once, read once)
                int prev_rel = -1;
                for (const Unit &u : L.units) {   // (SAGE: one mean per distinct relation, shared by its channels)
                    if (d.kind == PEA_KIND_SAGE && u.rel == prev_rel) continue;
                    prev_rel = u.rel;
                    const Relation &R = plan->rels[(size_t)u.rel];
                    const double rows_with_edges = (double)(R.n_short - R.n_short0) + R.n_direct + R.n_hub;
                    m->compulsory_bytes += 4.0 * (double)R.e_kept + 4.0 * (Nr + 1.0) + 8.0 * rows_with_edges * d.emb_dim;
                }
            } else {
                m->compulsory_bytes += 8.0 * Nr * L.n_cols + 8.0 * Nr * cont_cols;
                for (const GroupPlan &g : L.groups)
                    m->compulsory_bytes += 4.0 * (double)plan->rels[(size_t)g.rel].e_kept + 4.0 * (Nr + 1.0);
            }
        }
        // statistics: messages and the algorithmic-byte yardstick of SURVEY.md 8(d)
        for (const Unit &u : L.units) {
            const Relation &R = plan->rels[(size_t)u.rel];
            const double Nn = (double)N;
            if (d.kind == PEA_KIND_SAGE) {  // the yardstick counts the reference's conv calls, whatever the schedule
                m->messages += R.e_kept;
                m->alg_bytes += (double)R.e_kept * (4.0 * u.in_w + 4.0) + 4.0 * (Nn + 1) + 4.0 * Nn * u.in_w + 4.0 * Nn * u.HF;
            } else {
                const double M = (double)R.e_kept + Nn;
                m->messages += R.e_kept + N;
                m->alg_bytes += M * (4.0 * u.HF + 4.0 + 4.0 * u.heads) + 4.0 * (Nn + 1) + 4.0 * Nn * u.in_w + 8.0 * Nn * u.HF +
                                (gat ? 8.0 * Nn * u.heads : 0.0);
            }
        }
    }
    if (m->fused2 || m->fused2_train) {
        m->mlp2_img_off = pack;
        pack = pad_off(pack + (size_t)P * (mlp2_image_bytes(d.kind, d.emb_dim, d.hidden_size) / sizeof(float)));
        m->mlp2_att_off = pack;
        pack = pad_off(pack + (size_t)2 * P * (size_t)d.emb_dim);
    }
    m->ld_x = pad_ld(x_cols);
    m->alg_bytes += 4.0 * (double)N * P * d.repr_dim + 4.0 * (double)N * d.repr_dim;
    m->compulsory_bytes += 8.0 * Nr * x_cols + 4.0 * Nr * d.repr_dim;
    m->pack_floats = pad_off(pack);
    size_t off = m->pack_floats;
    for (Level &L : m->levels) {
        L.off_t += off;
        L.off_a += off;
        L.off_o += off;
    }
    off += ws;
    m->off_x = off;
    off = pad_off(off + (size_t)N * (size_t)m->ld_x);
    m->off_partial = off;
    m->partial_floats = pad_off(partial_max);
    off += m->partial_floats;
    for (Level &L : m->levels)
        for (GroupPlan &g : L.groups)
            if (g.xch_ld) g.xch_off += off;
    off += xch;
    if (m->backward) {  // training: softmax statistics, gradient buffers, gradient pack (same layout as the weight pack)
        int max_w = 4;
        for (Level &L : m->levels) {
            L.ld_stats = pad4(2 * std::max(L.n_heads, 1));
            L.ld_k = pad4(std::max(L.n_heads, 1));
            L.ld_side = pad4(4 * std::max(L.n_heads, 1));
            L.off_stats = off; off = pad_off(off + (size_t)N * (size_t)L.ld_stats);
            L.off_dt = off;    off = pad_off(off + (size_t)N * (size_t)L.ld_t);
            L.off_do = off;    off = pad_off(off + (size_t)N * (size_t)std::max(L.ld_o, 4));
            // (SAGE on the two-step training schedule: the first level's side region holds the root-term gradient blocks)
            const bool side_rows = sage || (m->fused2_train && d.kind == PEA_KIND_SAGE && &L == &m->levels[0]);
            L.off_side = off;  off = pad_off(off + (size_t)N * (size_t)std::max(L.ld_side, side_rows ? L.ld_t : 4));
            L.off_dad = off;   off = pad_off(off + (size_t)N * (size_t)L.ld_k);
            L.off_das = off;   off = pad_off(off + (size_t)N * (size_t)L.ld_k);
            max_w = std::max(max_w, std::max(L.ld_t, L.ld_o));
        }
        m->off_dx = off;      off = pad_off(off + (size_t)N * (size_t)m->ld_x);
        m->off_gpack = off;   m->gpack_floats = m->pack_floats; off = pad_off(off + m->gpack_floats);
        m->off_colsum = off;  off = pad_off(off + (size_t)2 * kColsumParts * (size_t)std::max(max_w, m->ld_x));  // two sums per pass
    }
    m->total_floats = off;
    return PEA_OK;
}

int init_model(pea_model *m, const pea_plan *plan, const pea_model_desc *desc) {
    PEA_REQUIRE(plan && desc, PEA_ERR_ARG, "model: null plan / desc");
    PEA_REQUIRE(desc->kind == PEA_KIND_GAT || desc->kind == PEA_KIND_GCN || desc->kind == PEA_KIND_SAGE, PEA_ERR_ARG, "model: kind %d", desc->kind);
    PEA_REQUIRE(desc->num_channels > 0 && desc->num_channels <= kMaxChannels && desc->steps && desc->relation_of, PEA_ERR_ARG,
                "model: %d channels (1..%d)", desc->num_channels, kMaxChannels);
    PEA_REQUIRE(desc->emb_dim > 0 && desc->hidden_size > 0 && desc->repr_dim > 0 && desc->heads > 0, PEA_ERR_ARG, "model: bad widths");
    PEA_REQUIRE(desc->fuse_mode == PEA_FUSE_ATT || desc->fuse_mode == PEA_FUSE_MEAN, PEA_ERR_ARG,
                "model: fuse mode %d ('concat' is unusable in the reference, models/base.py:175 vs :197)", desc->fuse_mode);
    m->plan = plan;
    m->d = *desc;
    m->steps.assign(desc->steps, desc->steps + desc->num_channels);
    int total = 0;
    m->chan_first.clear();
    for (int p = 0; p < desc->num_channels; ++p) {
        PEA_REQUIRE(m->steps[(size_t)p] >= 1, PEA_ERR_ARG, "model: channel %d has %d steps", p, m->steps[(size_t)p]);
        m->chan_first.push_back(total);
        total += m->steps[(size_t)p];
    }
    m->relation_of.assign(desc->relation_of, desc->relation_of + total);
    m->backward = desc->enable_backward != 0;
    m->reverse_of.clear();
    if (m->backward) {
        PEA_REQUIRE(desc->reverse_of != nullptr, PEA_ERR_ARG, "model: enable_backward needs reverse_of (relation -> reversed relation)");
        m->reverse_of.assign(desc->reverse_of, desc->reverse_of + plan->rels.size());
    }
    m->d.steps = nullptr;
    m->d.relation_of = nullptr;
    m->d.reverse_of = nullptr;
    m->n_slots_per_layer = desc->kind == PEA_KIND_GAT ? 4 : desc->kind == PEA_KIND_GCN ? 2 : 3;
    return build_schedule(m);
}

}  // namespace

// params: [sum steps][slots] device pointers, channel-major.  ldx: row stride of x.
// out_x / ld_out_x: when non-null the last-layer outputs go there instead of the workspace X (single-conv
// entry points); relu_last applies relu to last layers too.
int model_forward(pea_model *m, int stage, const float *const *params, const float *x, int64_t ldx, const float *att,
                  int masked, float *wsf, float *out_repr, float *out_stack, float *out_x, int64_t ld_out_x, int relu_last,
                  hipStream_t stream, bool training, int part, const FuseSelect *sel) {
    const pea_model_desc &d = m->d;
    pea_plan *plan = const_cast<pea_plan *>(m->plan);
    const int64_t N = plan->N;
    const int kind = d.kind;
    const int slots = m->n_slots_per_layer;
    float *pack = wsf;
    float *X = out_x ? out_x : wsf + m->off_x;
    const int64_t ldX = out_x ? ld_out_x : m->ld_x;
    float *partial = wsf + m->off_partial;
    auto param = [&](const Unit &u, int slot) -> const float * {
        return params[(size_t)(m->chan_first[(size_t)u.p] + u.s) * (size_t)slots + (size_t)slot];
    };

    const bool sharded = plan->shard_world > 1;
    const int n_levels = (int)m->levels.size();
    if (sharded) PEA_REQUIRE(plan->owned_rows != nullptr || plan->n_owned == 0, PEA_ERR_ARG, "sharded plan without owned rows");
    const int *own_rows = sharded ? plan->owned_rows : nullptr;
    const int64_t n_own = sharded ? plan->n_owned : N;

    // ---- pack weights (they change every optimizer step) ----
    auto pack_weights = [&]() -> int {
    std::vector<PackJob> pj;
        for (Level &L : m->levels) {
            for (size_t ui = 0; ui < L.units.size(); ++ui) {
                const Unit &u = L.units[ui];
                PackJob j{};
                j.kind = kind;
                j.B = pack + u.b_off;
                j.ldb = u.ldb;
                j.in = u.in_w;
                j.HF = u.HF;
                j.F = u.F;
                j.bias = pack + u.bias_off;
                j.w0 = param(u, 0);
                PEA_REQUIRE(j.w0 != nullptr, PEA_ERR_ARG, "forward: null weight pointer (channel %d step %d)", u.p, u.s);
                if (kind == PEA_KIND_GAT) {
                    j.w1 = param(u, 1);
                    j.w2 = param(u, 2);
                    j.w3 = param(u, 3);
                    PEA_REQUIRE(j.w1 && j.w2, PEA_ERR_ARG, "forward: null att_i/att_j (channel %d step %d)", u.p, u.s);
                    j.att_src = pack + L.att_src_off + u.t_col;
                    j.att_dst = pack + L.att_dst_off + u.t_col;
                } else if (kind == PEA_KIND_GCN) {
                    j.w3 = param(u, 1);
                } else {
                    j.w3 = param(u, 1);
                    j.w1 = param(u, 2);
                    PEA_REQUIRE(j.w1 != nullptr, PEA_ERR_ARG, "forward: null lin_root.weight (channel %d step %d)", u.p, u.s);
                    if (m->sage2) {
                        j.kind = PEA_PACK_SAGE2;
                        j.B2 = pack + u.root_off;
                        j.ldb2 = u.ld_root;
                    }
                }
                if (L.shared_input) {  // the last unit clears the block's padding columns
                    if (ui + 1 == L.units.size()) {
                        const int used = L.n_cols;
                        j.zero_col = used - u.t_col;
                        j.zero_n = L.n_out - used;
                    }
                } else {
                    const int used = u.HF;
                    j.zero_col = used;
                    j.zero_n = u.ldb - used;
                }
                pj.push_back(j);
            }
        }
        PEA_TRY(launch_pack(pj.data(), (int)pj.size(), stream));

        return PEA_OK;
    };

    auto level_io = [&](int s, float *&T, float *&O, const float *&In, int64_t &ldIn) {
        Level &L = m->levels[(size_t)s];
        T = wsf + L.off_t;
        O = wsf + L.off_o;
        In = s == 0 ? x : wsf + m->levels[(size_t)s - 1].off_o;
        ldIn = s == 0 ? ldx : m->levels[(size_t)s - 1].ld_o;
    };

    // neighbour aggregation of level s
    auto run_groups = [&](int s, AggMode mode) -> int {
        Level &L = m->levels[(size_t)s];
        float *T, *O;
        const float *In;
        int64_t ldIn;
        level_io(s, T, O, In, ldIn);
        std::vector<AggGroup> gs;
        for (const GroupPlan &g : L.groups) {
            Relation &R = plan->rels[(size_t)g.rel];
            const bool via_slots = sharded && s > 0;  // gather sources arrive through the exchange buffer
            if (via_slots) PEA_REQUIRE(R.col_slot != nullptr, PEA_ERR_ARG, "relation %d has no exchange layout (pea_plan_set_sources)", g.rel);
            AggGroup a{};
            a.rowptr = R.rowptr;
            a.col = via_slots ? R.col_slot : R.col;
            a.short_rows = R.short_rows;
            a.long_items = R.long_items;
            a.hub_rows = R.hub_rows;
            a.hub_first = R.hub_first;
            a.hub_count = R.hub_count;
            a.n_short = R.n_short;
            // rows without incoming edges of a layer that feeds another layer are not aggregated at all: the next
            // transform reads T_s for them (GemmJob::a1_mask)
            const bool skip0 = mode != AGG_MEAN && !g.last && (plan->flags & PEA_PLAN_SELF_LOOPS) && !training;
            const bool mean2 = mode == AGG_MEAN && m->sage2;  // SAGE on the GAT/GCN schedule: mean of T_s rows, added to the root term
            // ... whose rows without incoming edges already hold their final value up to the relu (mean = 0): they are
            // skipped too, and the next level's transforms apply the relu when they load them (GemmJob::a1_mask)
            const bool skip0_mean = mean2 && !g.last && !training;
            if (training && mode == AGG_GAT) {  // keep (max, denominator) per (row, head) for the backward
                a.stats = wsf + L.off_stats + 2 * g.a_k;
                a.ld_stats = L.ld_stats;
            }
            if (skip0 || skip0_mean) {
                a.short_rows = R.short_rows + R.n_short0;
                a.n_short = R.n_short - R.n_short0;
            }
            a.n_long = R.n_long;
            a.n_hub = R.n_hub;
            a.W = g.W;
            a.F = g.F;
            a.partial = partial + g.partial_off;
            a.neg_slope = d.negative_slope;
            {
                const double loops = (mode != AGG_MEAN && (plan->flags & PEA_PLAN_SELF_LOOPS)) ? 1.0 : 0.0;
                a.msgs_short = (double)R.edges_short + loops * a.n_short;
                a.msgs_long = (double)R.edges_long + loops * R.n_direct;
                a.idx_share = mode == AGG_MEAN ? 1.0 : (double)g.n_convs;
                a.table_rows = via_slots ? (double)R.slots_per_rank * plan->shard_world : (double)R.src_span;
            }
            if (mean2) {
                a.feat = via_slots ? wsf + g.xch_off : T + g.col;
                a.ld_feat = via_slots ? g.xch_ld : L.ld_t;
                a.feat_self = T + g.col;  // unused (no self loop)
                a.ld_self = L.ld_t;
                a.accum = 1;              // out already holds lin_root(x_i) + bias (run_gemm)
                if (g.last) {
                    a.out = X + g.out_col;
                    a.ld_out = (int)ldX;
                    a.relu = relu_last;
                } else {
                    a.out = O + g.out_col;
                    a.ld_out = L.ld_o;
                    a.relu = 1;
                }
            } else if (mode == AGG_MEAN) {
                a.feat = via_slots ? wsf + g.xch_off : In + g.col;
                a.ld_feat = via_slots ? g.xch_ld : (int)ldIn;
                a.feat_self = In + g.col;  // unused (no self loop)
                a.ld_self = (int)ldIn;
                a.out = T + g.out_col;
                a.ld_out = L.ld_t;
            } else {
                a.feat = via_slots ? wsf + g.xch_off : T + g.col;
                a.ld_feat = via_slots ? g.xch_ld : L.ld_t;
                a.feat_self = T + g.col;
                a.ld_self = L.ld_t;
                a.att_src = pack + L.att_src_off + g.col;
                a.att_dst = pack + L.att_dst_off + g.col;
                a.bias = pack + g.bias_off;
                a.self_loop = (plan->flags & PEA_PLAN_SELF_LOOPS) ? 1 : 0;
                if (g.last) {
                    a.out = X + g.out_col;
                    a.ld_out = (int)ldX;
                    a.relu = relu_last;
                } else {
                    a.out = O + g.out_col;
                    a.ld_out = L.ld_o;
                    a.relu = 1;
                }
                if (mode == AGG_GCN) {
                    const bool fc = d.gcn_deg_from_col != 0;
                    PEA_TRY(ensure_dinv(plan, g.rel, fc, stream));
                    a.dinv_self = fc ? R.dinv_col : R.dinv_row;
                    a.dinv = a.dinv_self;
                    if (via_slots) {
                        PEA_TRY(ensure_dinv_slots(plan, g.rel, fc, stream));
                        a.dinv = fc ? R.dinv_col_slot : R.dinv_row_slot;
                    }
                }
            }
            // LDS image of the relation's most frequent sources for the long-row kernel (item popularity is Zipf-like:
            // a few hundred item rows serve a large share of the item -> user messages).  OFF unless PEA_HOT=1: measured
            // slower than the plain kernel on the 25m-shaped graph, kept for the record
            // and for graphs with heavier skew.
            {
                const char *hot_env = getenv("PEA_HOT");   // read per forward: tests and A/B runs flip it inside one process
                const bool hot_on = hot_env && atoi(hot_env) != 0;
                if (hot_on && R.n_long > 0 && a.W >= 16) {
                    const int K = std::min(1024, (160 * 1024 - 2048) / (4 * a.W + 4)) & ~7;
                    const HotVariant *hv = nullptr;
                    PEA_TRY(ensure_hot(plan, g.rel, K, via_slots, stream, &hv));
                    if (hv) {
                        a.hot_col = hv->col;
                        a.hot_nodes = hv->nodes;
                        a.hot_K = hv->K;
                        a.hot_frac = R.e_kept > 0 ? (double)hv->hot_edges / (double)R.e_kept : 0.0;
                    }
                }
            }
            gs.push_back(a);
        }
        for (size_t b = 0; b < gs.size(); b += kMaxAggGroups)
            PEA_TRY(launch_aggregate(mode, gs.data() + b, (int)std::min<size_t>(kMaxAggGroups, gs.size() - b), stream));
        return PEA_OK;
    };

    // dense transform of level s (GAT/GCN: before the aggregation; SAGE: after it)
    auto run_gemm = [&](int s) -> int {
        Level &L = m->levels[(size_t)s];
        float *T, *O;
        const float *In;
        int64_t ldIn;
        level_io(s, T, O, In, ldIn);
        // SAGE on the GAT/GCN schedule: besides T_s = In W_rel^T (the gather source, built like GCN's below) every unit
        // gets its root term  In W_root^T + bias  written where the aggregation will add the neighbour mean
        auto push_root_jobs = [&](std::vector<GemmJob> &jobs) {
            if (!m->sage2) return;
            for (const Unit &u : L.units) {
                GemmJob J{};
                J.A1 = In + u.in_col;
                J.lda1 = (int)ldIn;
                J.K1 = u.in_w;
                J.B = pack + u.root_off;
                J.ldb = u.ld_root;
                J.n_out = u.ld_root;
                J.bias = pack + u.bias_off;
                J.n_seg = 1;
                J.seg[0].c0 = 0;
                J.seg[0].c1 = u.HF;
                J.seg[0].dst = u.last ? X + u.o_col : O + u.o_col;
                J.seg[0].ld = u.last ? (int)ldX : L.ld_o;
                J.seg[0].relu = 0;          // relu comes after the mean has been added (finish_row)
                if (s > 0 && !training) {   // see the per-unit jobs below: edge-less rows of the previous layer, relu on load
                    for (const Unit &up : m->levels[(size_t)s - 1].units) {
                        if (up.p != u.p) continue;
                        J.a1_mask = plan->rels[(size_t)up.rel].deg0;
                        J.a1_alt = J.A1;
                        J.lda_alt = J.lda1;
                    }
                }
                if (sharded) {
                    J.rows = own_rows;
                    J.n_rows = n_own;
                }
                jobs.push_back(J);
            }
        };
        if (kind == PEA_KIND_SAGE && !m->sage2) {
            std::vector<GemmJob> jobs;
            for (const Unit &u : L.units) {
                GemmJob J{};
                J.A1 = T + u.t_col;
                J.lda1 = L.ld_t;
                J.K1 = u.in_w;
                J.A2 = In + u.in_col;
                J.lda2 = (int)ldIn;
                J.K2 = u.in_w;
                J.B = pack + u.b_off;
                J.ldb = u.ldb;
                J.n_out = u.ldb;
                J.bias = pack + u.bias_off;
                J.n_seg = 1;
                J.seg[0].c0 = 0;
                J.seg[0].c1 = u.HF;
                if (u.last) {
                    J.seg[0].dst = X + u.o_col;
                    J.seg[0].ld = (int)ldX;
                    J.seg[0].relu = relu_last;
                } else {
                    J.seg[0].dst = O + u.o_col;
                    J.seg[0].ld = L.ld_o;
                    J.seg[0].relu = 1;
                }
                jobs.push_back(J);
            }
            return launch_gemm_batch(jobs.data(), (int)jobs.size(), own_rows, n_own, stream);
        }
        if (L.shared_input && !sharded) {
            GemmJob J{};
            J.A1 = In;
            J.lda1 = (int)ldIn;
            J.K1 = d.emb_dim;
            J.B = pack + L.b_off;
            J.ldb = L.ldb;
            J.n_out = L.n_out;
            J.n_seg = 1;
            J.seg[0].c0 = 0;
            J.seg[0].c1 = L.n_cols;
            J.seg[0].dst = T;
            J.seg[0].ld = L.ld_t;
            J.no_narrow = 1;  // same kernel family as the per-relation jobs of a sharded plan (bit-identical results)
            PEA_TRY(launch_gemm(J, nullptr, N, stream));
            std::vector<GemmJob> roots;
            push_root_jobs(roots);
            return launch_gemm_batch(roots.data(), (int)roots.size(), nullptr, N, stream);
        }
        if (L.shared_input) {
            // sharded level 0: x is replicated, so each rank transforms, per relation, exactly the rows it will
            // read: its own rows plus that relation's source nodes (plan need_rows)
            std::vector<GemmJob> rel_jobs;
            size_t i = 0;
            while (i < L.units.size()) {
                size_t j = i;
                const Relation &R = plan->rels[(size_t)L.units[i].rel];
                // one job per run of channels whose relations read the same row list (same relation, or relations
                // the host gave one shared list: own rows + the union of their few source nodes) and whose columns are
                // contiguous: wider jobs reuse the input fragment over more column tiles
                while (j < L.units.size()) {
                    const Relation &Rj = plan->rels[(size_t)L.units[j].rel];
                    const bool same_rows = L.units[j].rel == L.units[i].rel || (Rj.n_need == R.n_need && Rj.need_hash == R.need_hash);
                    const bool contiguous = j == i || L.units[j].t_col == L.units[j - 1].t_col + L.units[j - 1].HF;
                    if (!same_rows || !contiguous) break;
                    ++j;
                }
                PEA_REQUIRE(R.need_rows != nullptr || R.n_need == 0, PEA_ERR_ARG, "relation %d has no need_rows (pea_plan_set_sources)", L.units[i].rel);
                const int c_beg = L.units[i].t_col, c_end = L.units[j - 1].t_col + L.units[j - 1].HF;
                GemmJob J{};
                J.A1 = In;
                J.lda1 = (int)ldIn;
                J.K1 = d.emb_dim;
                J.B = pack + L.b_off + c_beg;
                J.ldb = L.ldb;
                J.n_out = c_end - c_beg;
                J.n_seg = 1;
                J.seg[0].c0 = 0;
                J.seg[0].c1 = c_end - c_beg;
                J.seg[0].dst = T + c_beg;
                J.seg[0].ld = L.ld_t;
                J.no_narrow = 1;
                J.rows = R.need_rows;       // one launch for all relations, each job with its own row list
                J.n_rows = R.n_need;
                rel_jobs.push_back(J);
                i = j;
            }
            push_root_jobs(rel_jobs);
            PEA_TRY(launch_gemm_batch(rel_jobs.data(), (int)rel_jobs.size(), nullptr, 0, stream));
            return PEA_OK;
        }
        std::vector<GemmJob> jobs;
        Level &Lp = m->levels[(size_t)s - 1];
        for (const Unit &u : L.units) {
            GemmJob J{};
            J.A1 = In + u.in_col;
            J.lda1 = (int)ldIn;
            J.K1 = u.in_w;
            J.B = pack + u.b_off;
            J.ldb = u.ldb;
            J.n_out = u.ldb;
            J.n_seg = 1;
            J.seg[0].c0 = 0;
            J.seg[0].c1 = u.HF;
            J.seg[0].dst = T + u.t_col;
            J.seg[0].ld = L.ld_t;
            if (m->sage2 && !training) {  // edge-less rows of the previous layer hold root + bias: relu on load
                for (const Unit &up : Lp.units) {
                    if (up.p != u.p) continue;
                    J.a1_mask = plan->rels[(size_t)up.rel].deg0;
                    J.a1_alt = J.A1;
                    J.lda_alt = J.lda1;
                }
            }
            if ((plan->flags & PEA_PLAN_SELF_LOOPS) && !training) {  // edge-less rows of the previous layer: read T_{s-1} (see run_groups)
                for (const Unit &up : Lp.units) {
                    if (up.p != u.p) continue;
                    Relation &Rp = plan->rels[(size_t)up.rel];
                    J.a1_mask = Rp.deg0;
                    J.a1_alt = wsf + Lp.off_t + up.t_col;
                    J.lda_alt = Lp.ld_t;
                    J.a1_bias = pack + Lp.bias_off + up.t_col;
                    if (kind == PEA_KIND_GCN) {
                        const bool fc = d.gcn_deg_from_col != 0;
                        PEA_TRY(ensure_dinv(plan, up.rel, fc, stream));
                        J.a1_scale = fc ? Rp.dinv_col : Rp.dinv_row;
                    }
                }
            }
            jobs.push_back(J);
        }
        push_root_jobs(jobs);
        return launch_gemm_batch(jobs.data(), (int)jobs.size(), own_rows, n_own, stream);
    };

    // ---- two-step inference schedule, stage 0: aggregate x per channel, then both transforms in one kernel -> T_1
    // `part` (sharded ranks, pea_model_forward_part): PEA_PART_SOURCES = weight packing, the first-layer aggregation and the
    // transforms of the first n_owned_first owned rows (the ones other ranks gather from: written into this rank's block of
    // the exchange buffers as well, so the host can start the all-gather); PEA_PART_REST = the transforms of the other
    // owned rows (nobody else reads them: they run behind the all-gather); PEA_PART_ALL = both.
    auto run_fused2_stage0 = [&]() -> int {
        Level &L0 = m->levels[0], &L1 = m->levels[1];
        const int64_t n_first = (sharded && plan->n_owned_first >= 0) ? std::min<int64_t>(plan->n_owned_first, n_own) : n_own;
        const bool do_first = part != PEA_PART_REST, do_rest = part != PEA_PART_SOURCES;
        auto run_mlp2 = [&](const Mlp2Launch &ML) -> int {
            if (part == PEA_PART_ALL) return launch_mlp2(ML, own_rows, n_own, stream);
            if (do_first) return launch_mlp2(ML, own_rows, n_first, stream);
            return launch_mlp2(ML, own_rows + n_first, n_own - n_first, stream);
        };
        // sharded: a channel's T_1 rows that are gather sources of layer 2 also go into the exchange buffer of its group
        auto set_exchange = [&](Mlp2Chan &C, const Unit &u1) -> int {
            C.x_slot = nullptr;
            C.x_buf = nullptr;
            C.x_ld = 0;
            if (!sharded) return PEA_OK;
            for (const GroupPlan &g : L1.groups) {
                if (u1.t_col < g.col || u1.t_col >= g.col + g.W) continue;
                const Relation &R2 = plan->rels[(size_t)g.rel];
                PEA_REQUIRE(R2.slot_of_node != nullptr && g.xch_ld > 0, PEA_ERR_ARG,
                            "relation %d has no exchange layout (pea_plan_set_sources)", g.rel);
                C.x_slot = R2.slot_of_node;
                C.x_buf = wsf + g.xch_off + (u1.t_col - g.col);
                C.x_ld = g.xch_ld;
                return PEA_OK;
            }
            PEA_REQUIRE(false, PEA_ERR_ARG, "fused schedule: channel %d has no layer-2 group", u1.p);
        };
        float *A0 = wsf + L0.off_t;
        Mlp2Launch ML{};
        ML.kind = kind;
        ML.n = (int)L0.units.size();
        ML.emb = d.emb_dim;
        ML.hid = d.hidden_size;
        ML.out = d.repr_dim;
        ML.x = x;
        ML.ldx = ldx;
        ML.a0 = A0;
        ML.ld_a0 = m->ld_a0;
        ML.t1 = wsf + L1.off_t;
        ML.ld_t1 = L1.ld_t;
        ML.images = pack + m->mlp2_img_off;
        if (training) {   // fused2_train: keep H (= O_0) for the backward
            ML.h0 = wsf + L0.off_o;
            ML.ld_h0 = L0.ld_o;
            m->last_x = x;
            m->last_ldx = ldx;
        }
        ML.bias1 = pack + L1.bias_off;
        ML.att_src1 = kind == PEA_KIND_GAT ? pack + L1.att_src_off : nullptr;
        ML.att_dst1 = kind == PEA_KIND_GAT ? pack + L1.att_dst_off : nullptr;
        const bool fc = d.gcn_deg_from_col != 0;
        std::vector<AggGroup> gs;
        size_t part_off = 0;
        if (kind == PEA_KIND_SAGE) {
            // one mean of x rows per distinct first relation (units are sorted by relation), shared by its channels; rows
            // without incoming edges are not visited (their mean is 0: mlp2 feeds zeros)
            ML.r1 = X;
            ML.ld_r1 = ldX;
            int n_rel = 0, prev_rel = -1;
            for (size_t ui = 0; ui < L0.units.size(); ++ui) {
                const Unit &u = L0.units[ui];
                const Unit *u1 = nullptr;
                for (const Unit &c : L1.units)
                    if (c.p == u.p) u1 = &c;
                PEA_REQUIRE(u1 != nullptr, PEA_ERR_ARG, "fused schedule: channel %d has no second layer", u.p);
                Relation &R = plan->rels[(size_t)u.rel];
                if (u.rel != prev_rel) {
                    prev_rel = u.rel;
                    AggGroup a{};
                    a.rowptr = R.rowptr;
                    a.col = R.col;
                    a.short_rows = R.short_rows + R.n_short0;
                    a.n_short = R.n_short - R.n_short0;
                    a.long_items = R.long_items;
                    a.n_long = R.n_long;
                    a.hub_rows = R.hub_rows;
                    a.hub_first = R.hub_first;
                    a.hub_count = R.hub_count;
                    a.n_hub = R.n_hub;
                    a.W = d.emb_dim;
                    a.F = d.emb_dim;
                    a.feat = x;
                    a.ld_feat = (int)ldx;
                    a.feat_self = x;
                    a.ld_self = (int)ldx;
                    a.out = A0 + (size_t)n_rel * d.emb_dim;
                    a.ld_out = m->ld_a0;
                    a.partial = partial + part_off;
                    part_off += (size_t)slots_of(m, u.rel) * partial_record_floats(d.emb_dim, d.emb_dim);
                    a.msgs_short = (double)R.edges_short;
                    a.msgs_long = (double)R.edges_long;
                    a.idx_share = 1.0;
                    a.table_rows = (double)R.src_span;
                    gs.push_back(a);
                    ++n_rel;
                }
                Mlp2Chan &C = ML.c[ui];
                C.w0 = param(u, 0);          // lin_rel.weight
                C.b0 = param(u, 1);          // lin_rel.bias
                C.w0_root = param(u, 2);     // lin_root.weight
                C.w1 = param(*u1, 0);
                C.b1 = param(*u1, 1);
                C.w1_root = param(*u1, 2);
                PEA_REQUIRE(C.w0 && C.w1 && C.w0_root && C.w1_root, PEA_ERR_ARG, "forward: null weight pointer (channel %d)", u.p);
                C.a0_col = (n_rel - 1) * d.emb_dim;
                C.t1_col = u1->t_col;
                C.r1_col = u1->o_col;
                C.h0_col = u.t_col;
                C.deg0 = R.deg0;
                PEA_TRY(set_exchange(C, *u1));
            }
            if (do_first) {
                PEA_TRY(launch_mlp2_pack(ML, stream));
                for (size_t b = 0; b < gs.size(); b += kMaxAggGroups)
                    PEA_TRY(launch_aggregate(AGG_MEAN, gs.data() + b, (int)std::min<size_t>(kMaxAggGroups, gs.size() - b), stream));
            }
            return run_mlp2(ML);
        }
        for (size_t ui = 0; ui < L0.units.size(); ++ui) {
            const Unit &u = L0.units[ui];
            const Unit *u1 = nullptr;
            for (const Unit &c : L1.units)
                if (c.p == u.p) u1 = &c;
            PEA_REQUIRE(u1 != nullptr, PEA_ERR_ARG, "fused schedule: channel %d has no second layer", u.p);
            Relation &R = plan->rels[(size_t)u.rel];
            Mlp2Chan &C = ML.c[ui];
            C.w0 = param(u, 0);
            C.w1 = param(*u1, 0);
            PEA_REQUIRE(C.w0 && C.w1, PEA_ERR_ARG, "forward: null weight pointer (channel %d)", u.p);
            C.ws = pack + m->mlp2_att_off + (size_t)2 * ui * d.emb_dim;
            C.wd = C.ws + d.emb_dim;
            C.a0_col = (int)ui * d.emb_dim;
            C.t1_col = u1->t_col;
            C.h0_col = u.t_col;
            C.deg0 = R.deg0;
            PEA_TRY(set_exchange(C, *u1));
            if (kind == PEA_KIND_GAT) {
                C.att_dst0 = param(u, 1);   // att_i multiplies the TARGET row
                C.att_src0 = param(u, 2);   // att_j multiplies the SOURCE row
                C.b0 = param(u, 3);
                C.att_dst1 = param(*u1, 1);
                C.att_src1 = param(*u1, 2);
                C.b1 = param(*u1, 3);
                PEA_REQUIRE(C.att_src0 && C.att_dst0 && C.att_src1 && C.att_dst1, PEA_ERR_ARG,
                            "forward: null att_i/att_j (channel %d)", u.p);
            } else {
                C.b0 = param(u, 1);
                C.b1 = param(*u1, 1);
                PEA_TRY(ensure_dinv(plan, u.rel, fc, stream));
                C.dinv = fc ? R.dinv_col : R.dinv_row;
            }
            // the channel's first-layer aggregation of x: rows without incoming edges are not visited (mlp2 reads x)
            AggGroup a{};
            a.rowptr = R.rowptr;
            a.col = R.col;
            a.short_rows = R.short_rows + R.n_short0;
            a.n_short = R.n_short - R.n_short0;
            a.long_items = R.long_items;
            a.n_long = R.n_long;
            a.hub_rows = R.hub_rows;
            a.hub_first = R.hub_first;
            a.hub_count = R.hub_count;
            a.n_hub = R.n_hub;
            a.W = d.emb_dim;
            a.F = d.emb_dim;
            a.feat = x;
            a.ld_feat = (int)ldx;
            a.feat_self = x;
            a.ld_self = (int)ldx;
            a.att_src = C.ws;
            a.att_dst = C.wd;
            a.out = A0 + C.a0_col;
            a.ld_out = m->ld_a0;
            if (training && kind == PEA_KIND_GAT) {   // (max, denominator) per (row, channel) for the x-space backward
                a.stats = wsf + L0.off_stats + 2 * (int)ui;
                a.ld_stats = L0.ld_stats;
            }
            a.self_loop = 1;
            a.neg_slope = d.negative_slope;
            a.partial = partial + part_off;
            part_off += (size_t)slots_of(m, u.rel) * partial_record_floats(d.emb_dim, d.emb_dim);
            a.dinv = C.dinv;
            a.dinv_self = C.dinv;
            a.msgs_short = (double)R.edges_short + a.n_short;
            a.msgs_long = (double)R.edges_long + R.n_direct;
            a.idx_share = 1.0;
            a.table_rows = (double)R.src_span;
            gs.push_back(a);
        }
        if (do_first) {
            PEA_TRY(launch_mlp2_pack(ML, stream));
            const AggMode mode = kind == PEA_KIND_GAT ? AGG_GAT : AGG_GCN;
            for (size_t b = 0; b < gs.size(); b += kMaxAggGroups)
                PEA_TRY(launch_aggregate(mode, gs.data() + b, (int)std::min<size_t>(kMaxAggGroups, gs.size() - b), stream));
        }
        return run_mlp2(ML);
    };

    // Stage k = the work between two exchanges of gather sources (all stages back to back when not sharded):
    //   GAT/GCN: [k == 0: pack, transform_0]  aggregate_k  [transform_{k+1}]        SAGE: [pack]  aggregate_k  transform_k
    // After stage k < last, the gather source of level k+1 is complete on its owner rows (T_{k+1} resp. O_k).
    const int s_beg = stage < 0 ? 0 : stage, s_end = stage < 0 ? n_levels : stage + 1;
    PEA_REQUIRE(s_beg >= 0 && s_end <= n_levels, PEA_ERR_ARG, "forward: stage %d of %d", stage, n_levels);
    PEA_REQUIRE(part == PEA_PART_ALL || (sharded && stage >= 0), PEA_ERR_ARG, "forward: parts are stages of a sharded plan");
    const bool splits = m->fused2 && !training;   // only stage 0 of the two-step schedule has a part nobody else reads
    const bool two_step = (m->fused2 && !training) || (m->fused2_train && training);
    for (int k = s_beg; k < s_end; ++k) {
        if (part == PEA_PART_REST && !(splits && k == 0)) continue;   // everything ran with PEA_PART_SOURCES
        // the two-step schedule packs everything it reads in its own launch (launch_mlp2_pack)
        if (k == 0 && !two_step) PEA_TRY(pack_weights());
        if (two_step) {
            if (k == 0) PEA_TRY(run_fused2_stage0());
            else PEA_TRY(run_groups(k, kind == PEA_KIND_GAT ? AGG_GAT : kind == PEA_KIND_GCN ? AGG_GCN : AGG_MEAN));
        } else if (kind == PEA_KIND_SAGE && !m->sage2) {
            PEA_TRY(run_groups(k, AGG_MEAN));
            PEA_TRY(run_gemm(k));
        } else {
            if (k == 0) PEA_TRY(run_gemm(0));
            PEA_TRY(run_groups(k, kind == PEA_KIND_GAT ? AGG_GAT : kind == PEA_KIND_GCN ? AGG_GCN : AGG_MEAN));
            if (k + 1 < n_levels) PEA_TRY(run_gemm(k + 1));
        }
        if (k == n_levels - 1 && (out_repr || out_stack))
            PEA_TRY(launch_fuse(N, d.num_channels, d.repr_dim, X, ldX, m->x_col, att, masked, d.fuse_mode, own_rows, n_own,
                                out_repr, out_stack, stream, sel));
    }
    return PEA_OK;
}

}  // namespace pea

// ---------------------------------------------------------------------------------------------- C ABI
using namespace pea;

extern "C" int pea_model_create(const pea_plan *plan, const pea_model_desc *desc, pea_model **out) {
    PEA_REQUIRE(out != nullptr, PEA_ERR_ARG, "model: out is null");
    *out = nullptr;
    pea_model *m = new pea_model();
    const int rc = init_model(m, plan, desc);
    if (rc != PEA_OK) {
        delete m;
        return rc;
    }
    *out = m;
    return PEA_OK;
}

extern "C" int pea_model_destroy(pea_model *model) {
    if (model) (void)hipFree(model->active_bits);
    delete model;
    return PEA_OK;
}

extern "C" size_t pea_model_workspace_bytes(const pea_model *model) {
    return model ? model->total_floats * sizeof(float) + 256 : 0;
}

extern "C" int pea_model_params_per_layer(const pea_model *model) { return model ? model->n_slots_per_layer : 0; }

extern "C" int pea_model_stats(const pea_model *model, int64_t *messages, double *algorithmic_bytes) {
    PEA_REQUIRE(model, PEA_ERR_ARG, "model stats: null model");
    if (messages) *messages = model->messages;
    if (algorithmic_bytes) *algorithmic_bytes = model->alg_bytes;
    return PEA_OK;
}

extern "C" double pea_model_compulsory_bytes(const pea_model *model) { return model ? model->compulsory_bytes : 0.0; }

float *aligned_ws(void *workspace) {
    return reinterpret_cast<float *>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
}

extern "C" int pea_model_forward(pea_model *model, const float *const *params_host, const float *x, const float *att,
                                 int masked_channel, void *workspace, size_t workspace_bytes, float *out_repr,
                                 float *out_stack, void *stream) {
    PEA_REQUIRE(model && params_host && x && workspace, PEA_ERR_ARG, "forward: null argument");
    PEA_REQUIRE(workspace_bytes >= pea_model_workspace_bytes(model), PEA_ERR_NOMEM, "forward: workspace %zu < %zu bytes",
                workspace_bytes, pea_model_workspace_bytes(model));
    PEA_REQUIRE(masked_channel >= -1 && masked_channel < model->d.num_channels, PEA_ERR_ARG, "forward: masked channel %d", masked_channel);
    PEA_REQUIRE(out_repr || out_stack, PEA_ERR_ARG, "forward: no output requested");
    PEA_REQUIRE(model->d.fuse_mode == PEA_FUSE_MEAN || att || !out_repr, PEA_ERR_ARG, "forward: att is required for 'att' fusion");
    return model_forward(model, -1, params_host, x, model->d.emb_dim, att, masked_channel, aligned_ws(workspace), out_repr,
                         out_stack, nullptr, 0, 0, (hipStream_t)stream, false);
}

extern "C" int pea_model_forward_train(pea_model *model, const float *const *params_host, const float *x, const float *att,
                                       int masked_channel, void *workspace, size_t workspace_bytes, float *out_repr,
                                       float *out_stack, void *stream) {
    PEA_REQUIRE(model && params_host && x && workspace, PEA_ERR_ARG, "forward_train: null argument");
    PEA_REQUIRE(model->backward, PEA_ERR_ARG, "forward_train: the model was created without enable_backward");
    PEA_REQUIRE(workspace_bytes >= pea_model_workspace_bytes(model), PEA_ERR_NOMEM, "forward_train: workspace too small");
    PEA_REQUIRE(masked_channel >= -1 && masked_channel < model->d.num_channels, PEA_ERR_ARG, "forward_train: masked channel");
    PEA_REQUIRE(out_repr || out_stack, PEA_ERR_ARG, "forward_train: no output requested");
    return model_forward(model, -1, params_host, x, model->d.emb_dim, att, masked_channel, aligned_ws(workspace), out_repr,
                         out_stack, nullptr, 0, 0, (hipStream_t)stream, true);
}

extern "C" int pea_model_num_stages(const pea_model *model) { return model ? (int)model->levels.size() : 0; }

extern "C" int pea_model_forward_stage(pea_model *model, int stage, const float *const *params_host, const float *x,
                                       const float *att, int masked_channel, void *workspace, size_t workspace_bytes,
                                       float *out_repr, float *out_stack, void *stream) {
    PEA_REQUIRE(model && params_host && x && workspace, PEA_ERR_ARG, "forward_stage: null argument");
    PEA_REQUIRE(stage >= 0 && stage < (int)model->levels.size(), PEA_ERR_ARG, "forward_stage: stage %d of %d", stage,
                (int)model->levels.size());
    PEA_REQUIRE(workspace_bytes >= pea_model_workspace_bytes(model), PEA_ERR_NOMEM, "forward_stage: workspace too small");
    PEA_REQUIRE(masked_channel >= -1 && masked_channel < model->d.num_channels, PEA_ERR_ARG, "forward_stage: masked channel %d", masked_channel);
    return model_forward(model, stage, params_host, x, model->d.emb_dim, att, masked_channel, aligned_ws(workspace), out_repr,
                         out_stack, nullptr, 0, 0, (hipStream_t)stream, false);
}

// One PART of a stage (see run_fused2_stage0) + the optional batch-row selection in the last stage's fusion launch.
extern "C" int pea_model_forward_part(pea_model *model, int stage, const pea_stage_opts *opts, const float *const *params_host,
                                      const float *x, const float *att, int masked_channel, void *workspace,
                                      size_t workspace_bytes, float *out_repr, float *out_stack, void *stream) {
    PEA_REQUIRE(model && opts && params_host && x && workspace, PEA_ERR_ARG, "forward_part: null argument");
    PEA_REQUIRE(stage >= 0 && stage < (int)model->levels.size(), PEA_ERR_ARG, "forward_part: stage %d of %d", stage,
                (int)model->levels.size());
    PEA_REQUIRE(opts->part == PEA_PART_ALL || opts->part == PEA_PART_SOURCES || opts->part == PEA_PART_REST, PEA_ERR_ARG,
                "forward_part: part %d", opts->part);
    PEA_REQUIRE(workspace_bytes >= pea_model_workspace_bytes(model), PEA_ERR_NOMEM, "forward_part: workspace too small");
    PEA_REQUIRE(masked_channel >= -1 && masked_channel < model->d.num_channels, PEA_ERR_ARG, "forward_part: masked channel %d", masked_channel);
    FuseSelect sel;
    const bool last = stage == (int)model->levels.size() - 1;
    if (opts->n_sel > 0) {
        PEA_REQUIRE(last && opts->part != PEA_PART_REST, PEA_ERR_ARG, "forward_part: batch rows are selected by the last stage");
        PEA_REQUIRE(opts->sel_ids && opts->sel_out && opts->err_flag && out_repr, PEA_ERR_ARG, "forward_part: selection needs ids, output, flag, table");
        sel.ids = opts->sel_ids;
        sel.id_stride = opts->sel_stride > 0 ? opts->sel_stride : 1;
        sel.n = opts->n_sel;
        sel.out = opts->sel_out;
        sel.err = opts->err_flag;
        sel.rank = model->plan->shard_rank;
        sel.world = model->plan->shard_world;
        sel.tile = model->plan->shard_tile;
    }
    return model_forward(model, stage, params_host, x, model->d.emb_dim, att, masked_channel, aligned_ws(workspace), out_repr,
                         out_stack, nullptr, 0, 0, (hipStream_t)stream, false, opts->part, opts->n_sel > 0 ? &sel : nullptr);
}

// 1 when stage `stage` writes this rank's rows of the NEXT level's exchange buffers itself (the two-step schedule's fused
// transform does): the host then only runs the all-gather; 0: the host packs them from the source table first.
extern "C" int pea_model_stage_fills_exchange(const pea_model *model, int stage) {
    return model && model->plan->shard_world > 1 && model->fused2 && stage == 0 ? 1 : 0;
}

// One stage of the TRAINING forward of a sharded model (keeps the softmax statistics and every level buffer the backward
// reads; no edge-less-row shortcuts): same stage / exchange protocol as pea_model_forward_stage.
extern "C" int pea_model_forward_stage_train(pea_model *model, int stage, const float *const *params_host, const float *x,
                                             const float *att, int masked_channel, void *workspace, size_t workspace_bytes,
                                             float *out_repr, float *out_stack, void *stream) {
    PEA_REQUIRE(model && params_host && x && workspace, PEA_ERR_ARG, "forward_stage_train: null argument");
    PEA_REQUIRE(model->backward, PEA_ERR_ARG, "forward_stage_train: the model was created without enable_backward");
    PEA_REQUIRE(stage >= 0 && stage < (int)model->levels.size(), PEA_ERR_ARG, "forward_stage_train: stage %d of %d", stage,
                (int)model->levels.size());
    PEA_REQUIRE(workspace_bytes >= pea_model_workspace_bytes(model), PEA_ERR_NOMEM, "forward_stage_train: workspace too small");
    PEA_REQUIRE(masked_channel >= -1 && masked_channel < model->d.num_channels, PEA_ERR_ARG, "forward_stage_train: masked channel %d", masked_channel);
    return model_forward(model, stage, params_host, x, model->d.emb_dim, att, masked_channel, aligned_ws(workspace), out_repr,
                         out_stack, nullptr, 0, 0, (hipStream_t)stream, true);
}

extern "C" int pea_model_num_exchanges(const pea_model *model, int level) {
    if (!model || level < 1 || level >= (int)model->levels.size() || model->plan->shard_world <= 1) return 0;
    return (int)model->levels[(size_t)level].groups.size();
}

extern "C" int pea_model_exchange_desc(const pea_model *model, int level, int k, pea_exchange_desc *out) {
    PEA_REQUIRE(model && out && k >= 0 && k < pea_model_num_exchanges(model, level), PEA_ERR_ARG, "exchange_desc: bad argument");
    const Level &L = model->levels[(size_t)level];
    const GroupPlan &g = L.groups[(size_t)k];
    const bool sage = model->d.kind == PEA_KIND_SAGE && !model->sage2;  // sage2 gathers T_level like GAT/GCN
    out->relation = g.rel;
    out->slots_per_rank = model->plan->rels[(size_t)g.rel].slots_per_rank;
    out->width = g.W;
    out->dst_ld = g.xch_ld;
    out->dst_offset_bytes = g.xch_off * sizeof(float);
    // the gather source of level `level`: T_level (GAT/GCN) or O_{level-1} (SAGE), columns [src_col, src_col + width)
    out->src_offset_bytes = (sage ? model->levels[(size_t)level - 1].off_o : L.off_t) * sizeof(float);
    out->src_ld = sage ? model->levels[(size_t)level - 1].ld_o : L.ld_t;
    out->src_col = g.col;
    return PEA_OK;
}

// ---- single conv layers: a one-channel, one-step schedule built on the host per call ----
static int single_conv(int kind, const pea_plan *plan, int relation, int in_channels, int heads, int out_channels,
                       const float *const *params, const float *x, int64_t ldx, float slope, int deg_from_col, int relu,
                       float *out, int64_t ldo, void *workspace, size_t workspace_bytes, void *stream, size_t *bytes_only) {
    pea_model m;
    m.single_conv = true;
    pea_model_desc d{};
    const int steps[1] = {1};
    const int rel[1] = {relation};
    d.kind = kind;
    d.num_channels = 1;
    d.steps = steps;
    d.relation_of = rel;
    d.emb_dim = in_channels;
    d.hidden_size = out_channels;
    d.repr_dim = out_channels;
    d.heads = heads;
    d.fuse_mode = PEA_FUSE_MEAN;
    d.gcn_deg_from_col = deg_from_col;
    d.negative_slope = slope;
    PEA_TRY(init_model(&m, plan, &d));
    if (bytes_only) {
        *bytes_only = m.total_floats * sizeof(float) + 256;
        return PEA_OK;
    }
    PEA_REQUIRE(x && out && workspace, PEA_ERR_ARG, "conv: null argument");
    PEA_REQUIRE(ldx >= in_channels && ldx % 4 == 0 && ldo >= (int64_t)heads * out_channels && ldo % 4 == 0, PEA_ERR_ARG,
                "conv: row strides (%lld, %lld) must be multiples of 4 covering the row", (long long)ldx, (long long)ldo);
    PEA_REQUIRE(workspace_bytes >= m.total_floats * sizeof(float) + 256, PEA_ERR_NOMEM, "conv: workspace too small");
    return model_forward(&m, -1, params, x, ldx, nullptr, -1, aligned_ws(workspace), nullptr, nullptr, out, ldo, relu,
                         (hipStream_t)stream, false);
}

extern "C" int pea_gat_conv(const pea_plan *plan, int relation, int in_channels, int heads, int out_channels, const float *x,
                            int64_t ldx, const float *lin_weight, const float *att_i, const float *att_j, const float *bias,
                            float negative_slope, int relu, float *out, int64_t ldo, void *workspace, size_t workspace_bytes,
                            void *stream) {
    const float *params[4] = {lin_weight, att_i, att_j, bias};
    return single_conv(PEA_KIND_GAT, plan, relation, in_channels, heads, out_channels, params, x, ldx, negative_slope, 0, relu,
                       out, ldo, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int pea_gcn_conv(const pea_plan *plan, int relation, int in_channels, int out_channels, const float *x, int64_t ldx,
                            const float *weight, const float *bias, int deg_from_col, int relu, float *out, int64_t ldo,
                            void *workspace, size_t workspace_bytes, void *stream) {
    const float *params[2] = {weight, bias};
    return single_conv(PEA_KIND_GCN, plan, relation, in_channels, 1, out_channels, params, x, ldx, 0.f, deg_from_col, relu, out,
                       ldo, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int pea_sage_conv(const pea_plan *plan, int relation, int in_channels, int out_channels, const float *x, int64_t ldx,
                             const float *rel_weight, const float *rel_bias, const float *root_weight, int relu, float *out,
                             int64_t ldo, void *workspace, size_t workspace_bytes, void *stream) {
    const float *params[3] = {rel_weight, rel_bias, root_weight};
    return single_conv(PEA_KIND_SAGE, plan, relation, in_channels, 1, out_channels, params, x, ldx, 0.f, 0, relu, out, ldo,
                       workspace, workspace_bytes, stream, nullptr);
}

extern "C" size_t pea_conv_workspace_bytes(const pea_plan *plan, int kind, int relation, int in_channels, int heads,
                                           int out_channels) {
    size_t bytes = 0;
    const int rc = single_conv(kind, plan, relation, in_channels, heads, out_channels, nullptr, nullptr, 0, 0.2f, 0, 0, nullptr,
                               0, nullptr, 0, nullptr, &bytes);
    return rc == PEA_OK ? bytes : 0;
}
