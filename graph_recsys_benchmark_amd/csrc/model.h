// Schedule data structures of one PEA model (built by model.hip, also read by the backward in model_bwd.hip).
#pragma once
#include "common.h"

namespace pea {

struct Unit {  // one channel at one level
    int p = 0, s = 0, rel = 0;
    int in_w = 0, heads = 1, F = 0, HF = 0;
    bool last = false;
    int in_col = 0;    // column of the input block (level 0: 0 in x; else in O_{s-1})
    int t_col = 0;     // column in T_s (GAT/GCN) or of its mean block in M_s (SAGE)
    int a_k = 0;       // first attention index in A_s
    int o_col = 0;     // column in O_s (non-last) or X (last)
    size_t b_off = 0;  // packed weight block (floats from the pack base)
    int ldb = 0;
    size_t bias_off = 0;
    size_t root_off = 0;  // SAGE inference schedule: packed lin_root block [in_w][ld_root]
    int ld_root = 0;
};

struct GroupPlan {  // one aggregation group of a level
    int rel = 0;
    int col = 0, W = 0, F = 0;  // columns [col, col+W) of the gather source
    int a_k = 0;                // first attention index
    bool last = false;
    int out_col = 0;            // column in O_s / X (GAT/GCN) or M_s (SAGE)
    size_t bias_off = 0;        // packed bias (floats from the pack base), GAT/GCN
    size_t partial_off = 0;     // floats from the partial base
    int n_convs = 1;            // reference conv calls this group serves (index reads it saves)
    size_t xch_off = 0;         // sharded, level >= 1: exchange buffer [world*M rows, xch_ld] (floats from the workspace base)
    int xch_ld = 0;
};

struct Level {
    std::vector<Unit> units;  // in buffer (column) order
    std::vector<GroupPlan> groups;
    int n_cols = 0, n_heads = 0;             // sum HF, sum heads (GAT/GCN)
    int ld_t = 0, ld_a = 0, ld_o = 0;        // strides of T_s (or M_s), A_s, O_s
    size_t off_t = 0, off_a = 0, off_o = 0;  // float offsets in the workspace
    bool shared_input = false;               // level 0 of GAT/GCN: one concatenated GEMM job
    size_t b_off = 0, bias_off = 0;          // concatenated weight block / per-level bias block
    size_t att_src_off = 0, att_dst_off = 0; // per-level att_j / att_i rows in column order (GAT)
    int ldb = 0, n_out = 0;
    // training / backward (allocated when the model was created with enable_backward)
    int ld_stats = 0, ld_k = 0, ld_side = 0;             // strides of stats [N,2*heads], d a_* [N,heads], side [N,4*heads]
    size_t off_stats = 0, off_dt = 0, off_do = 0, off_side = 0, off_dad = 0, off_das = 0;
    size_t gatt_src_off = 0, gatt_dst_off = 0, gbias_off = 0;  // gradient rows (floats from the grad-pack base), column order
};

}  // namespace pea

struct pea_model {
    const pea_plan *plan = nullptr;
    pea_model_desc d{};
    std::vector<int> steps, relation_of;  // copies of the host arrays
    std::vector<int> chan_first;          // index of (p,0) in relation_of
    std::vector<pea::Level> levels;
    int n_slots_per_layer = 0;
    int ld_x = 0;
    pea::ChanCols x_col{};                // column of channel p in X
    size_t pack_floats = 0;               // workspace layout (floats): [pack | levels | X | partial]
    size_t off_x = 0, off_partial = 0, partial_floats = 0;
    size_t total_floats = 0;
    int64_t messages = 0;
    double alg_bytes = 0.0;
    double compulsory_bytes = 0.0;        // HBM floor of one forward: every buffer of the schedule written once and read once
    bool single_conv = false;             // pea_*_conv: any output width, X goes to the caller's buffer
    bool backward = false;                // training buffers allocated
    // SAGE without training buffers -- or training on the two-step schedule below -- runs on the GAT/GCN schedule: transform
    // first (mean_j(W x_j) = W mean_j(x_j)), so the layers gather output-width rows; the root term lin_root(x_i) + bias is
    // written first and the mean is added to it
    bool sage2 = false;
    // GAT / GCN models of 2-step channels (every reference configuration) without training buffers run the TWO-STEP
    // INFERENCE SCHEDULE: the first layer aggregates x itself (logits from x . (W^T att)), and ONE kernel applies both
    // layers' transforms to the aggregate (csrc/mlp2.hip): T_0 / O_0 are never materialised.  PEA_FUSED2=0: level-wise.
    bool fused2 = false;
    // The same schedule for TRAINING (round 3; GAT with one head, GCN, SAGE; emb == hidden 64 / 128, single GPU;
    // PEA_FUSED2_TRAIN=0: level-wise; GCN / SAGE: their linear aggregation's backward is the reverse aggregation, model_bwd.hip):
    // the first layer aggregates x (softmax statistics kept), the fused transform also stores the hidden tile H = O_0;
    // the backward then runs the first layer's softmax passes in x space
    // (gather sources: x rows and the rows of dA_0 = dZ_0 W_0) -- T_0 is never built, and the first layer's forward gathers
    // and gradient gathers read the 70 MB table x instead of nine 64-column blocks of a 630 MB one.
    bool fused2_train = false;
    // two-step training schedule, level 0: rows whose gradient dA_0 is not identically zero (flags [N], compacted ids and
    // their device-side count: csrc/rows.hip); null: every row
    const unsigned char *active0 = nullptr;
    const int *active0_list = nullptr, *active0_count = nullptr;
    const float *last_x = nullptr;        // x of the last training forward (the level-0 backward gathers its rows)
    int64_t last_ldx = 0;
    int ld_a0 = 0;                        // row stride of A_0 (first-layer aggregates of x, P * emb columns), in the T_0 region
    size_t mlp2_img_off = 0, mlp2_att_off = 0;   // floats from the pack base: weight images, x-space attention vectors
    const unsigned char *active_rows = nullptr;  // pea_model_set_active_rows: rows with a non-zero final-output gradient
    unsigned *active_bits = nullptr;             // the same as a bitmap, rebuilt by the last level's backward (owned: hipMalloc once)
    std::vector<int> reverse_of;          // relation -> index of the reversed relation in the plan (-1: absent)
    size_t off_dx = 0, off_gpack = 0, gpack_floats = 0, off_colsum = 0;
};

