// Internal declarations shared by the peahip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/peahip.h"

namespace pea {

void set_error(const char *fmt, ...);
const char *get_error();

#define PEA_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pea::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,                 \
                           hipGetErrorString(e_));                                             \
            return PEA_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define PEA_REQUIRE(cond, code, ...)                                                           \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            pea::set_error(__VA_ARGS__);                                                       \
            return (code);                                                                     \
        }                                                                                      \
    } while (0)

#define PEA_TRY(expr)                                                                          \
    do {                                                                                       \
        int rc_ = (expr);                                                                      \
        if (rc_ != PEA_OK) return rc_;                                                         \
    } while (0)

// per-launch HIP-event timing (prof.hip); a no-op unless pea_profile_enable(1) was called
bool prof_enabled();
class ProfScope {
  public:
    // units: algorithmic bytes of SURVEY.md 8(d) attributed to the launch; gathered: bytes the launch itself must pull
    // through the memory system (every message's row chunk + its source index, each once; rows served from an LDS
    // image excluded); table: footprint of the table those rows come from (what decides the cache tier)
    ProfScope(const char *name, hipStream_t stream, double units = 0.0, double gathered = 0.0, double table = 0.0);
    ~ProfScope();
  private:
    int idx_;
    hipStream_t stream_;
};

// ---------------------------------------------------------------- launch tape (tape.hip)
// Every kernel launch of the library goes through PEA_LAUNCH.  While a tape is recording on the calling thread
// (pea_tape_begin .. pea_tape_end) each launch is also stored as a closure holding its grid, block and arguments BY VALUE;
// pea_tape_replay re-issues them on a stream without re-running the host logic that built them (schedule walks,
// descriptor structs, lazy checks): one hipLaunchKernel per kernel and nothing else.  Unlike a hipGraph replay the
// kernels enter the stream exactly as eager launches do (a rank of 8 ran 0.305 ms per step from four hipGraphs against
// 0.279 ms eager; enqueueing it eagerly through ctypes took 0.26 ms of host time, a tape replay ~0.06 ms).
// The caller owns validity, as with a graph: a tape is good while every pointer it captured still means the same.
struct Tape {
    std::vector<std::function<void(hipStream_t)>> ops;
};
Tape *tape_active();
template <class F>
inline void launch_rec(F f, hipStream_t stream) {
    if (Tape *t = tape_active()) t->ops.emplace_back(f);
    f(stream);
}
#define PEA_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    pea::launch_rec([=](hipStream_t s_) { hipLaunchKernelGGL(kernel, grid, block, shmem, s_, __VA_ARGS__); }, stream)
#define PEA_MEMSET_ASYNC(ptr, value, bytes, stream) \
    pea::launch_rec([=](hipStream_t s_) { (void)hipMemsetAsync(ptr, value, bytes, s_); }, stream)

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kShortDeg = 32;      // rows with <= this many kept edges go to the row-per-subgroup kernel
constexpr int kChunk = 512;        // hub rows are cut into chunks of at most this many edges
constexpr int64_t kSliceMinEdges = 2000000;   // relations below this are never source-sliced
// target gather footprint of one source slice.  An XCD's L2 is 4 MiB, but fewer, larger slices win: on the 42 MB user
// table of the 25m-shaped graph 8 slices of 5.2 MB (one per XCD, one phase) beat 16 x 2.6 MB and 24 x 1.7 MB (0.53 vs 0.55
// vs 0.57 ms for the first-layer gather; unsliced: 0.89 ms) -- every extra slice cuts the hub rows into more segments
constexpr double kSliceBytes = 6.0e6;
constexpr int kSliceMinSegment = 32;          // only rows with >= slices*this edges are cut per slice

// One work item of the row-per-wave kernel: edges [beg, end) of `row`; slot < 0 writes the output row,
// slot >= 0 writes a partial record (hub chunk) that the merge kernel folds in chunk order.
struct LongItem {
    int row, beg, end, slot;
};

// LDS-staged hot sources of one relation (plan.hip: ensure_hot): the K most frequent gather sources get rank 0..K-1;
// `col` holds, per CSR slot, either the plain source id / exchange slot (>= 0) or -(rank + 2) for a hot source, so the
// kernel knows without a lookup whether the row sits in its workgroup's LDS image (-1 stays the "no edge" sentinel).
struct HotVariant {
    int K = 0;
    bool via_slots = false;
    int *col = nullptr;        // device [e_kept]
    int *nodes = nullptr;      // device [K]: row index (node id, or exchange slot) of each hot source, by rank
    int64_t hot_edges = 0;     // kept edges whose source is hot
};

struct Relation {
    int64_t e_in = 0;       // COO edges handed over
    int64_t e_kept = 0;     // after self-loop removal (if the plan does that)
    int slices = 1;         // > 1: edges inside a row are grouped by source slice (8 XCDs x phases)
    int slice_min = 0;
    int64_t slice_span = 1;
    int max_deg = 0;
    int64_t src_span = 0;   // max - min + 1 over the source ids of the kept edges (footprint of the gather table in rows)
    int *rowptr = nullptr;  // device [N+1]
    int *col = nullptr;     // device [e_kept] source ids, destination-sorted, stable
    int *eid = nullptr;     // device [e_kept] original COO edge index of each CSR slot (PEA_PLAN_EDGE_IDS)
    float *dinv_row = nullptr, *dinv_col = nullptr;  // device [N], GCN deg^-1/2 (lazy)
    float *invdeg = nullptr;                         // device [N], 1 / max(in-degree, 1) (SAGE backward, lazy)
    // work lists (device), restricted to the rows this rank owns
    int *short_rows = nullptr;   // owned rows with <= kShortDeg kept edges; the first n_short0 of them have none
    int n_short = 0, n_short0 = 0;
    unsigned char *deg0 = nullptr;  // device [N]: 1 where the row has no kept edge (self loop only)
    LongItem *long_items = nullptr;
    int n_long = 0, n_direct = 0;  // n_direct: items that finish their row themselves (slot == -1)
    int *hub_rows = nullptr, *hub_first = nullptr, *hub_count = nullptr;  // per hub row: first slot, #chunks
    int n_hub = 0, n_slots = 0;
    int64_t rows_owned = 0, edges_owned = 0;
    int64_t edges_short = 0, edges_long = 0;  // owned edges by kernel (short rows / long items + hub chunks)
    // multi-GPU (set by pea_plan_set_sources): source ids renamed to slots of the exchange buffer [world*M rows]
    int *col_slot = nullptr;
    int64_t slots_per_rank = 0;                        // M
    float *dinv_row_slot = nullptr, *dinv_col_slot = nullptr;  // GCN deg^-1/2 in slot order (lazy)
    int *slot_of_node = nullptr;                       // device [N], -1 = not a source
    int *need_rows = nullptr;                          // device: rows whose level-0 transform this rank computes
    int64_t n_need = 0;
    unsigned long long need_hash = 0;                  // FNV-1a of need_rows (equal lists -> one shared transform job)
    std::vector<HotVariant> hot;                       // built lazily, one per (K, col / col_slot)
};

}  // namespace pea

struct pea_plan {
    int64_t N = 0;
    int flags = 0;
    int shard_rank = 0, shard_world = 1, shard_tile = 256;
    int gather_row_bytes = 0;  // hint: bytes of one gathered source row (0 = never slice by source)
    std::vector<pea::Relation> rels;
    int max_slots = 0;  // max hub chunks over relations (sizes the partial workspace)
    float *ones = nullptr;      // device [N] of 1.0f (SAGE backward)
    int *owned_rows = nullptr;  // device, sharded plans only
    int64_t n_owned = 0;
    int64_t n_owned_first = -1; // pea_plan_set_owned_split: the first rows of owned_rows are the ones other ranks read (-1: unset)
};

namespace pea {

int ensure_dinv(pea_plan *plan, int rel, bool from_col, hipStream_t stream);
int ensure_dinv_slots(pea_plan *plan, int rel, bool from_col, hipStream_t stream);
// hot-source variant of relation `rel` for an LDS image of K rows (nullptr in *out when the relation is too small or its
// K most frequent sources carry too few of its edges to pay for the image)
int ensure_hot(pea_plan *plan, int rel, int K, bool via_slots, hipStream_t stream, const HotVariant **out);
constexpr int64_t kHotMinEdges = 1000000;   // relations below this never get an LDS image
constexpr double kHotMinFraction = 0.15;    // ... nor those whose top-K sources carry less than this share of the edges

// ---------------------------------------------------------------- aggregation (agg.hip)
// AGG_GAT_BWD_D / _S: the two gather passes of the GAT backward (agg.hip): D walks a destination row's in-edges
// (gathers T_j) and yields d a_dst; S walks a source row's out-edges over the REVERSED relation (gathers the output
// gradient rows g_i) and yields dT_j and d a_src.
// AGG_SUM_BWD_S: out_j = dinv_self_j (sum_i dinv_i feat_i + [self_loop] dinv_self_j feat_self_j) like AGG_GCN, on the
//   backward kernels' batch-sparse walk (row_active: only flagged rows i are fetched; the surviving edges of several
//   64-edge batches are queued): the last layer's reverse aggregation of GCN / SAGE, whose dX is the batch's rows only
enum AggMode { AGG_GAT = 0, AGG_GCN = 1, AGG_MEAN = 2, AGG_GAT_BWD_D = 3, AGG_GAT_BWD_S = 4, AGG_WSUM = 5, AGG_SUM_BWD_S = 6 };

// One horizontal group: C channel-heads of width F that share a relation, columns contiguous.
struct AggGroup {
    const int *rowptr;
    const int *col;
    const int *short_rows;
    const LongItem *long_items;
    const int *hub_rows, *hub_first, *hub_count;
    int n_short, n_long, n_hub;
    const float *feat;   // gather source; row col[e] at feat + col[e]*ld_feat (node ids, or exchange slots when sharded)
    const float *feat_self;  // the destination node's own row (self loop, a_dst), indexed by node id, stride ld_self
    const float *att_src;  // GAT: att_j flattened over the group's columns [W] (multiplies the SOURCE row)
    const float *att_dst;  // GAT: att_i flattened [W] (multiplies the TARGET row)
    const int *eid;          // AGG_WSUM: original edge index of each CSR slot
    const float *edge_w;     // AGG_WSUM: per-edge weight in the caller's COO order (KGAT/KGCN att_map, NGCF coefficient)
    const float *dinv;   // GCN deg^-1/2 indexed like `col`
    const float *dinv_self;  // GCN deg^-1/2 indexed by node id
    const float *bias;   // [W] or null
    float *out;          // row i at out + i*ld_out
    float *partial;      // hub partial records
    int ld_feat, ld_self, ld_out;
    int W;               // columns of this group (multiple of 4, <= 256)
    int F;               // columns per attention group (GAT), W % F == 0
    int relu;
    int self_loop;       // add the i->i message (GAT/GCN)
    float neg_slope;
    // bookkeeping for the live roofline measurement (messages reduced by the short / long launches; how many
    // reference conv calls share this group's index read)
    double msgs_short, msgs_long, idx_share;
    double table_rows;   // rows of the table the gathers of this group read (source-id span of the relation, or slots)
    // training: per (row, head) softmax statistics (m in the log2 domain, S) written by the GAT forward
    float *stats;        // [N, ld_stats], already offset to this group's first head (2 floats per head)
    int ld_stats;
    // backward passes (see AggMode): row-local inputs indexed by node id, all offset to the group's first column/head
    const float *g_self;   // D: output-gradient row of the destination (masked by relu), stride ld_g
    const float *o_self;   // D: conv output row (post-relu O_s or X), stride ld_g
    const float *side;     // S: gathered [a_dst, m, 1/(S+eps), c] per head, stride ld_side (indexed like `col`)
    float *side_out;       // D: writes the same record for its row
    float *ksum;           // D: d a_dst out / S: d a_src out, one float per head, stride ld_k
    const float *da_dst;   // S: d a_dst of the row (from the D pass), stride ld_k
    int accum;             // AGG_MEAN: add the mean to the row already in `out` (SAGE inference schedule: root term)
    // LDS-staged hot sources (long-row kernel only): hot_col replaces `col` there, hot_nodes[k] = row of `feat` held at
    // image row k, hot_K rows; hot_frac = share of the relation's edges served from the image (bookkeeping)
    const int *hot_col, *hot_nodes;
    int hot_K;
    double hot_frac;
    const unsigned char *row_active;  // optional [N]: 0 = the row's output gradient is exactly zero (D: row, S: gathered row)
    // the same flags as a bitmap (bit n of word n / 32): what the S pass tests per GATHERED row -- 20 KB for the 162 k users of
    // the 25m-shaped graph stay in a CU's L1, the byte array (one scattered byte per edge, 162 KB) does not
    const unsigned *row_active_bits;
    // S pass: [N] flags of the FORWARD relation: 1 = the row has no incoming edge there, its softmax is its self loop
    // alone (alpha = 1, d z = 0 up to rounding): the D pass skips such rows, the S pass adds g_row for the self loop
    // without a side record, and their d a_dst reads as 0 (the level's d a_dst buffer is cleared first)
    const unsigned char *deg0_self;
    int skip_long;         // host-side: the long items of this group were launched by the fat-lane kernel already
    int ld_g, ld_side, ld_k;
};

// Rows a rank owns under tile-interleaved ownership, as arithmetic: the q-th own row is
//   row(q) = ((q / tile) * world + rank) * tile + q % tile        (q in [0, n): n = own tiles * tile; rows >= N are skipped)
// world == 1: row(q) = q.  The row-wise reductions of the backward (column sums, weight gradients, relu masks) walk
// q instead of all N rows when the plan is sharded.
struct RowMap {
    int tile = 1, world = 1, rank = 0;
    int shift = -1;  // log2(tile) when the tile is a power of two (the default 256 is): shifts instead of 64-bit divisions
    int64_t n = 0;   // number of q values
    int64_t N = 0;   // rows of the tables
    // explicit row list with its length in DEVICE memory (two-step training schedule: the rows whose layer-2 input gradient
    // is not identically zero, compacted on the device -- the host never learns the count, so nothing synchronises);
    // n is then the capacity of the list
    const int *list = nullptr;
    const int *count = nullptr;
    __device__ int64_t size() const { return count ? (int64_t)*count : n; }
    __host__ __device__ int64_t row(int64_t q) const {
        if (list) return list[q];
        if (world == 1) return q;
        if (shift >= 0) return ((((q >> shift) * world) + rank) << shift) + (q & (int64_t)(tile - 1));
        return ((q / tile) * world + rank) * tile + q % tile;
    }
};
inline RowMap make_rowmap(int64_t N, int tile, int world, int rank) {
    RowMap m;
    m.N = N;
    if (world <= 1) {
        m.n = N;
        return m;
    }
    m.tile = tile;
    m.world = world;
    m.rank = rank;
    if ((tile & (tile - 1)) == 0) {
        m.shift = 0;
        while ((1 << m.shift) < tile) ++m.shift;
    }
    const int64_t tiles = (N + tile - 1) / tile;                       // all tiles
    const int64_t own_tiles = tiles > rank ? (tiles - rank + world - 1) / world : 0;
    m.n = own_tiles * tile;
    return m;
}

inline RowMap make_rowmap_list(int64_t N, const int *list, const int *count_dev, int64_t capacity) {
    RowMap m;
    m.N = N;
    m.n = capacity;
    m.list = list;
    m.count = count_dev;
    return m;
}

constexpr int kMaxAggGroups = 16;
int launch_aggregate(AggMode mode, const AggGroup *groups, int n_groups, hipStream_t stream);
size_t partial_record_floats(int W, int F);
// backward helpers (agg_bwd.hip)
constexpr int kColsumParts = 512;
int launch_gat_backward(AggMode mode, const AggGroup *groups, int n_groups, hipStream_t stream);
int launch_colsum(const RowMap &rows, int W, int F, const float *A, int lda, const float *S, int lds, float scale, float *part,
                  float *out, hipStream_t stream);
// two scaled column sums over ONE read of A (S1 == null: one); `part` holds 2 * kColsumParts * W floats
int launch_colsum2(const RowMap &rows, int W, int F, const float *A, int lda, const float *S0, const float *S1, int lds,
                   float scale, float *part, float *out0, float *out1, hipStream_t stream);
int launch_relu_mask(const RowMap &rows, int W, float *G, int ldg, const float *O, int ldo, hipStream_t stream);

// ---------------------------------------------------------------- dense transform (gemm.hip)
struct GemmSegment {   // output columns [c0, c1) of the job go to dst[row*ld + (c - c0)]
    int c0, c1;
    float *dst;
    int ld;
    int relu;
    // optional gate (same column mapping as dst): out = gate[row*ld_gate + (c - c0)] > 0 ? value : 0 -- the relu mask of
    // the layer whose output gradient this job writes (the backward's dIn = dT W), applied in the epilogue instead of by
    // a separate pass over the buffer.  Persistent kernel only, jobs of at most 64 output columns per item (see gemm.hip).
    const float *gate;
    int ld_gate;
};
constexpr int kMaxSegments = 4;
struct GemmJob {
    // Rows flagged in a1_mask take their A1 block from a1_alt instead, as relu(scale[row] * a1_alt[row] + a1_bias):
    // a destination row with no incoming edge is exactly  conv(x)_i = h_i (+bias)  (GAT: alpha_ii = 1; GCN:
    // dinv_i^2 * h_i), so the aggregation skips it and the next layer's transform reads T_s directly.
    const unsigned char *a1_mask;
    const float *a1_alt, *a1_bias, *a1_scale;
    int lda_alt;
    const float *A1;  // [N, K1] row stride lda1
    const float *A2;  // optional second source [N, K2] (SAGE root term), K = K1 + K2
    int lda1, lda2, K1, K2;
    const float *B;   // packed [K1+K2][ldb] k-major
    int ldb;
    int n_out;        // columns (multiple of 4)
    const float *bias;  // [n_out] or null (packed alongside B)
    int n_seg;
    GemmSegment seg[kMaxSegments];
    // optional per-job row set (sharded first layer: each relation's transform covers its own row list); null = the
    // row set of the launch
    const int *rows;
    int64_t n_rows;
    // 1: never route this job to the narrow-output kernel.  The kernels differ in k summation order, so a transform must
    // pick its kernel from properties that do not depend on how jobs were cut: the first-layer (shared-input) jobs are
    // merged per relation when sharded and all together on one GPU, and both must give the same bits.
    int no_narrow;
};
int launch_gemm(const GemmJob &job, const int *rows, int64_t n_rows, hipStream_t stream);
int launch_gemm_batch(const GemmJob *jobs, int n_jobs, const int *rows, int64_t n_rows, hipStream_t stream);

// weight packing: produces the k-major extended weight blocks the GEMM consumes
struct PackJob {
    int kind;            // PEA_KIND_*
    const float *w0;     // GAT lin.weight [HF,in] | GCN weight [in,F] | SAGE lin_rel.weight [F,in]
    const float *w1;     // GAT att_i [HF]         |  -                | SAGE lin_root.weight [F,in]
    const float *w2;     // GAT att_j [HF]         |  -                |  -
    const float *w3;     // bias source [HF] (GAT bias | GCN bias | SAGE lin_rel.bias), may be null -> zeros
    float *B;            // destination block start (column offset already applied), row stride ldb
    float *bias;         // destination bias block [HF]
    float *att_src, *att_dst;  // GAT: destination blocks [HF] for att_j / att_i
    int ldb;
    int in, HF, F;       // input width, output width, width per attention group
    int zero_col, zero_n;  // padding columns (relative to B) to clear in every k row
    float *B2;           // kind PEA_PACK_SAGE2: destination of lin_root.weight^T [in][ldb2] (w0 -> B, w1 -> B2, w3 -> bias)
    int ldb2;
};
constexpr int PEA_PACK_SAGE2 = 3;
int launch_pack(const PackJob *jobs_host, int n_jobs, hipStream_t stream);

// ---------------------------------------------------------------- two-step inference schedule, dense half (mlp2.hip)
// T_1[n, c] = relu(in_c(n) . W0_c + b0_c) . W1_c for every 2-step channel c, in_c(n) = the first layer's aggregate of x
// (A_0) or, for rows without incoming edges there, x[n] itself: both transforms of a channel chained in one kernel.
constexpr int kMaxMlp2Chan = 32;   // the launch descriptor travels as a kernel argument (4 KB limit)
struct Mlp2Chan {
    const float *w0, *b0, *w1;            // first-layer weight / bias, second-layer weight (raw parameter tensors)
    const float *att_src0, *att_dst0;     // GAT first layer: att_j, att_i [hid]
    const float *b1, *att_src1, *att_dst1;  // second layer: bias (may be null), GAT att_j / att_i [out]
    const float *w0_root, *w1_root;       // SAGE: lin_root.weight of the two layers (w0 / w1 are lin_rel.weight, b0 / b1 its bias)
    int r1_col;                           // SAGE: column of the channel's root term in r1
    float *ws, *wd;                       // GAT: out: att vectors in x space, (W^T att) * log2(e)  [emb]
    const unsigned char *deg0;            // [N] 1 = no incoming edge under the channel's first relation
    const float *dinv;                    // GCN: deg^-1/2 of the first relation (node-indexed), else null
    int a0_col, t1_col;                   // column of the channel in A_0 / T_1
    // sharded: rows that are gather sources of the second layer also go straight into the exchange buffer of the
    // channel's layer-2 group (x_slot[row] >= 0: row slot of that buffer; the pack launch of round 2 is gone)
    const int *x_slot;
    float *x_buf;                         // exchange buffer + the channel's column inside the group
    int x_ld;
    int h0_col;                           // training: column of the channel's hidden block in Mlp2Launch::h0
};
struct Mlp2Launch {
    int kind, n, emb, hid, out, per_pass;
    // channel groups (what fits the LDS together) run side by side: workgroups [blk_start[g], blk_start[g + 1]) keep the
    // images of channels [g * per_pass, ...) resident and share that group's (tile, channel) items
    int n_groups;
    int blk_start[kMaxMlp2Chan + 1];
    const float *x;                       // [N, ldx]
    const float *a0;                      // [N, ld_a0]
    float *t1;                            // [N, ld_t1]
    float *images;                        // [n][mlp2_image_bytes] packed weights (written by launch_mlp2_pack)
    // the second layer's aggregation reads its bias / attention rows from the level's packed rows (column t1_col of the
    // channel); the same pack launch writes them, so the schedule needs no other weight packing
    float *bias1, *att_src1, *att_dst1;
    // SAGE: in_c(n) = [mean_j x_j | x_n] (the mean block a0_col is shared by the channels of one first relation, zero for
    // rows without incoming edges), and the second product gives T_1 = H lin_rel^T (the gather source of layer 2) AND the
    // root term H lin_root^T + bias, written to r1 where the layer-2 aggregation adds the neighbour mean
    float *r1;
    int64_t ld_r1;
    int64_t ldx, ld_a0, ld_t1;
    // training (two-step schedule with a backward): the hidden tile H = relu(in W0 + b0) is also stored -- h0 [N, ld_h0], the
    // channel's block at column h0_col.  Null: inference.  (The input rows of edge-less nodes are NOT copied into A_0: the
    // backward's weight-gradient reduction reads x for them, pea_gw_job::b_mask.)
    float *h0;
    int64_t ld_h0;
    Mlp2Chan c[kMaxMlp2Chan];
};
size_t mlp2_image_bytes(int kind, int emb, int hid);
bool mlp2_supported(int kind, int emb, int hid, int out);
int launch_mlp2_pack(const Mlp2Launch &L, hipStream_t stream);
int launch_mlp2(const Mlp2Launch &L, const int *rows, int64_t n_rows, hipStream_t stream);

// ---------------------------------------------------------------- fusion / scoring (fuse_score.hip)
constexpr int kMaxChannels = 64;
struct ChanCols {
    int c[kMaxChannels];
};
// optional batch-row selection riding in the fusion launch of a sharded rank: sel_out[k] = the fused row of node
// ids[k * id_stride] when this rank owns it (row / tile % world == rank), zeros otherwise -- what the loss all-reduce sums
struct FuseSelect {
    const int64_t *ids = nullptr;
    int64_t id_stride = 1, n = 0;
    float *out = nullptr;      // [n, R]
    int *err = nullptr;        // |= 1 when an id is outside [0, N)
    int rank = 0, world = 1, tile = 1;
};
int launch_fuse(int64_t N, int P, int R, const float *stack, int64_t ld, const ChanCols &col_of_channel,
                const float *att, int masked, int mode, const int *rows, int64_t n_rows, float *out,
                float *out_stack, hipStream_t stream, const FuseSelect *sel = nullptr);

int model_forward(pea_model *m, int stage, const float *const *params, const float *x, int64_t ldx, const float *att,
                  int masked, float *wsf, float *out_repr, float *out_stack, float *out_x, int64_t ld_out_x,
                  int relu_last, hipStream_t stream, bool training, int part = 0, const FuseSelect *sel = nullptr);

}  // namespace pea
float *aligned_ws(void *workspace);
namespace pea {

}  // namespace pea
