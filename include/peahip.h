/*
 * peahip -- C-ABI of the MI355X (gfx950) metapath-GNN aggregation path.
 *
 * The reference (ecml-peagnn/graph_recsys_benchmark) has no FFI of its own: its hot path is a
 * Python class surface that bottoms out in torch-geometric 1.5.0 / torch-scatter 2.0.5 ops.  This
 * header is what a maintainer binds (ctypes) to replace, for that path only:
 *
 *   pea_plan_*          graph_recsys_benchmark/utils/general_utils.py:280-395 (update_pea_graph_input:
 *                       the P x S list of int64 COO [2,E] tensors is the wire format handed over here)
 *   pea_gat_conv        torch_geometric.nn.GATConv.forward   -- ctor sites models/peagat.py:16-21,
 *   pea_gcn_conv        torch_geometric.nn.GCNConv.forward      models/peagcn.py:16-21,
 *   pea_sage_conv       torch_geometric.nn.SAGEConv.forward     models/peasage.py:16-21; call site
 *                                                               models/base.py:138-139
 *   pea_fuse            models/base.py:193-203   (channel stack, ablation mask, 'att' / 'mean' fusion)
 *   pea_bpr_score       models/base.py:208-214 (predict) + models/base.py:46-48 (BPR loss)
 *   pea_model_*         models/base.py:129-140 + 191-206 (PEABaseChannel.forward loop x P channels,
 *                       then fusion) as ONE scheduled sequence of launches on one stream
 *   pea_rank_eval       solvers.py:85-96 (predict 1+99 candidates, sort, hit vector -> rank of the
 *                       positive; HR/NDCG/AUC follow utils/rec_utils.py:7-30)
 *
 * Conventions
 *   - extern "C"; every function returns int: 0 = ok, < 0 = error (pea_last_error() gives a
 *     thread-local message).  No exceptions, no C++ or torch types cross the boundary.
 *   - All array arguments are DEVICE pointers unless the name ends in _host.  The caller owns every
 *     buffer; the library allocates only the opaque handles' internals.
 *   - `stream` is a hipStream_t passed as void* (0 = null stream).  Every launch goes on it; no
 *     function synchronises the device except pea_plan_create (one-time graph preprocessing).
 *   - fp32 features, int64 ids at the surface (values < num_nodes < 2^31; the plan narrows to int32).
 *   - Handles are immutable after creation and may be used from one stream at a time.
 *   - One device per process (the deployment model is one process per GPU, torch.distributed.run): per-kernel
 *     attributes and the CU count are cached per process after the first launch.
 *
 *   pea_model_forward_train / _backward_level, pea_grad_weight, pea_dense_batch
 *                       solvers.py:213-216 (loss.backward() through the convs: sparse half / dense half)
 *   pea_weighted_aggregate   nn/kgat_conv.py:36-44, nn/kgcn_conv.py:32-37, nn/ngcf_conv.py:42-45 (message + scatter)
 *   pea_sample_negatives     datasets/movielens.py:920-940 (an on-GPU sampler NEXT TO the bit-exact host mirror)
 *   pea_entity_reg           models/base.py:50-73 (entity-aware regulariser of the loss, value + gradient rows)
 *   pea_bpr_train            models/base.py:193-214 + 46-48 under autograd (fusion + scorer + BPR loss, forward + backward)
 *   pea_model_forward_stage[_train], pea_model_backward_level (phases), pea_rows_pack / _unpack / _select_owned,
 *   pea_grad_weight_sharded, pea_dense_batch_rows
 *                       no counterpart in the reference (its step is single-process): one rank's share of a step
 *                       sharded by destination rows over the GPUs of a node; the collectives between the stages are
 *                       RCCL calls made by the host (graph_recsys_benchmark_amd/sharding.py)
 */
#ifndef PEAHIP_H_
#define PEAHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEA_OK 0
#define PEA_ERR_ARG (-1)      /* bad argument (shape, null pointer, unsupported width)          */
#define PEA_ERR_RANGE (-2)    /* an edge_index / id value outside [0, num_nodes)                */
#define PEA_ERR_HIP (-3)      /* a HIP runtime call failed                                      */
#define PEA_ERR_NOMEM (-4)    /* workspace too small / allocation failed                        */
#define PEA_ERR_DEVICE (-5)   /* no gfx950 device / kernel image missing                        */

/* conv kinds */
#define PEA_KIND_GAT 0
#define PEA_KIND_GCN 1
#define PEA_KIND_SAGE 2

/* plan flags */
#define PEA_PLAN_SELF_LOOPS 1 /* drop existing self loops, add one per node (GAT: remove_self_loops +
                                 add_self_loops; GCN: add_remaining_self_loops with unit weights).
                                 Leave clear for SAGE, which aggregates the edge list as given.   */

#define PEA_PLAN_EDGE_IDS 2   /* also keep, per CSR slot, the index of the edge in the caller's COO (needed by
                                 pea_weighted_aggregate, whose weights come in COO order)              */

/* fusion modes (models/base.py:197-203; 'concat' is unusable in the reference, not offered) */
#define PEA_FUSE_ATT 0
#define PEA_FUSE_MEAN 1

typedef struct pea_plan pea_plan;
typedef struct pea_model pea_model;

const char *pea_version(void);
const char *pea_last_error(void);
/* number of visible HIP devices whose arch is gfx950 (0 => compute entry points return PEA_ERR_DEVICE) */
int pea_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Graph plan: destination-sorted CSR (multi-edges kept, edge order preserved inside a row), degree
 * bins and hub-row chunking for every DISTINCT relation of a model.  One-time; synchronises.
 *
 * coo_host[r] : DEVICE pointer to relation r's int64 [2, num_edges[r]] row-major COO
 *               (row 0 = source j, row 1 = target i: PyG flow source_to_target); the array of
 *               pointers itself lives on the host.
 * gather_row_bytes : hint, bytes of one gathered source row (e.g. 4*hidden_size*heads).  When a relation has
 *               >= 2 M edges and its source rows overflow an XCD's L2, the edges of its big hub rows are grouped
 *               by source-id slice and laid out so that one XCD works on one slice at a time (speed only;
 *               the edge order inside such rows is then (slice, COO order)).  0 = plain COO order everywhere.
 * shard_*     : destination-row ownership for multi-GPU runs: row i belongs to rank
 *               (i / shard_tile) % shard_world.  world = 1 -> every row.
 * ---------------------------------------------------------------------------------------------- */
int pea_plan_create(int64_t num_nodes, int n_relations, const int64_t *const *coo_host,
                    const int64_t *num_edges_host, int flags, int gather_row_bytes, int shard_rank,
                    int shard_world, int shard_tile, void *stream, pea_plan **out);
int pea_plan_destroy(pea_plan *plan);
/* info_host[0..8] = {edges kept, max in-degree, #short rows, #long items, #hub rows, #hub chunks,
 *                    rows owned, edges owned, source slices} for one relation */
int pea_plan_relation_info(const pea_plan *plan, int relation, int64_t *info_host);
/* copies the CSR of one relation back (tests): rowptr int32 [N+1], col int32 [edges kept] (device) */
int pea_plan_export_csr(const pea_plan *plan, int relation, int32_t *rowptr, int32_t *col, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-model schedule.  P channels; channel p runs steps[p] conv layers over relations
 * relation_of[p][s] (indices into the plan), all of one `kind`, widths emb_dim -> hidden*heads ->
 * ... -> repr_dim exactly as PEA{GAT,GCN,Sage}Channel builds them (models/peagat.py:14-21):
 * every layer but the last of a multi-step channel has `heads` heads (GAT), the last has 1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pea_model_desc {
    int kind;               /* PEA_KIND_*                                                     */
    int num_channels;       /* P                                                              */
    const int *steps;       /* [P] host                                                       */
    const int *relation_of; /* [sum steps] host, channel-major                                */
    int emb_dim, hidden_size, repr_dim, heads;
    int fuse_mode;          /* PEA_FUSE_*                                                     */
    int gcn_deg_from_col;   /* 0 = degree over the source index (PyG <= 1.5.0), 1 = target    */
    float negative_slope;   /* GAT leaky_relu slope (0.2)                                     */
    int enable_backward;    /* 1: also lay out the training buffers (softmax statistics, gradients)   */
    const int *reverse_of;  /* [plan relations] host: index of each relation's REVERSED relation in the
                               plan (its transposed CSR drives the backward gathers); needed iff
                               enable_backward                                                        */
} pea_model_desc;

int pea_model_create(const pea_plan *plan, const pea_model_desc *desc, pea_model **out);
int pea_model_destroy(pea_model *model);
size_t pea_model_workspace_bytes(const pea_model *model);
/* number of float* parameter slots per conv layer for this kind:
 *   GAT : lin.weight [H*F, in], att_i [H*F], att_j [H*F], bias [H*F]          -> 4
 *   GCN : weight [in, F], bias [F]                                            -> 2
 *   SAGE: lin_rel.weight [F, in], lin_rel.bias [F], lin_root.weight [F, in]   -> 3              */
int pea_model_params_per_layer(const pea_model *model);

/* params_host: host array of DEVICE pointers, channel-major then layer-major then slot (see above).
 * x [N, emb_dim]; att [P, repr_dim] (ignored for PEA_FUSE_MEAN); masked_channel = -1 or the channel
 * zeroed before fusion (models/base.py:194-195).
 * out_repr [N, repr_dim]; out_stack: optional [N, P, repr_dim] (channel order, torch.cat layout), may
 * be NULL.  Only rows owned by this rank's shard are written (all rows when shard_world = 1).      */
int pea_model_forward(pea_model *model, const float *const *params_host, const float *x,
                      const float *att, int masked_channel, void *workspace, size_t workspace_bytes,
                      float *out_repr, float *out_stack, void *stream);
/* ---- training (solvers.py:213-216 zero_grad / loss / backward / step) -----------------------
 * pea_model_forward_train = pea_model_forward that also keeps what the backward needs (per-row softmax statistics)
 * and computes every row through the aggregation kernels.  pea_model_backward_level runs the SPARSE half of one
 * level's backward in the workspace: relu masks, bias and attention-vector gradient reductions, and the gradient
 * gathers over the reversed relations (GAT: two passes, see csrc/agg_bwd.hip).  The dense half (dW = In^T dT,
 * dIn = dT W -- plain GEMMs) is left to the caller's BLAS on views of the same workspace; pea_model_describe returns
 * the buffer layout those views need.  phase: 0 for GAT/GCN; SAGE has phase 0 (masks, bias grads) and phase 1
 * (reverse mean aggregation of the neighbour-mean gradients).  Single GPU. */
int pea_model_forward_train(pea_model *model, const float *const *params_host, const float *x, const float *att,
                            int masked_channel, void *workspace, size_t workspace_bytes, float *out_repr,
                            float *out_stack, void *stream);
int pea_model_backward_level(pea_model *model, int level, int phase, void *workspace, size_t workspace_bytes,
                             void *stream);
int pea_model_describe(const pea_model *model, int64_t *out_host, int max_len, int *needed_host);
/* Optional hint for the next pea_model_backward_level calls (GAT): row_active [N] device bytes, 0 = the gradient of the
 * row's FINAL conv output is exactly zero (a BPR step only touches the rows its batch names).  Such rows are skipped by
 * the last layer's gradient gathers -- the result is the same sum with the zero terms left out.  NULL = every row.
 * The buffer must stay valid until the backward calls have been issued; call again with NULL afterwards.            */
int pea_model_set_active_rows(pea_model *model, const unsigned char *row_active);

/* ---- dense half of the backward (the GEMMs the reference leaves to autograd: torch.nn.Linear / matmul inside the
 * PyG convs, graph_recsys_benchmark/models/base.py:138-139 under loss.backward(), solvers.py:214) -------------------
 * pea_grad_weight:  out[i][j] = sum_n a[n*lda + i] * b[n*ldb + j]   (i < ma, j < nb; out row stride ldo) for a batch of
 *                   jobs over the same n_rows: the weight gradients dW = dT^T In (GAT lin, SAGE lin_rel / lin_root) or
 *                   In^T dT (GCN).  Row parts are reduced in a fixed order: bitwise reproducible, no atomics.
 * pea_dense_batch:  out[n][c] = sum_k a[n*lda + k] * w[k*ldw + c]   (k and n_out multiples of 4): the input gradients
 *                   dIn = dT W of one level, all channels in one launch.
 * pea_model_backward_level: phase | PEA_BWD_PREMASKED = the level's output gradients already carry the relu mask (its
 *                   producer was a gated pea_dense_batch): the level's own relu-mask pass is skipped.                 */
#define PEA_BWD_PREMASKED 0x100
typedef struct pea_gw_job {
    const float *a; int64_t lda; int ma;
    const float *b; int64_t ldb; int nb;
    float *out; int64_t ldo;
    /* optional (NULL: none): rows n with b_mask[n] != 0 take their b operand from b_alt[n*ldb_alt + j] instead.  Two-step
     * training schedule: the input of the first transform is the aggregate A_0[n] where node n has incoming edges under the
     * channel's first relation and x[n] itself where it has none (b = A_0 block, b_mask = the relation's edge-less flags,
     * b_alt = x) -- the x rows are never copied into A_0. */
    const unsigned char *b_mask; const float *b_alt; int64_t ldb_alt;
    const float *b_alt_scale;   /* optional [rows]: the alternative row is multiplied by it (GCN: x[n] * deg^-1, the self-loop norm) */
} pea_gw_job;
typedef struct pea_dense_job {
    const float *a; int64_t lda; int k;
    const float *w; int64_t ldw; int n_out;
    float *out; int64_t ldo;
    /* optional relu gate (NULL: none): out[n][c] = gate[n*ld_gate + c] > 0 ? (a w)[n][c] : 0 -- the mask `F.relu` of the
     * layer below puts on its output gradient (models/base.py:138), applied in the product's epilogue; k <= 128 */
    const float *gate; int64_t ld_gate;
} pea_dense_job;
size_t pea_grad_weight_workspace_bytes(void);
int pea_grad_weight(int64_t n_rows, int n_jobs, const pea_gw_job *jobs_host, void *workspace, size_t workspace_bytes,
                    void *stream);
int pea_dense_batch(int64_t n_rows, int n_jobs, const pea_dense_job *jobs_host, void *stream);
/* Two-step training schedule (GAT; csrc/model.h: fused2_train), dense half of the first layer's backward in ONE launch
 * (csrc/mlp2_bwd.hip) -- loss.backward() (solvers.py:215) through conv_1.lin -> F.relu -> conv_0.lin (models/base.py:138-139):
 *   dz[n, dz_col:+hid] = (dt1[n, dt1_col:+out] . w1)  where  h[n, h_col:+hid] > 0, else 0     (w1 = layer-2 lin.weight [out, hid])
 *   da[n, da_col:+emb] =  dz[n, ...] . w0                                                    (w0 = layer-1 lin.weight [hid, emb])
 * for every channel of the list; emb, hid in {64, 128}, out a multiple of 4 <= 32; all columns / strides multiples of 4;
 * weights_in_out = 1: the weights are laid out [in, out] (GCNConv.weight: w0 [emb, hid], w1 [hid, out]). */
typedef struct pea_mlp2_bwd_chan {
    const float *w0, *w1;
    int dt1_col, h_col, dz_col, da_col;
} pea_mlp2_bwd_chan;
size_t pea_mlp2_backward_data_workspace_bytes(int n_chan, int emb, int hid, int out);
int pea_mlp2_backward_data(int64_t n_rows, int n_chan, const pea_mlp2_bwd_chan *chans_host, int emb, int hid, int out,
                           const float *dt1, int64_t ld_dt1, const float *h, int64_t ld_h, float *dz, int64_t ld_dz,
                           float *da, int64_t ld_da, const int32_t *rows, const int32_t *count_dev, int weights_in_out,
                           void *workspace, size_t workspace_bytes, void *stream);
/* The same for SAGEConv layers (lin_rel / lin_root, both [out, in]; out <= 16): the hidden row feeds two second-layer products
 * and is fed by two first-layer ones (nn SAGEConv: lin_rel(mean_j x_j) + lin_root(x_i), models/base.py:138-139):
 *   dz[n, dz_col:+hid]   = (dt1[n, dt1_col:+out] . w1 + dr1[n, dr1_col:+out] . w1_root)  where  h > 0, else 0
 *   dm[n, da_col:+emb]   =  dz . w0          gradient of the neighbour mean (still to be spread over the reversed relation)
 *   dxr[n, dxr_col:+emb] =  dz . w0_root     gradient of the row's own x through the root term
 * chans_host[c] carries lin_rel of both layers and the shared columns, sage_host[c] the root weights and their columns. */
typedef struct pea_mlp2_bwd_chan_sage {
    const float *w0_root, *w1_root;
    int dr1_col, dxr_col;
} pea_mlp2_bwd_chan_sage;
int pea_mlp2_backward_data_sage(int64_t n_rows, int n_chan, const pea_mlp2_bwd_chan *chans_host,
                                const pea_mlp2_bwd_chan_sage *sage_host, int emb, int hid, int out, const float *dt1,
                                int64_t ld_dt1, const float *dr1, int64_t ld_dr1, const float *h, int64_t ld_h, float *dz,
                                int64_t ld_dz, float *dm, int64_t ld_dm, float *dxr, int64_t ld_dxr, const int32_t *rows,
                                const int32_t *count_dev, void *workspace, size_t workspace_bytes, void *stream);
/* Gradient support of a training step.  The loss reads the batch's rows only (models/base.py:46-48), so the input gradient of
 * the LAST conv layer is identically zero on every node that is neither a batch row nor an in-neighbour of one.
 *   pea_rows_nonzero     flags[n] = 1 where src[n, 0:width] holds a non-zero; list = the ids of those rows in ascending order,
 *                        *count_dev = how many -- all in device memory: the host never reads the count (no synchronisation)
 *   rows / count_dev     of pea_mlp2_backward_data: compute the listed rows only (NULL, NULL: rows 0 .. n_rows - 1)
 *   pea_grad_weight_rows pea_grad_weight over the listed rows (capacity = the list's allocation, <= num_rows)
 *   pea_model_set_active_rows0   level 0 of the two-step training schedule: the bias gradient walks the list; with flags the
 *                        gradient gathers also skip the rows not flagged (pays when few rows are live; without flags the
 *                        caller keeps dA_0 zero outside the list: pea_rows_zero); all NULL: every row                     */
size_t pea_rows_nonzero_workspace_bytes(int64_t n_rows);
int pea_rows_nonzero(int64_t n_rows, int width, const float *src, int64_t ld, unsigned char *flags, int32_t *list,
                     int32_t *count_dev, void *workspace, size_t workspace_bytes, void *stream);
/* ... or flagged in or_flags [n_rows] (may be NULL; must not alias flags): SAGE has no self loops, so the batch's own rows (which
 * receive the root term's gradient) are not among the rows the last layer's gradient gathers reach -- the caller adds them */
int pea_rows_nonzero_or(int64_t n_rows, int width, const float *src, int64_t ld, const unsigned char *or_flags,
                        unsigned char *flags, int32_t *list, int32_t *count_dev, void *workspace, size_t workspace_bytes,
                        void *stream);
int pea_grad_weight_rows(int64_t num_rows, const int32_t *rows, const int32_t *count_dev, int64_t capacity, int n_jobs,
                         const pea_gw_job *jobs_host, void *workspace, size_t workspace_bytes, void *stream);
int pea_model_set_active_rows0(pea_model *model, const unsigned char *flags, const int32_t *list, const int32_t *count_dev);
/* table[list[q], 0:width] = 0 for q < *count_dev: un-does the rows a previous step wrote in a table kept zero elsewhere */
int pea_rows_zero(float *table, int64_t ld, int width, const int32_t *list, const int32_t *count_dev, void *stream);
/* dst[n, 0:width] = sum over b < n_blocks of src[n, b*width : (b+1)*width] (blocks added in order: reproducible): the sum
 * of the per-channel dx parts of the two-step training schedule (the P uses of self.x, models/base.py:193).            */
int pea_block_sum(int64_t n_rows, int n_blocks, int width, const float *src, int64_t ld, float *dst, int64_t ld_dst,
                  void *stream);
/* Sharded training (one rank's share; the host mirror sums the shares with an all-reduce):
 *   pea_grad_weight_sharded  the same reduction over the rows rank `shard_rank` owns (row i -> rank (i / tile) % world)
 *   pea_dense_batch_rows     out = a w on the listed rows only (device int32 [n_rows]; e.g. the rows a rank owns) */
int pea_grad_weight_sharded(int64_t n_rows, int shard_tile, int shard_world, int shard_rank, int n_jobs,
                            const pea_gw_job *jobs_host, void *workspace, size_t workspace_bytes, void *stream);
int pea_dense_batch_rows(int64_t n_rows, const int32_t *rows, int n_jobs, const pea_dense_job *jobs_host, void *stream);

/* ---- device-side BPR negative sampler (an addition next to the bit-exact host mirror of the reference's
 * datasets/movielens.py:920-940 in graph_recsys_benchmark_amd/utils/sampling.py) -----------------------------------
 * For every training positive p and j < k writes row p*k + j of out_triples = (pos_u[p], pos_i[p], i-):
 *   seen_keys_sorted == NULL : i- uniform over [item_lo, item_lo + num_items)               ('random' strategy)
 *   otherwise                : i- uniform over the items with no key  u * num_items + (i - item_lo)  in the ascending
 *                              int64 table (the user's training positives), by rejection       ('unseen' strategy)
 * Stream: Philox4x32-10, key = seed, counter = (row lo, row hi, attempt, offset); exhausted (device int) counts rows
 * whose 64 attempts all hit seen items (their last draw is kept).  All pointers are device pointers.            */
int pea_sample_negatives(int64_t n_pos, int k, const int64_t *pos_u, const int64_t *pos_i, int64_t item_lo,
                         int64_t num_items, const int64_t *seen_keys_sorted, int64_t n_keys, uint64_t seed,
                         uint32_t offset, int64_t *out_triples, int64_t ld_out, int *exhausted, void *stream);

/* ---- multi-GPU (one process per GPU; the library itself never calls RCCL) -------------------------
 * A sharded plan (shard_world > 1) owns destination rows tile-interleaved.  The host mirror computes the layouts
 * with torch ops (graph_recsys_benchmark_amd/sharding.py) and hands them over:
 *   pea_plan_set_owned_rows  rows this rank produces (int32, ascending)
 *   pea_plan_set_sources     per relation used at a level >= 1: slot_of_node [N] (int32, -1 = not a source;
 *                            slot = owner_rank * slots_per_rank + index among that owner's sources) -- the source ids
 *                            of the CSR are renamed to slots of the exchange buffer; and need_rows = the rows whose
 *                            level-0 transform this rank computes itself (own rows + the relation's sources).
 * The forward then runs stage by stage; between stage k and k+1 the host fills, for every exchange descriptor of
 * level k+1, rows [rank*M, rank*M + own) of the exchange buffer from the source buffer and all-gathers it
 * (ncclAllGather via torch.distributed, in place).  Only owned rows of out_repr / out_stack are written.      */
typedef struct pea_exchange_desc {
    int relation;
    int64_t slots_per_rank;    /* M: rows each rank contributes (padded)                              */
    int width;                 /* columns exchanged                                                   */
    int dst_ld;                /* row stride (floats) of the exchange buffer [world*M, dst_ld]        */
    size_t dst_offset_bytes;   /* exchange buffer, from the 256-byte aligned workspace base           */
    size_t src_offset_bytes;   /* node-major source buffer [N, src_ld] in the workspace               */
    int src_ld, src_col;       /* the exchanged columns are [src_col, src_col + width)                */
} pea_exchange_desc;
int pea_plan_set_owned_rows(pea_plan *plan, const int32_t *rows, int64_t n, void *stream);
int pea_plan_set_sources(pea_plan *plan, int relation, const int32_t *slot_of_node, int64_t slots_per_rank,
                         const int32_t *need_rows, int64_t n_need, void *stream);
int pea_model_num_stages(const pea_model *model);
int pea_model_forward_stage(pea_model *model, int stage, const float *const *params_host, const float *x,
                            const float *att, int masked_channel, void *workspace, size_t workspace_bytes,
                            float *out_repr, float *out_stack, void *stream);
/* the same stage of the TRAINING forward (model created with enable_backward): keeps what pea_model_backward_level
 * reads.  Backward of a sharded model, per level: phase 0, host fills in the rows of dX / dO_s / side_s that other ranks
 * own (sources of the reversed relation), phase 2 (GAT / GCN); SAGE: phase 0, host dense half + fill-in of dM_s, phase 1. */
int pea_model_forward_stage_train(pea_model *model, int stage, const float *const *params_host, const float *x,
                                  const float *att, int masked_channel, void *workspace, size_t workspace_bytes,
                                  float *out_repr, float *out_stack, void *stream);
/* A stage in parts, so that the exchange after it can run behind the work that does not feed it (round 3):
 *   PEA_PART_ALL      the whole stage (what pea_model_forward_stage runs)
 *   PEA_PART_SOURCES  everything the next exchange needs: the stage restricted to the first n_first owned rows
 *                     (pea_plan_set_owned_split: the owned rows that are gather sources of a later level come first in
 *                     the owned-row list) -- where pea_model_stage_fills_exchange() says so, those rows are written into
 *                     this rank's block of the exchange buffers directly and the host starts the all-gather right away
 *   PEA_PART_REST     the remaining owned rows (only this rank reads them)
 * Stages that cannot be split run entirely under PEA_PART_SOURCES and do nothing under PEA_PART_REST.
 * Last stage: n_sel > 0 adds the batch's rows to the fusion launch: sel_out[k, 0:repr_dim] = the fused row of node
 * sel_ids[k * sel_stride] when this rank owns it, zeros otherwise (the operand of the loss all-reduce; reference
 * models/base.py:46-47 reads cached_repr[unids] / [inids]); *err_flag |= 1 for an id outside [0, num_nodes).        */
enum { PEA_PART_ALL = 0, PEA_PART_SOURCES = 1, PEA_PART_REST = 2 };
typedef struct pea_stage_opts {
    int part;
    const int64_t *sel_ids;
    int64_t sel_stride, n_sel;
    float *sel_out;
    int32_t *err_flag;
} pea_stage_opts;
int pea_plan_set_owned_split(pea_plan *plan, int64_t n_first);
int pea_model_forward_part(pea_model *model, int stage, const pea_stage_opts *opts, const float *const *params_host,
                           const float *x, const float *att, int masked_channel, void *workspace, size_t workspace_bytes,
                           float *out_repr, float *out_stack, void *stream);
int pea_model_stage_fills_exchange(const pea_model *model, int stage);
int pea_model_num_exchanges(const pea_model *model, int level);
int pea_model_exchange_desc(const pea_model *model, int level, int k, pea_exchange_desc *out);

/* ------------------------------------------------------------------------------------------------
 * Launch tape.  No counterpart in the reference (its step is a Python loop over torch ops).  A step of this library is a
 * fixed sequence of kernel launches whose arguments only change when a buffer moves; for launch-bound sizes (a rank of 8:
 * ten launches of 10-90 us) building the launch descriptors again every step costs as much host time as the GPU needs to
 * run them.  pea_tape_begin .. pea_tape_end records every launch the library issues from the calling thread (they also
 * execute); pea_tape_replay re-issues them on `stream` -- same kernels, same order, arguments captured by value, entering
 * the stream as ordinary launches (not a hipGraph: a graph replay of the same sequence ran slower).  Validity is the
 * caller's: replay only while every captured pointer still refers to the same buffer.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pea_tape pea_tape;
int pea_tape_create(pea_tape **out);
int pea_tape_destroy(pea_tape *tape);
int pea_tape_begin(pea_tape *tape);
int pea_tape_end(pea_tape *tape);
int pea_tape_length(const pea_tape *tape);
int pea_tape_replay(const pea_tape *tape, void *stream);

/* messages reduced by one forward (sum over channels/steps of kept edges + self loops), and the
 * algorithmic HBM bytes of SURVEY.md section 8(d) for this model -- the roofline yardstick. */
int pea_model_stats(const pea_model *model, int64_t *messages, double *algorithmic_bytes);
/* Compulsory HBM bytes of one forward of THIS schedule: x read once, every level buffer (T_s, O_s of the channels that
 * continue, X) written once and read once, every aggregation group's index arrays read once, the fused table written
 * once.  step time x 8 TB/s over this number = distance of the whole step from the DRAM floor (bench.py `hbm_floor`). */
double pea_model_compulsory_bytes(const pea_model *model);

/* ------------------------------------------------------------------------------------------------
 * Single conv layers (the drop-in GATConv/GCNConv/SAGEConv modules call these).
 * x [N, in] with row stride ldx (floats); out [N, heads*out_channels] with row stride ldo.
 * workspace: pea_conv_workspace_bytes(...) bytes (0 = bad arguments, see pea_last_error()).
 * ---------------------------------------------------------------------------------------------- */
size_t pea_conv_workspace_bytes(const pea_plan *plan, int kind, int relation, int in_channels, int heads,
                                int out_channels);
int pea_gat_conv(const pea_plan *plan, int relation, int in_channels, int heads, int out_channels,
                 const float *x, int64_t ldx, const float *lin_weight, const float *att_i,
                 const float *att_j, const float *bias, float negative_slope, int relu, float *out,
                 int64_t ldo, void *workspace, size_t workspace_bytes, void *stream);
int pea_gcn_conv(const pea_plan *plan, int relation, int in_channels, int out_channels, const float *x,
                 int64_t ldx, const float *weight, const float *bias, int deg_from_col, int relu,
                 float *out, int64_t ldo, void *workspace, size_t workspace_bytes, void *stream);
int pea_sage_conv(const pea_plan *plan, int relation, int in_channels, int out_channels, const float *x,
                  int64_t ldx, const float *rel_weight, const float *rel_bias, const float *root_weight,
                  int relu, float *out, int64_t ldo, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Weighted neighbour sum with per-edge weights in the caller's COO order -- the message/aggregate half of the
 * reference's own baseline convs (nn/kgat_conv.py:36-44, nn/kgcn_conv.py:32-37: x_j * att_map; nn/ngcf_conv.py:42-45
 * with its degree coefficient folded into the weight):   out_i = sum_{e: j -> i} w_e x_j.
 * x [N, width] stride ldx, out [N, width] stride ldo, edge_weight [num_edges of the relation].  The plan must carry
 * PEA_PLAN_EDGE_IDS and, like those convs after remove_self_loops, no self loops are added.
 * ---------------------------------------------------------------------------------------------- */
size_t pea_weighted_aggregate_workspace_bytes(const pea_plan *plan, int relation, int width);
int pea_weighted_aggregate(const pea_plan *plan, int relation, int width, const float *x, int64_t ldx,
                           const float *edge_weight, float *out, int64_t ldo, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fusion of the channel stack (models/base.py:196-203).
 * stack: channel p's [N, R] block starts at column col_of_channel_host[p] of a row-major
 * [N, ld_stack] matrix (torch.cat layout: col = p*R, ld = P*R).  att [P, R] or NULL for MEAN.
 * ---------------------------------------------------------------------------------------------- */
int pea_fuse(int64_t num_nodes, int P, int R, const float *stack, int64_t ld_stack,
             const int *col_of_channel_host, const float *att, int masked_channel, int fuse_mode,
             float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * BPR scoring (models/base.py:208-214, 46-48): for each of B rows (u, i+, i-) of `triples`
 * (int64, row stride triple_stride >= 3 -- 9 for entity-aware batches):
 *   pos = fc2(relu(fc1([repr[u] || repr[i+]])));  neg likewise;  loss = -sum(log(sigmoid(pos - neg)))
 * fc1_w [R, 2R], fc1_b [R], fc2_w [1, R], fc2_b [1].  pos/neg [B] (may be NULL), loss [1].
 * workspace: pea_bpr_workspace_bytes(B) bytes.  Deterministic (no float atomics).  Stays asynchronous: a row
 * with an id outside [0, num_nodes) contributes nothing and sets the int at workspace[0] to 1 (the host
 * mirror checks it when asked to validate).
 * ---------------------------------------------------------------------------------------------- */
size_t pea_bpr_workspace_bytes(int64_t B);
int pea_bpr_score(int64_t B, int R, int64_t num_nodes, const float *repr, const int64_t *triples,
                  int64_t triple_stride, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                  const float *fc2_b, float *pos, float *neg, float *loss, void *workspace,
                  size_t workspace_bytes, void *stream);
/* predict only (models/base.py:208-214): pred[b] for pairs (unids[b], inids[b]).  pea_predict and
 * pea_rank_eval synchronise the stream once to report ids outside [0, num_nodes) as PEA_ERR_RANGE. */
int pea_predict(int64_t B, int R, int64_t num_nodes, const float *repr, const int64_t *unids,
                const int64_t *inids, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                const float *fc2_b, float *pred, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Batched evaluator (solvers.py:56-96 per-user loop as one launch): U users, C candidates each
 * (candidate 0 is the held-out positive, 1..C-1 the sampled negatives, ids drawn on the host with
 * np.random.choice for bit-exactness).  Writes per user: scores [U, C], rank of the positive
 * (number of negatives scoring strictly higher + ties placed as torch.sort(descending, stable)
 * would: a tie with an earlier index wins), auc [U], eval loss [U] (= -sum log sigmoid(pos - neg)).
 * ---------------------------------------------------------------------------------------------- */
int pea_rank_eval(int64_t U, int C, int R, int64_t num_nodes, const float *repr, const int64_t *unids,
                  const int64_t *cand /*[U, C]*/, const float *fc1_w, const float *fc1_b,
                  const float *fc2_w, const float *fc2_b, float *scores, int32_t *rank, float *auc,
                  float *loss, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Entity-aware regulariser of the loss (models/base.py:50-73, only with --entity_aware=true):
 *   reg = -sum_b log sigmoid((|x[i]-x[e+]|^2 - |x[i]-x[e-]|^2) * m_i) - sum_b log sigmoid((|x[u]-x[f+]|^2 - |x[u]-x[f-]|^2) * m_u)
 * over the B rows (u, i+, i-, e+, e-, m_i, f+, f-, m_u) of `batch` (int64, row stride >= 9).  One launch gathers the six
 * x rows of every batch row; fixed-order reduction.  grad_rows (optional, [6B, emb_dim]): d reg / d (the six gathered
 * rows), order i, e+, e-, u, f+, f- -- the caller scatters them into dx.  A node id out of range makes *out_reg NaN
 * and sets the int32 flag at the start of the workspace.
 * ---------------------------------------------------------------------------------------------- */
size_t pea_entity_reg_workspace_bytes(int64_t B, int emb_dim);
int pea_entity_reg(int64_t B, int emb_dim, int64_t num_nodes, const float *x, int64_t ldx, const int64_t *batch,
                   int64_t batch_stride, float *out_reg, float *grad_rows, void *workspace, size_t workspace_bytes,
                   void *stream);

/* ------------------------------------------------------------------------------------------------
 * Training-step head (solvers.py:213-214, loss = model.loss(batch); loss.backward()): channel fusion
 * (models/base.py:193-203), the fc1 / fc2 scorer (:208-214) and the BPR loss (:46-48) of the B triples of a batch,
 * forward and backward in one launch.  rows [3B, ld_rows]: the P*R stack row of (user, pos item, neg item) of triple b at
 * rows 3b, 3b+1, 3b+2.  att: [P*R] ('att' fusion) or NULL ('mean').  Outputs: *out_loss; grad_rows [3B, P*R] = d loss /
 * d rows; and the operands of the parameter gradients, which the caller reduces with pea_grad_weight (fixed order):
 *   dhx [2B, R+4], zx [2B, 3R+4]:  G = dhx^T zx  ->  d fc1.weight = G[0:R, 0:2R], d fc1.bias = G[0:R, 3R],
 *                                                     d fc2.weight = G[R, 2R:3R],  d fc2.bias = G[R, 3R]
 *   dsc [3B, P4] (P4 = P rounded up to 4; 'att' only):  d att[p, :] = (dsc^T rows)[p, p*R:(p+1)*R]
 * repr_dim R: a multiple of 4, <= 32 (pea_bpr_train_supported).  No atomics, fixed reduction order.
 * pea_rows_scatter_sum: dst[id, col_of_channel[p] + c] = sum over the positions k with ids[k] == id of src[k, p*R + c],
 * added in increasing k (the index backward of rows = stack[ids]: the batch's gradient rows into the node-indexed
 * output-gradient buffer: the batch's (id, position) keys are sorted in LDS by one workgroup, then one wave per node adds
 * its rows in position order; ids < 0 or >= num_rows (the rows of dst) are skipped; n <= 16384, P*R <= 1024).
 * ---------------------------------------------------------------------------------------------- */
size_t pea_bpr_train_workspace_bytes(int64_t B);
int pea_bpr_train_supported(int P, int R);
int pea_bpr_train(int64_t B, int P, int R, const float *rows, int64_t ld_rows, const float *att, const float *fc1_w,
                  const float *fc1_b, const float *fc2_w, const float *fc2_b, float *out_loss, float *grad_rows,
                  float *dhx, float *zx, float *dsc, void *workspace, size_t workspace_bytes, void *stream);
size_t pea_rows_scatter_sum_workspace_bytes(int64_t n);
int pea_rows_scatter_sum(int64_t n, const int64_t *ids, const float *src, int64_t ld_src, int P, int R,
                         const int *col_of_channel_host, float *dst, int64_t ld_dst, int64_t num_rows, void *workspace,
                         size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU exchange helpers (one process per GPU; the collectives themselves are RCCL calls made by the host
 * mirror, graph_recsys_benchmark_amd/sharding.py).  No counterpart in the reference (single-process forward,
 * models/base.py:191-206).  All widths / strides / columns in floats, multiples of 4.
 *   pea_rows_pack    dst[k, 0:width]                = table[nodes[k], col:col+width]
 *   pea_rows_unpack  table[nodes[k], col:col+width] = src[src_rows ? src_rows[k] : k, 0:width]
 *   pea_rows_select_owned  out[k, 0:width] = ((ids[k*id_stride] / tile) % world == rank) ? table[ids[..], 0:width] : 0
 *                    (the batch rows a rank contributes to the loss all-reduce; *err_flag |= 1 for an id outside
 *                    [0, num_nodes), device int32 owned by the caller)
 * ---------------------------------------------------------------------------------------------- */
int pea_rows_pack(const float *table, int64_t ld, int col, int width, const int32_t *nodes, int64_t n, float *dst,
                  int64_t dst_ld, void *stream);
int pea_rows_unpack(const float *src, int64_t src_ld, const int32_t *src_rows, int width, const int32_t *nodes, int64_t n,
                    float *table, int64_t ld, int col, void *stream);
int pea_rows_select_owned(const float *table, int64_t ld, int width, int64_t num_nodes, const int64_t *ids,
                          int64_t id_stride, int64_t n, int rank, int world, int tile, float *out, int32_t *err_flag,
                          void *stream);
/* Batched pack / unpack against ONE rank-major exchange buffer (rank blocks `rank_stride` floats apart): job q moves the
 * columns [col, col + width) of table rows nodes[0..n) to / from the rows at buf_off (floats from the start of a rank
 * block, `width` floats per row).  pack: row k of the job goes to this rank's block (`rank_block`).  unpack: row k comes
 * from slot slots[k] = rank * slots_per_rank + local of `buffer` (= rank 0's block).  One launch per <= 24 jobs: a sharded
 * backward level fills in all its gradient buffers with one pack, ONE collective and one unpack. */
typedef struct pea_xchg_job {
    float *table; int64_t ld; int col; int width;
    const int32_t *nodes; const int32_t *slots; int64_t n;
    int64_t buf_off; int slots_per_rank;
} pea_xchg_job;
int pea_rows_pack_batch(int n_jobs, const pea_xchg_job *jobs_host, float *rank_block, void *stream);
int pea_rows_unpack_batch(int n_jobs, const pea_xchg_job *jobs_host, const float *buffer, int64_t rank_stride, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Live per-launch timing (bench.py's roofline leg): when enabled every kernel launch of the library is
 * bracketed by HIP events on the caller's stream.  pea_profile_read waits for them and returns, in launch
 * order, name (32 bytes each), milliseconds and the algorithmic bytes attributed to the launch; it clears
 * the log.  Not for use under hipGraph capture.
 * ---------------------------------------------------------------------------------------------- */
int pea_profile_enable(int on);
int pea_profile_read(int max_records, char *names_host, float *ms_host, double *bytes_host, int *count_host);
/* The same, with two more per-launch figures of the aggregation kernels: `gathered` = bytes the launch itself must
 * pull through the memory system (row chunk + source index of every message, each once), `table` = footprint in bytes
 * of the largest table those rows come from (decides the cache tier the gather is served from). */
int pea_profile_read_ex(int max_records, char *names_host, float *ms_host, double *bytes_host, double *gathered_host,
                        double *table_host, int *count_host);

#ifdef __cplusplus
}
#endif
#endif /* PEAHIP_H_ */
