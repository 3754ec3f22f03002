"""Host side of the HIP path: graph plan + whole-model schedule behind the reference's module surface.

    GraphPlan  <-  the P x S list of int64 COO tensors built by update_pea_graph_input
                   (reference graph_recsys_benchmark/utils/general_utils.py:280-395)
    PEAEngine  <-  PEABaseRecsysModel.forward (reference models/base.py:191-206) for one conv kind

PyTorch is used for device memory and streams only; all arithmetic runs in libpeahip.so.
"""
import ctypes as C

import torch

from . import _lib
from .sharding import ShardLayout

KINDS = {'gat': _lib.KIND_GAT, 'gcn': _lib.KIND_GCN, 'sage': _lib.KIND_SAGE}
GRAPHS_ENABLED = True     # bench.py clears it for its per-kernel HIP-event pass (events are not part of a captured graph)


def _shard_replay_mode():
    import os
    mode = os.environ.get('PEA_SHARD_REPLAY', 'tape')
    return mode if mode in ('tape', 'graph', 'eager') else 'tape'

# parameter slots per conv layer, in the order pea_model_forward expects (include/peahip.h)
PARAM_SLOTS = {
    'gat': ('lin.weight', 'att_i', 'att_j', 'bias'),
    'gcn': ('weight', 'bias'),
    'sage': ('lin_rel.weight', 'lin_rel.bias', 'lin_root.weight'),
}


def _fingerprint(ei):
    e = ei.shape[1]
    if e == 0:
        return (0, 0)
    idx = torch.arange(e, device=ei.device, dtype=torch.int64)
    h = (ei[0] * 1000003 + ei[1] * 998244353 + idx * 7919).sum()
    return (e, int(h.item()))


class GraphPlan:
    """Destination-sorted CSR + degree bins for every DISTINCT relation of a metapath list.

    Equal-content tensors are planned once: update_pea_graph_input creates a fresh
    torch.flip(user2item) copy per use (utils/general_utils.py:300-307), so dedupe is by content.
    """

    def __init__(self, num_nodes, meta_path_edge_index_list, self_loops, shard_rank=0, shard_world=1,
                 shard_tile=256, gather_row_bytes=256, with_reverse=False):
        lib = _lib.require_device()
        self.num_nodes = int(num_nodes)
        self.self_loops = bool(self_loops)
        self.shard = (int(shard_rank), int(shard_world), int(shard_tile))
        uniq, keys = [], {}
        self.relation_of = []
        for eil in meta_path_edge_index_list:
            row = []
            for ei in eil:
                if ei.dim() != 2 or ei.shape[0] != 2 or ei.dtype != torch.int64:
                    raise ValueError('edge_index must be an int64 [2, E] tensor, got %s %s' % (ei.dtype, tuple(ei.shape)))
                if not ei.is_cuda:
                    raise RuntimeError('edge_index must live on the GPU (the HIP path has no CPU fallback)')
                ei = ei.contiguous()
                fp = _fingerprint(ei)
                found = None
                for cand in keys.get(fp, []):
                    if torch.equal(uniq[cand], ei):
                        found = cand
                        break
                if found is None:
                    found = len(uniq)
                    uniq.append(ei)
                    keys.setdefault(fp, []).append(found)
                row.append(found)
            self.relation_of.append(row)
        # training: the backward gathers run over the REVERSED relations; add the ones the model does not use itself
        self.reverse_of = None
        if with_reverse:
            self.reverse_of = []
            for r in range(len(uniq)):
                flipped = torch.flip(uniq[r], dims=[0]).contiguous()
                fp = _fingerprint(flipped)
                found = None
                for cand in keys.get(fp, []):
                    if torch.equal(uniq[cand], flipped):
                        found = cand
                        break
                if found is None:
                    found = len(uniq)
                    uniq.append(flipped)
                    keys.setdefault(fp, []).append(found)
                self.reverse_of.append(found)
            # reversed relations appended above get their own entry too (reverse of a reverse is the original)
            for r in range(len(self.reverse_of), len(uniq)):
                self.reverse_of.append(next(i for i, rr in enumerate(self.reverse_of) if rr == r))
        self.num_relations = len(uniq)
        self._uniq = uniq
        ptrs = (C.c_void_p * len(uniq))(*[t.data_ptr() for t in uniq])
        nedge = (C.c_int64 * len(uniq))(*[t.shape[1] for t in uniq])
        handle = C.c_void_p()
        _lib.check(lib.pea_plan_create(self.num_nodes, len(uniq), ptrs, nedge,
                                       _lib.PLAN_SELF_LOOPS if self_loops else 0, int(gather_row_bytes), self.shard[0],
                                       self.shard[1], self.shard[2], _lib.current_stream(), C.byref(handle)))
        self._h = handle
        self.device = uniq[0].device
        self.layout = ShardLayout(self.num_nodes, *self.shard)
        self.source_layouts = {}
        if self.shard[1] > 1:
            lays = [self.layout.source_layout(ei) for ei in uniq]
            # owned rows, the ones OTHER ranks read first: the sources of the relations used at a step >= 1 (the exchanges
            # between the stages move exactly these rows).  A stage computes them first, the all-gather starts, and the rest
            # of the stage (rows only this rank reads) runs behind it (pea_model_forward_part).
            own64 = self.layout.owned_rows(self.device)
            later = sorted({r for row in self.relation_of for r in row[1:]})
            is_src = torch.zeros(self.num_nodes, dtype=torch.bool, device=self.device)
            for r in later:
                is_src[lays[r].src_nodes] = True
            first = is_src[own64]
            own = torch.cat([own64[first], own64[~first]]).to(torch.int32).contiguous()
            self.own_rows_i32 = own
            self.n_own_first = int(first.sum().item())
            _lib.check(lib.pea_plan_set_owned_rows(handle, _lib.ptr(own), own.numel(), _lib.current_stream()))
            _lib.check(lib.pea_plan_set_owned_split(handle, self.n_own_first))
            # relations with few source nodes (attribute -> item ...) share ONE first-layer row list: own rows + the
            # union of their sources, so their channels' transforms run as one wide job (a few extra rows, same results)
            small = [r for r, lay in enumerate(lays) if lay.src_nodes.numel() * 10 <= own.numel()]
            if len(small) > 1:
                shared = torch.unique(torch.cat([self.layout.owned_rows(self.device)] +
                                                [lays[r].src_nodes for r in small])).to(torch.int32)
                for r in small:
                    lays[r].need_rows = shared
            for r, lay in enumerate(lays):
                need = lay.need_rows.contiguous()
                _lib.check(lib.pea_plan_set_sources(handle, r, _lib.ptr(lay.slot_of_node.contiguous()), lay.slots_per_rank,
                                                    _lib.ptr(need), need.numel(), _lib.current_stream()))
                self.source_layouts[r] = lay

    def relation_info(self, r):
        info = (C.c_int64 * 9)()
        _lib.check(_lib.load().pea_plan_relation_info(self._h, r, info))
        names = ('edges', 'max_degree', 'short_rows', 'long_items', 'hub_rows', 'hub_chunks', 'rows_owned', 'edges_owned',
                 'slices')
        return dict(zip(names, [int(v) for v in info]))

    def gcn_self_norm(self, r, from_col=False):
        """float32 [N]: deg^-1 of GCNConv's normalisation under relation r (self loops dropped, one added per node; degree over
        the source index -- PyG 1.5.0 -- or the target index): the weight dinv_i^2 of a node's own row, which is ALL of the
        aggregate of a node without incoming edges.  Built once per relation and degree side."""
        cache = self.__dict__.setdefault('_self_norm', {})
        key = (r, bool(from_col))
        if key not in cache:
            ei = self._uniq[r]
            keep = ei[0] != ei[1]
            idx = ei[1 if from_col else 0][keep]
            deg = torch.bincount(idx, minlength=self.num_nodes).to(torch.float32) + 1.0
            cache[key] = (1.0 / deg).contiguous()
        return cache[key]

    def edgeless_mask(self, r):
        """uint8 [N]: 1 where node n has no kept incoming edge under relation r (GAT / GCN: its conv output is its own
        transformed row; the two-step schedules feed x[n] instead of an aggregate).  Built once per relation."""
        cache = self.__dict__.setdefault('_edgeless', {})
        if r not in cache:
            rowptr, _ = self.export_csr(r)
            cache[r] = (rowptr[1:] == rowptr[:-1]).to(torch.uint8).contiguous()
        return cache[r]

    def export_csr(self, r):
        info = self.relation_info(r)
        rowptr = torch.empty(self.num_nodes + 1, dtype=torch.int32, device=self.device)
        col = torch.empty(max(info['edges'], 1), dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().pea_plan_export_csr(self._h, r, _lib.ptr(rowptr), _lib.ptr(col), _lib.current_stream()))
        return rowptr, col[:info['edges']]

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.load().pea_plan_destroy(h)
            except Exception:
                pass


class PEAEngine:
    """One scheduled PEA forward: P channels x S conv layers of one kind, then fusion."""

    def __init__(self, plan, kind, steps, emb_dim, hidden_size, repr_dim, heads=1, channel_aggr='att',
                 gcn_deg_from='row', negative_slope=0.2, enable_backward=False):
        if channel_aggr not in ('att', 'mean'):
            # 'concat' cannot work in the reference either (models/base.py:175 sizes fc1 for 'cat')
            raise NotImplementedError('Other aggr methods not implemeted!')
        if kind not in KINDS:
            raise ValueError(kind)
        if gcn_deg_from not in ('row', 'col'):
            raise ValueError(gcn_deg_from)
        lib = _lib.require_device()
        self.plan, self.kind = plan, kind
        self.steps = [int(s) for s in steps]
        if len(self.steps) != len(plan.relation_of) or any(len(r) != s for r, s in zip(plan.relation_of, self.steps)):
            raise AssertionError('meta_path_steps does not match the edge index lists')
        self.P = len(self.steps)
        self.repr_dim, self.emb_dim = int(repr_dim), int(emb_dim)
        self.channel_aggr = channel_aggr
        flat = [r for row in plan.relation_of for r in row]
        self._steps_c = (C.c_int * self.P)(*self.steps)
        self._rel_c = (C.c_int * len(flat))(*flat)
        self.enable_backward = bool(enable_backward)
        self._gcn_from_col = gcn_deg_from == 'col'
        self._rev_c = None
        if self.enable_backward:
            if plan.reverse_of is None:
                raise ValueError('training needs a GraphPlan built with with_reverse=True')
            self._rev_c = (C.c_int * len(plan.reverse_of))(*plan.reverse_of)
        desc = _lib.ModelDesc(KINDS[kind], self.P, self._steps_c, self._rel_c, int(emb_dim), int(hidden_size),
                              int(repr_dim), int(heads), _lib.FUSE_ATT if channel_aggr == 'att' else _lib.FUSE_MEAN,
                              1 if gcn_deg_from == 'col' else 0, float(negative_slope), 1 if self.enable_backward else 0,
                              self._rev_c)
        handle = C.c_void_p()
        _lib.check(lib.pea_model_create(plan._h, C.byref(desc), C.byref(handle)))
        self._h = handle
        self.workspace_bytes = int(lib.pea_model_workspace_bytes(handle))
        self._ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=plan.device)
        self.n_layers = sum(self.steps)
        self.slots = len(PARAM_SLOTS[kind])
        msgs, ab = C.c_int64(), C.c_double()
        _lib.check(lib.pea_model_stats(handle, C.byref(msgs), C.byref(ab)))
        self.messages, self.algorithmic_bytes = int(msgs.value), float(ab.value)
        self.compulsory_bytes = float(lib.pea_model_compulsory_bytes(handle))   # HBM floor of one forward (bench.py)
        self.sharded = plan.shard[1] > 1
        self.n_stages = int(lib.pea_model_num_stages(handle))
        # float32 view of the workspace from its 256-byte aligned base (the exchanges index into it)
        skew = (-self._ws.data_ptr()) % 256
        self._wsf = self._ws[skew:skew + (self.workspace_bytes - 256) // 4 * 4].view(torch.float32)
        self._exchanges = []            # per level >= 1: [(desc, source table view, exchange buffer view, layout)]
        if self.sharded:
            n, world = plan.num_nodes, plan.shard[1]
            for level in range(self.n_stages):
                row = []
                for k in range(int(lib.pea_model_num_exchanges(handle, level))):
                    d = _lib.ExchangeDesc()
                    _lib.check(lib.pea_model_exchange_desc(handle, level, k, C.byref(d)))
                    src = self._wsf[d.src_offset_bytes // 4:d.src_offset_bytes // 4 + n * d.src_ld].view(n, d.src_ld)
                    rows = world * d.slots_per_rank
                    dst = self._wsf[d.dst_offset_bytes // 4:d.dst_offset_bytes // 4 + rows * d.dst_ld].view(rows, d.dst_ld)
                    row.append((d, src, dst, plan.source_layouts[d.relation]))
                self._exchanges.append(row)
            # stages whose transform writes this rank's exchange rows itself (no pack launch before the all-gather)
            self._fills = [bool(lib.pea_model_stage_fills_exchange(handle, k)) for k in range(self.n_stages)]

    def forward(self, layer_params, x, att=None, masked=None, want_stack=False, train=False, gather=True, out=None,
                select_ids=None):
        """layer_params: list (channel-major, then step) of tuples of tensors in PARAM_SLOTS order
        (a missing bias may be None).  train=True keeps what backward() needs (single GPU).  gather=False (sharded
        plans only): skip the final all-gather; only the rows this rank owns are defined in the result.  select_ids
        (sharded, int64 [K]): also returns [K, repr_dim] rows = the fused rows of those nodes this rank owns, zeros for
        the others (one launch with the fusion; the caller all-reduces them: ShardLayout.reduce_rows)."""
        lib = _lib.load()
        n = self.plan.num_nodes
        if x.shape != (n, self.emb_dim) or x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError('x must be a CUDA float32 [%d, %d] tensor' % (n, self.emb_dim))
        if len(layer_params) != self.n_layers:
            raise ValueError('expected %d conv layers, got %d' % (self.n_layers, len(layer_params)))
        keep = [x.contiguous()]
        # The pointer table is rebuilt only when a parameter's storage moved (optimizers update in place): the checks and
        # the ctypes stores below were 80 us of a 300 us step on the launch-bound presets.  A cached table is reused only if
        # every tensor was contiguous float32 when it was built, so the raw pointers are the tensors' own.
        sig = tuple(None if t is None else t.data_ptr() for lp in layer_params for t in lp)
        cached = getattr(self, '_ptr_cache', None)
        if cached is not None and cached[0] == sig:
            ptrs = cached[1]
        else:
            ptrs = (C.c_void_p * (self.n_layers * self.slots))()
            k, plain = 0, True
            for lp in layer_params:
                if len(lp) != self.slots:
                    raise ValueError('each %s layer needs %d parameter tensors' % (self.kind, self.slots))
                for t in lp:
                    if t is None:
                        ptrs[k] = None
                    else:
                        t = t.detach()
                        if t.dtype != torch.float32 or not t.is_cuda:
                            raise ValueError('parameters must be CUDA float32 tensors')
                        plain = plain and t.is_contiguous()
                        t = t.contiguous()
                        keep.append(t)
                        ptrs[k] = t.data_ptr()
                    k += 1
            self._ptr_cache = (sig, ptrs) if plain and len(sig) == self.n_layers * self.slots else None
        att_t = None
        if self.channel_aggr == 'att':
            if att is None:
                raise ValueError("att is required for channel_aggr='att'")
            att_t = att.detach().reshape(self.P, self.repr_dim).contiguous()
            keep.append(att_t)
        if out is None:
            out = torch.empty((n, self.repr_dim), dtype=torch.float32, device=x.device)
        stack = torch.empty((n, self.P, self.repr_dim), dtype=torch.float32, device=x.device) if want_stack else None
        m = -1 if masked is None else int(masked)
        if not self.sharded:
            fn = lib.pea_model_forward_train if train else lib.pea_model_forward
            _lib.check(fn(self._h, ptrs, _lib.ptr(keep[0]), _lib.ptr(att_t), m, _lib.ptr(self._ws),
                          self.workspace_bytes, _lib.ptr(out), _lib.ptr(stack), _lib.current_stream()))
            return (out, stack) if want_stack else out
        # Sharded forward (train=True: the training schedule, stage by stage, same exchanges): stage k computes this rank's rows of level k (and the transform feeding level k+1); the
        # gather sources of level k+1 are then all-gathered from their owners; after the last stage the fused rows
        # (not the per-metapath stack) are all-gathered: the fusion is row-local under row ownership.
        shard = self.plan.layout
        picked = None
        if train:
            if select_ids is not None:
                raise ValueError('select_ids rides in the inference stages (the training step exchanges stack rows)')
            for k in range(self.n_stages):
                _lib.check(lib.pea_model_forward_stage_train(self._h, k, ptrs, _lib.ptr(keep[0]), _lib.ptr(att_t), m,
                                                             _lib.ptr(self._ws), self.workspace_bytes, _lib.ptr(out),
                                                             _lib.ptr(stack), _lib.current_stream()))
                if k + 1 < self.n_stages:
                    for d, src, dst, lay in self._exchanges[k + 1]:
                        shard.exchange_sources(dst, src, lay, d.src_col, d.width)
        else:
            # stage k in two parts: (1) everything the exchange after it needs -- the all-gathers of the next level's gather
            # sources then start (asynchronously under RCCL) -- (2) the rows only this rank reads, behind them; the stream
            # waits for the collectives just before the next stage.  The last stage's fusion launch also picks the batch's
            # rows for the loss all-reduce (select_ids).
            opts = _lib.StageOpts(_lib.PART_ALL, None, 1, 0, None, None)
            for k in range(self.n_stages):
                last = k + 1 == self.n_stages
                opts.part = _lib.PART_ALL if last else _lib.PART_SOURCES
                if last and select_ids is not None:
                    ids = select_ids if select_ids.dtype == torch.int64 else select_ids.to(torch.int64)
                    picked = torch.empty((ids.numel(), self.repr_dim), dtype=torch.float32, device=x.device)
                    keep.append(ids)
                    flag = shard._err_flag(x.device)
                    opts.sel_ids, opts.sel_stride, opts.n_sel = ids.data_ptr(), ids.stride(0) if ids.dim() == 1 else 1, ids.numel()
                    opts.sel_out, opts.err_flag = picked.data_ptr(), flag.data_ptr()
                _lib.check(lib.pea_model_forward_part(self._h, k, C.byref(opts), ptrs, _lib.ptr(keep[0]), _lib.ptr(att_t), m,
                                                      _lib.ptr(self._ws), self.workspace_bytes, _lib.ptr(out),
                                                      _lib.ptr(stack), _lib.current_stream()))
                if last:
                    break
                filled = self._fills[k]
                pending = [shard.exchange_sources(dst, src, lay, d.src_col, d.width, packed=filled, async_op=True)
                           for d, src, dst, lay in self._exchanges[k + 1]]
                opts.part = _lib.PART_REST
                _lib.check(lib.pea_model_forward_part(self._h, k, C.byref(opts), ptrs, _lib.ptr(keep[0]), _lib.ptr(att_t), m,
                                                      _lib.ptr(self._ws), self.workspace_bytes, _lib.ptr(out),
                                                      _lib.ptr(stack), _lib.current_stream()))
                shard.wait_all(pending)
        if gather:
            shard.allgather_rows(out)
            if want_stack:
                shard.allgather_rows(stack)
        if select_ids is not None:
            return out, picked
        return (out, stack) if want_stack else out

    # ------------------------------------------------------------------ sharded loss step, launches replayed from hipGraphs
    def _param_table(self, layer_params):
        sig = tuple(None if t is None else t.data_ptr() for lp in layer_params for t in lp)
        cached = getattr(self, '_ptr_cache', None)
        if cached is None or cached[0] != sig:
            return None, sig
        return cached[1], sig

    def sharded_loss(self, layer_params, x, att, triples, fc1_w, fc1_b, fc2_w, fc2_b):
        """One rank's share of  loss = model.loss(batch)  without autograd (reference models/base.py:43-48 in training
        mode: full-graph forward, then the BPR term of the batch) on a sharded plan, with every launch sequence between two
        collectives recorded ONCE on a launch tape (csrc/tape.hip; or captured into a hipGraph, PEA_SHARD_REPLAY=graph) and
        replayed while the argument pointers stay the same (parameters and
        x are updated in place by an optimizer; the batch is copied into a static id buffer):

            graph[stage 0, rows other ranks read] -> all-gather(s) of the next level's sources, asynchronous
            graph[stage 0, remaining rows]        -> stream waits for the all-gathers
            graph[last stage + fusion + the batch's rows] -> all-reduce of [3B, repr_dim] -> graph[scorer + BPR sum]

        A rank of 8 on the 25m-shaped graph issues ~10 launches of 10-90 us each; enqueueing them one by one through
        ctypes took as long as running them.  Returns (loss, fused rows of this rank [N, repr_dim]); both are static
        buffers that the next call overwrites (the loss is cloned).  PEA_SHARD_REPLAY=eager or GRAPHS_ENABLED = False (bench.py's
        per-kernel event pass): the same sequence, eager."""
        lib = _lib.load()
        shard = self.plan.layout
        n, b = self.plan.num_nodes, triples.shape[0]
        dev = x.device
        st = getattr(self, '_sl', None)
        if st is None or st['b'] != b:
            st = self._sl = {
                'b': b, 'ids': torch.empty(3 * b, dtype=torch.int64, device=dev),
                'out': torch.empty((n, self.repr_dim), dtype=torch.float32, device=dev),
                'picked': torch.empty((3 * b, self.repr_dim), dtype=torch.float32, device=dev),
                'local': torch.arange(3 * b, dtype=torch.int64, device=dev).view(b, 3),
                'loss': torch.empty((), dtype=torch.float32, device=dev),
                'ws': torch.zeros(int(lib.pea_bpr_workspace_bytes(b)), dtype=torch.uint8, device=dev),
                'key': None, 'graphs': None}
            # (the scorer reads local positions 0 .. 3B-1 of `picked`; node ids are validated where they are used, in the
            # fusion launch's row selection: ShardLayout._err_flag, checked by check_pending_errors like a bad BPR triple)
        t3 = triples[:, :3]
        st['ids'].view(b, 3).copy_(t3)
        att_t = att.detach().reshape(self.P, self.repr_dim) if self.channel_aggr == 'att' else None
        if att_t is not None and not att_t.is_contiguous():
            att_t = att_t.contiguous()
        fc = [t.detach() for t in (fc1_w, fc1_b, fc2_w, fc2_b)]
        if any(not t.is_contiguous() for t in fc):
            fc = [t.contiguous() for t in fc]
        ptrs, sig = self._param_table(layer_params)
        if ptrs is None:         # first call / a parameter moved: the eager forward validates and rebuilds the pointer table
            self.forward(layer_params, x, att=att, gather=False, out=st['out'])
            ptrs, sig = self._param_table(layer_params)
            if ptrs is None:
                raise ValueError('sharded_loss needs contiguous float32 parameters')
            st['key'] = None
        xd = x.detach()
        flag = shard._err_flag(dev)
        stream = _lib.current_stream

        def part(k, which, select):
            opts = _lib.StageOpts(which, None, 1, 0, None, None)
            if select:
                opts.sel_ids, opts.sel_stride, opts.n_sel = st['ids'].data_ptr(), 1, 3 * b
                opts.sel_out, opts.err_flag = st['picked'].data_ptr(), flag.data_ptr()
            _lib.check(lib.pea_model_forward_part(self._h, k, C.byref(opts), ptrs, _lib.ptr(xd), _lib.ptr(att_t), -1,
                                                  _lib.ptr(self._ws), self.workspace_bytes, _lib.ptr(st['out']), None, stream()))

        def score():
            _lib.check(lib.pea_bpr_score(b, self.repr_dim, 3 * b, _lib.ptr(st['picked']), _lib.ptr(st['local']), 3,
                                         _lib.ptr(fc[0]), _lib.ptr(fc[1]), _lib.ptr(fc[2]), _lib.ptr(fc[3]), None, None,
                                         _lib.ptr(st['loss']), _lib.ptr(st['ws']), st['ws'].numel(), stream()))

        last = self.n_stages - 1
        steps = []                                   # (name, callable) of every launch sequence between two collectives
        for k in range(self.n_stages):
            if k < last:
                steps.append((('src', k), lambda k=k: part(k, _lib.PART_SOURCES, False)))
                steps.append((('rest', k), lambda k=k: part(k, _lib.PART_REST, False)))
            else:
                steps.append((('last', k), lambda k=k: part(k, _lib.PART_ALL, True)))
        steps.append((('score', 0), score))
        # replay mode: 'tape' (default: the library's launch tape, csrc/tape.hip), 'graph' (hipGraphs: slower on the GPU
        # side, kept for comparison), 'eager' (PEA_SHARD_REPLAY=eager, or while bench.py times every launch with events)
        mode = _shard_replay_mode() if GRAPHS_ENABLED else 'eager'
        key = (mode, sig, xd.data_ptr(), None if att_t is None else att_t.data_ptr(), tuple(t.data_ptr() for t in fc),
               self._ws.data_ptr())
        if mode != 'eager' and st['key'] != key:
            for _, fn in steps:                      # warm-up on the real stream: lazy plan state (GCN norms ...) is built here
                fn()
            torch.cuda.synchronize()
            recs = {}
            for name, fn in steps:
                if mode == 'graph':
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        fn()
                else:
                    g = _lib.Tape()
                    with g.record():
                        fn()
                recs[name] = g
            st['key'], st['graphs'] = key, recs
        run = (lambda name, fn: fn()) if mode == 'eager' else (lambda name, fn: st['graphs'][name].replay())
        it = iter(steps)
        for k in range(self.n_stages):
            if k < last:
                name, fn = next(it)
                run(name, fn)
                filled = self._fills[k]
                pending = [shard.exchange_sources(dst, src, lay, d.src_col, d.width, packed=filled, async_op=True)
                           for d, src, dst, lay in self._exchanges[k + 1]]
                name, fn = next(it)
                run(name, fn)
                shard.wait_all(pending)
            else:
                name, fn = next(it)
                run(name, fn)
        shard.reduce_rows_(st['picked'])
        name, fn = next(it)
        run(name, fn)
        return st['loss'].clone(), st['out']

    def forward_graphed(self, layer_params, x, att=None, masked=None):
        """forward() with the launches of the schedule captured ONCE into a hipGraph and replayed while the argument
        pointers stay the same (parameters are updated in place by the optimizer, so a training loop keeps replaying):
        for launch-bound sizes (MovieLens latest-small: ~13 launches of 10-40 us each).  Unsharded, no stack.  The
        returned tensor is a static buffer that the next call overwrites."""
        if self.sharded:
            raise NotImplementedError('the graphed forward is single-GPU')
        key = (x.data_ptr(), None if att is None else att.data_ptr(), masked,
               tuple(None if t is None else t.data_ptr() for lp in layer_params for t in lp))
        if getattr(self, '_graph_key', None) != key:
            self._graph_out = self.forward(layer_params, x, att=att, masked=masked).clone()   # warm-up: attributes, lazy state
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.forward(layer_params, x, att=att, masked=masked, out=self._graph_out)
            self._graph, self._graph_key = graph, key
        self._graph.replay()
        return self._graph_out

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.load().pea_model_destroy(h)
            except Exception:
                pass


def _rows2d(t):
    if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
        raise ValueError('expected a CUDA float32 [rows, cols] view with unit column stride')
    return t


_gw_ws = {}


class RowSet:
    """Rows of a table that hold a non-zero, on the device: flags uint8 [N], ids int32 [N] (the first `count` valid, ascending),
    count int32 [1] -- the host never reads the count (csrc/rows.hip).  Buffers are allocated once and refilled."""

    def __init__(self, n, device):
        lib = _lib.require_device()
        self.n = int(n)
        self.flags = torch.zeros(self.n, dtype=torch.uint8, device=device)
        self.ids = torch.zeros(self.n, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self._ws = torch.empty(int(lib.pea_rows_nonzero_workspace_bytes(self.n)), dtype=torch.uint8, device=device)

    def zero_rows_of(self, table, width):
        """table[ids[q], 0:width] = 0 for the listed rows."""
        table = _rows2d(table)
        _lib.check(_lib.load().pea_rows_zero(_lib.ptr(table), table.stride(0), int(width), _lib.ptr(self.ids), _lib.ptr(self.count),
                                             _lib.current_stream()))

    def fill_from(self, table, width, also=None):
        """flags / ids / count of the rows of table [N, ld] whose first `width` columns hold a non-zero -- or that are flagged
        in `also` (uint8 [N])."""
        table = _rows2d(table)
        _lib.check(_lib.load().pea_rows_nonzero_or(self.n, int(width), _lib.ptr(table), table.stride(0), _lib.ptr(also),
                                                   _lib.ptr(self.flags), _lib.ptr(self.ids), _lib.ptr(self.count),
                                                   _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()))
        return self


def grad_weight(pairs, shard=None, rows=None):
    """[a_q^T b_q for (a_q, b_q) in pairs]: a_q [N, ma], b_q [N, nb] float32 views (row strides free) over the same N
    rows -- the weight gradients of one level in one pair of launches (pea_grad_weight: fixed-order reduction).  A pair may
    carry (mask uint8 [N], alt [N, nb]): rows flagged in mask take their b operand from alt (include/peahip.h, pea_gw_job).
    shard = (rank, world, tile): this rank's SHARE (the sum over the rows it owns); the caller all-reduces."""
    lib = _lib.require_device()
    if not pairs:
        return []
    n = pairs[0][0].shape[0]
    dev = pairs[0][0].device
    jobs = (_lib.GwJob * len(pairs))()
    outs = []
    for q, item in enumerate(pairs):
        a, b = _rows2d(item[0]), _rows2d(item[1])
        mask, alt = (item[2], _rows2d(item[3])) if len(item) > 2 else (None, None)     # rows flagged in mask read alt instead of b
        alt_scale = item[4] if len(item) > 4 else None                                  # ... times alt_scale[row] (float32 [N])
        if a.shape[0] != n or b.shape[0] != n:
            raise ValueError('grad_weight: operands of one call must share the row count')
        out = torch.empty((a.shape[1], b.shape[1]), dtype=torch.float32, device=dev)
        outs.append(out)
        if mask is not None and (mask.dtype != torch.uint8 or mask.numel() != n or alt.shape != b.shape):
            raise ValueError('grad_weight: the row mask must be uint8 [N] and the alternative operand shaped like b')
        jobs[q] = _lib.GwJob(a.data_ptr(), a.stride(0), a.shape[1], b.data_ptr(), b.stride(0), b.shape[1],
                             out.data_ptr(), out.stride(0), None if mask is None else mask.data_ptr(),
                             None if alt is None else alt.data_ptr(), 0 if alt is None else alt.stride(0),
                             None if alt_scale is None else alt_scale.data_ptr())
    ws = _gw_ws.get(dev)
    if ws is None:
        ws = _gw_ws[dev] = torch.empty(int(lib.pea_grad_weight_workspace_bytes()), dtype=torch.uint8, device=dev)
    if rows is not None:        # RowSet: the reduction runs over the listed rows only (the others contribute exact zeros)
        _lib.check(lib.pea_grad_weight_rows(n, _lib.ptr(rows.ids), _lib.ptr(rows.count), rows.n, len(pairs), jobs, _lib.ptr(ws),
                                            ws.numel(), _lib.current_stream()))
    elif shard is not None and shard[1] > 1:
        _lib.check(lib.pea_grad_weight_sharded(n, int(shard[2]), int(shard[1]), int(shard[0]), len(pairs), jobs, _lib.ptr(ws),
                                               ws.numel(), _lib.current_stream()))
    else:
        _lib.check(lib.pea_grad_weight(n, len(pairs), jobs, _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    return outs


def dense_batch(triples_, rows=None):
    """out_q = a_q @ w_q for (a_q [N, k], w_q [k, n_out], out_q [N, n_out] view) in triples_: one launch (the input
    gradients dIn = dT W of one level); k, n_out and a's row stride must be multiples of 4.  rows (int32 device tensor):
    only those rows are computed (the rows a rank owns in a sharded training step).  A fourth element gate_q [N, n_out]
    (k <= 128) zeroes the outputs where gate_q <= 0: the relu mask of the layer below, applied in the epilogue."""
    lib = _lib.require_device()
    if not triples_:
        return
    n = triples_[0][0].shape[0]
    jobs = (_lib.DenseJob * len(triples_))()
    keep = []
    for q, item in enumerate(triples_):
        a, w, out = item[:3]
        gate = item[3] if len(item) > 3 else None
        a, out = _rows2d(a), _rows2d(out)
        w = _rows2d(w if w.stride(1) == 1 else w.contiguous())
        keep.append(w)
        if a.shape != (n, w.shape[0]) or out.shape != (n, w.shape[1]):
            raise ValueError('dense_batch: shapes %s @ %s -> %s' % (tuple(a.shape), tuple(w.shape), tuple(out.shape)))
        g_ptr, g_ld = None, 0
        if gate is not None:
            gate = _rows2d(gate)
            if gate.shape != out.shape:
                raise ValueError('dense_batch: gate %s for output %s' % (tuple(gate.shape), tuple(out.shape)))
            g_ptr, g_ld = gate.data_ptr(), gate.stride(0)
        jobs[q] = _lib.DenseJob(a.data_ptr(), a.stride(0), w.shape[0], w.data_ptr(), w.stride(0), w.shape[1],
                                out.data_ptr(), out.stride(0), g_ptr, g_ld)
    if rows is not None:
        _lib.check(lib.pea_dense_batch_rows(rows.numel(), _lib.ptr(rows), len(triples_), jobs, _lib.current_stream()))
    else:
        _lib.check(lib.pea_dense_batch(n, len(triples_), jobs, _lib.current_stream()))


_m2b_ws = {}


def mlp2_backward_data(chans, emb, hid, out, dt1, h, dz, da, rows=None, weights_in_out=False):
    """Both products of the first layer's backward data path in one launch (csrc/mlp2_bwd.hip): for every channel
    (w0 [hid, emb], w1 [out, hid], dt1_col, h_col, dz_col, da_col) of `chans`:  dz = (dt1 . w1) where h > 0 else 0;
    da = dz . w0.  dt1, h, dz, da: float32 [N, ld] views of the training workspace.  rows (RowSet): the listed rows only."""
    lib = _lib.require_device()
    n = dt1.shape[0]
    arr = (_lib.Mlp2BwdChan * len(chans))()
    keep = []
    for q, (w0, w1, c1, ch, cz, ca) in enumerate(chans):
        w0, w1 = w0.detach(), w1.detach()
        if not w0.is_contiguous():
            w0 = w0.contiguous()
        if not w1.is_contiguous():
            w1 = w1.contiguous()
        keep += [w0, w1]
        arr[q] = _lib.Mlp2BwdChan(w0.data_ptr(), w1.data_ptr(), int(c1), int(ch), int(cz), int(ca))
    need = int(lib.pea_mlp2_backward_data_workspace_bytes(len(chans), int(emb), int(hid), int(out)))
    if need == 0:
        raise ValueError('mlp2_backward_data: unsupported widths (%d, %d, %d)' % (emb, hid, out))
    key = (dt1.device, need)
    ws = _m2b_ws.get(key)
    if ws is None:
        ws = _m2b_ws[key] = torch.empty(need, dtype=torch.uint8, device=dt1.device)
    _lib.check(lib.pea_mlp2_backward_data(n, len(chans), arr, int(emb), int(hid), int(out), _lib.ptr(dt1), dt1.stride(0),
                                          _lib.ptr(h), h.stride(0), _lib.ptr(dz), dz.stride(0), _lib.ptr(da), da.stride(0),
                                          None if rows is None else _lib.ptr(rows.ids), None if rows is None else _lib.ptr(rows.count),
                                          1 if weights_in_out else 0, _lib.ptr(ws), ws.numel(), _lib.current_stream()))


def mlp2_backward_data_sage(chans, emb, hid, out, dt1, dr1, h, dz, dm, dxr, rows=None):
    """SAGE form of mlp2_backward_data (pea_mlp2_backward_data_sage): per channel (w_rel0 [hid, emb], w_root0, w_rel1 [out, hid],
    w_root1, dt1_col, dr1_col, h_col, dz_col, dm_col, dxr_col):  dz = (dt1 . w_rel1 + dr1 . w_root1) where h > 0 else 0;
    dm = dz . w_rel0;  dxr = dz . w_root0."""
    lib = _lib.require_device()
    n = dt1.shape[0]
    arr = (_lib.Mlp2BwdChan * len(chans))()
    arr_s = (_lib.Mlp2BwdChanSage * len(chans))()
    keep = []
    for q, (w0, w0r, w1, w1r, c1, cr, ch, cz, cm, cx) in enumerate(chans):
        ws_ = [t.detach() if t.is_contiguous() else t.detach().contiguous() for t in (w0, w0r, w1, w1r)]
        keep += ws_
        arr[q] = _lib.Mlp2BwdChan(ws_[0].data_ptr(), ws_[2].data_ptr(), int(c1), int(ch), int(cz), int(cm))
        arr_s[q] = _lib.Mlp2BwdChanSage(ws_[1].data_ptr(), ws_[3].data_ptr(), int(cr), int(cx))
    need = int(lib.pea_mlp2_backward_data_workspace_bytes(len(chans), int(emb), int(hid), int(out)))
    if need == 0:
        raise ValueError('mlp2_backward_data_sage: unsupported widths (%d, %d, %d)' % (emb, hid, out))
    key = (dt1.device, need)
    ws = _m2b_ws.get(key)
    if ws is None:
        ws = _m2b_ws[key] = torch.empty(need, dtype=torch.uint8, device=dt1.device)
    _lib.check(lib.pea_mlp2_backward_data_sage(n, len(chans), arr, arr_s, int(emb), int(hid), int(out), _lib.ptr(dt1), dt1.stride(0),
                                               _lib.ptr(dr1), dr1.stride(0), _lib.ptr(h), h.stride(0), _lib.ptr(dz), dz.stride(0),
                                               _lib.ptr(dm), dm.stride(0), _lib.ptr(dxr), dxr.stride(0),
                                               None if rows is None else _lib.ptr(rows.ids),
                                               None if rows is None else _lib.ptr(rows.count), _lib.ptr(ws), ws.numel(),
                                               _lib.current_stream()))


def block_sum(src, n_blocks, width):
    """[N, width] = sum of the n_blocks side-by-side column blocks of src [N, >= n_blocks * width] (fixed order)."""
    lib = _lib.require_device()
    src = _rows2d(src)
    out = torch.empty((src.shape[0], width), dtype=torch.float32, device=src.device)
    _lib.check(lib.pea_block_sum(src.shape[0], int(n_blocks), int(width), _lib.ptr(src), src.stride(0), _lib.ptr(out), out.stride(0),
                                 _lib.current_stream()))
    return out


_pending_err = []      # error flags of bpr_score calls that have not been read back yet (one int32 view each)


def check_pending_errors():
    """Raise IndexError if a bpr_score call since the last check saw a node id outside [0, num_nodes) -- what the
    reference raises at `cached_repr[unids]` (models/base.py:209-210).  The flag lives on the device, so reading it is
    a stream synchronize: it is read where the host synchronizes anyway (predict, rank_eval, model.eval(),
    bpr_score(validate=True)); until then the loss of such a batch is NaN, never a silently smaller sum."""
    global _pending_err
    flags, _pending_err = _pending_err, []
    if flags and int(torch.stack([f.reshape(-1)[0] for f in flags]).max().item()) != 0:
        for f in flags:
            if f.numel() == 1 and f.dim() == 1:
                f.zero_()          # a persistent flag (sharding.ShardLayout) is re-armed
        raise IndexError('index out of range in BPR triples')


def bpr_score(repr_, triples, fc1_w, fc1_b, fc2_w, fc2_b, want_preds=False, validate=False):
    """loss = -sum(log(sigmoid(pos - neg))) over rows (u, i+, i-) of `triples` (reference models/base.py:46-48,
    208-214).  Returns a 0-dim tensor (and pos, neg [B] when asked).  A triple with a node id out of range makes the
    loss NaN on the device and IndexError at the next check_pending_errors() (validate=True: at once)."""
    lib = _lib.require_device()
    if triples.dtype != torch.int64 or triples.dim() != 2 or triples.shape[1] < 3:
        raise ValueError('triples must be int64 [B, >=3]')
    if triples.stride(1) != 1:
        triples = triples.contiguous()
    b = triples.shape[0]
    r = repr_.shape[1]
    dev = repr_.device
    ws_bytes = int(lib.pea_bpr_workspace_bytes(b))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    pos = torch.empty(b, dtype=torch.float32, device=dev) if want_preds else None
    neg = torch.empty(b, dtype=torch.float32, device=dev) if want_preds else None
    args = [t.detach().contiguous() for t in (repr_, fc1_w, fc1_b, fc2_w, fc2_b)]
    _lib.check(lib.pea_bpr_score(b, r, repr_.shape[0], _lib.ptr(args[0]), _lib.ptr(triples), triples.stride(0),
                                 _lib.ptr(args[1]), _lib.ptr(args[2]), _lib.ptr(args[3]), _lib.ptr(args[4]),
                                 _lib.ptr(pos), _lib.ptr(neg), _lib.ptr(loss), _lib.ptr(ws), ws_bytes,
                                 _lib.current_stream()))
    if len(_pending_err) >= 64:          # bounded: a long async loop never grows the list
        check_pending_errors()
    _pending_err.append(ws[:4].view(torch.int32)[0])
    if validate:
        check_pending_errors()
    return (loss, pos, neg) if want_preds else loss


class _EntityReg(torch.autograd.Function):
    """reg(x) with the gradient rows of the same launch (csrc/entity.hip); backward = one index_add into dx."""

    @staticmethod
    def forward(ctx, x, batch9):
        reg, rows = _entity_launch(x, batch9, want_grad=True)
        ctx.save_for_backward(rows, batch9)
        ctx.shape = x.shape
        return reg

    @staticmethod
    def backward(ctx, g):
        rows, t = ctx.saved_tensors
        ids = t[:, [1, 3, 4, 0, 6, 7]].reshape(-1)            # order of the six gradient rows per batch row
        dx = torch.zeros(ctx.shape, dtype=rows.dtype, device=rows.device)
        dx.index_add_(0, ids, rows * g)
        return dx, None


def _entity_launch(x, batch9, want_grad):
    lib = _lib.require_device()
    if batch9.dtype != torch.int64 or batch9.dim() != 2 or batch9.shape[1] < 9:
        raise ValueError('entity-aware batches are int64 [B, 9] (u, i+, i-, 6 entity columns)')
    if batch9.stride(1) != 1:
        batch9 = batch9.contiguous()
    xd = x.detach()
    if xd.stride(1) != 1 or xd.stride(0) % 4:
        xd = xd.contiguous()
    b, f = batch9.shape[0], xd.shape[1]
    ws_bytes = int(lib.pea_entity_reg_workspace_bytes(b, f))
    if ws_bytes == 0:
        raise ValueError('entity_reg: emb_dim %d must be a multiple of 4, <= 256' % f)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xd.device)
    reg = torch.empty((), dtype=torch.float32, device=xd.device)
    rows = torch.empty((6 * b, f), dtype=torch.float32, device=xd.device) if want_grad else None
    _lib.check(lib.pea_entity_reg(b, f, xd.shape[0], _lib.ptr(xd), xd.stride(0), _lib.ptr(batch9), batch9.stride(0),
                                  _lib.ptr(reg), _lib.ptr(rows), _lib.ptr(ws), ws_bytes, _lib.current_stream()))
    if len(_pending_err) >= 64:
        check_pending_errors()
    _pending_err.append(ws[:4].view(torch.int32)[0])
    return reg, rows


def entity_reg(x, batch9):
    """Entity-aware regulariser (reference models/base.py:50-73) in one HIP launch; differentiable with respect to x
    when x requires grad (gradient rows come out of the same launch)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _EntityReg.apply(x, batch9)
    return _entity_launch(x, batch9, want_grad=False)[0]


def predict(repr_, unids, inids, fc1_w, fc1_b, fc2_w, fc2_b):
    """fc2(relu(fc1([repr[u] || repr[i]]))) -> [B, 1]  (reference models/base.py:208-214)."""
    lib = _lib.require_device()
    check_pending_errors()
    unids = unids.to(torch.int64).contiguous()
    inids = inids.to(torch.int64).contiguous()
    if unids.shape != inids.shape or unids.dim() != 1:
        raise ValueError('unids / inids must be 1-d of equal length')
    b = unids.shape[0]
    out = torch.empty(b, dtype=torch.float32, device=repr_.device)
    args = [t.detach().contiguous() for t in (repr_, fc1_w, fc1_b, fc2_w, fc2_b)]
    rc = lib.pea_predict(b, repr_.shape[1], repr_.shape[0], _lib.ptr(args[0]), _lib.ptr(unids), _lib.ptr(inids),
                         _lib.ptr(args[1]), _lib.ptr(args[2]), _lib.ptr(args[3]), _lib.ptr(args[4]), _lib.ptr(out),
                         _lib.current_stream())
    if rc == -2:
        raise IndexError(_lib.last_error())
    _lib.check(rc)
    return out.view(-1, 1)


def rank_eval(repr_, unids, cand, fc1_w, fc1_b, fc2_w, fc2_b):
    """Batched evaluator (reference solvers.py:56-96): cand [U, C], column 0 = the held-out positive.
    Returns scores [U, C], rank of the positive [U] (int32), auc [U], eval loss [U]."""
    lib = _lib.require_device()
    check_pending_errors()
    unids = unids.to(torch.int64).contiguous()
    cand = cand.to(torch.int64).contiguous()
    u, c = cand.shape
    dev = repr_.device
    scores = torch.empty((u, c), dtype=torch.float32, device=dev)
    rank = torch.empty(u, dtype=torch.int32, device=dev)
    auc = torch.empty(u, dtype=torch.float32, device=dev)
    loss = torch.empty(u, dtype=torch.float32, device=dev)
    args = [t.detach().contiguous() for t in (repr_, fc1_w, fc1_b, fc2_w, fc2_b)]
    rc = lib.pea_rank_eval(u, c, repr_.shape[1], repr_.shape[0], _lib.ptr(args[0]), _lib.ptr(unids), _lib.ptr(cand),
                           _lib.ptr(args[1]), _lib.ptr(args[2]), _lib.ptr(args[3]), _lib.ptr(args[4]),
                           _lib.ptr(scores), _lib.ptr(rank), _lib.ptr(auc), _lib.ptr(loss), _lib.current_stream())
    if rc == -2:
        raise IndexError(_lib.last_error())
    _lib.check(rc)
    return scores, rank, auc, loss


def bpr_train_raw(picked, att, fc1_w, fc1_b, fc2_w, fc2_b):
    """One launch of csrc/bpr_train.hip + the two fixed-order reductions (pea_grad_weight): returns
    (loss, grad_rows [3B, P*R], (d_att | None, d_fc1_w, d_fc1_b, d_fc2_w, d_fc2_b)) for picked [3B, P, R]."""
    lib = _lib.require_device()
    n3, p_, r_ = picked.shape
    b = n3 // 3
    rows = picked.detach()
    if not rows.is_contiguous():
        rows = rows.contiguous()
    dev = rows.device
    keep = [t.detach().contiguous() for t in (fc1_w, fc1_b, fc2_w, fc2_b)]
    att_c = att.detach().contiguous().view(-1) if att is not None else None
    grad_rows = torch.empty((n3, p_ * r_), dtype=torch.float32, device=dev)
    dhx = torch.empty((2 * b, r_ + 4), dtype=torch.float32, device=dev)
    zx = torch.empty((2 * b, 3 * r_ + 4), dtype=torch.float32, device=dev)
    p4 = (p_ + 3) // 4 * 4
    dsc = torch.empty((n3, p4), dtype=torch.float32, device=dev) if att is not None else None
    ws_bytes = int(lib.pea_bpr_train_workspace_bytes(b))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _lib.check(lib.pea_bpr_train(b, p_, r_, _lib.ptr(rows), p_ * r_, _lib.ptr(att_c), _lib.ptr(keep[0]), _lib.ptr(keep[1]),
                                 _lib.ptr(keep[2]), _lib.ptr(keep[3]), _lib.ptr(loss), _lib.ptr(grad_rows), _lib.ptr(dhx),
                                 _lib.ptr(zx), _lib.ptr(dsc), _lib.ptr(ws), ws_bytes, _lib.current_stream()))
    g = grad_weight([(dhx, zx)])[0]                    # [R + 4, 3R + 4]
    d_att = None
    if att is not None:
        full = grad_weight([(dsc, rows.view(n3, p_ * r_))])[0]      # [P4, P * R]: the diagonal R-blocks are d att[p]
        d_att = torch.stack([full[q, q * r_:(q + 1) * r_] for q in range(p_)]).view(att.shape)
    return loss, grad_rows, (d_att, g[:r_, :2 * r_], g[:r_, 3 * r_], g[r_, 2 * r_:3 * r_].view(fc2_w.shape),
                             g[r_, 3 * r_].view(fc2_b.shape))


def rows_scatter_sum(ids, src, num_channels, repr_dim, col_of_channel, dst):
    """dst[id, col_of_channel[p] + c] = sum_{k: ids[k] == id} src[k, p * R + c] in increasing k (ids < 0 skipped): the
    deterministic index backward of rows = stack[ids] into the node-indexed gradient buffer (an LDS sort of the batch's
    ids + one wave per node); at most ROWS_SCATTER_MAX positions."""
    lib = _lib.require_device()
    cols = (C.c_int * num_channels)(*[int(c) for c in col_of_channel])
    ws_bytes = int(lib.pea_rows_scatter_sum_workspace_bytes(ids.numel()))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=src.device)
    _lib.check(lib.pea_rows_scatter_sum(ids.numel(), _lib.ptr(ids), _lib.ptr(src), src.stride(0), int(num_channels), int(repr_dim),
                                        cols, _lib.ptr(dst), dst.stride(0), dst.shape[0], _lib.ptr(ws), ws_bytes, _lib.current_stream()))


ROWS_SCATTER_MAX = 16384      # positions pea_rows_scatter_sum sorts in LDS (a BPR batch of up to 5461 triples)


class _BprTrainLoss(torch.autograd.Function):
    """loss = -sum_b log sigmoid(score(u_b, i+_b) - score(u_b, i-_b)) over the fused rows of the batch's stack rows, and its
    whole backward, in one HIP launch (csrc/bpr_train.hip) + two fixed-order reductions for the parameter gradients
    (pea_grad_weight).  Reference: models/base.py:193-203 (fusion), :208-214 (fc1 / fc2 scorer), :46-48 (BPR loss) under
    loss.backward() (solvers.py:213-214)."""

    @staticmethod
    def forward(ctx, picked, att, fc1_w, fc1_b, fc2_w, fc2_b):
        loss, grad_rows, head = bpr_train_raw(picked, att, fc1_w, fc1_b, fc2_w, fc2_b)
        ctx.grads = (grad_rows.view(picked.shape),) + head
        return loss

    @staticmethod
    def backward(ctx, g):
        return tuple(None if t is None else t * g for t in ctx.grads)


def bpr_train_supported(num_channels, repr_dim):
    return bool(_lib.require_device().pea_bpr_train_supported(int(num_channels), int(repr_dim)))


def bpr_train_loss(picked, att, fc1_w, fc1_b, fc2_w, fc2_b):
    """Differentiable BPR loss of a batch from its stack rows picked [3B, P, R] (rows 3b, 3b+1, 3b+2 = user, positive,
    negative of triple b): channel fusion ('att' with the [.., P, R] attention tensor, 'mean' with att=None), the fc1 / fc2
    scorer and the loss, forward and backward in HIP."""
    return _BprTrainLoss.apply(picked, att, fc1_w, fc1_b, fc2_w, fc2_b)
