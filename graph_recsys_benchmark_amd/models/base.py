"""Host mirror of the reference's PEA model surface, with the arithmetic routed to the gfx950 HIP library.

Mirrors (names, kwargs, attributes, error behaviour) graph_recsys_benchmark/models/base.py:
    GraphRecsysModel      :29-96   loss() / eval() contract (BPR term :43-48, eval cache :88-96)
    PEABaseChannel        :129-140 relu between steps, none after the last
    PEABaseRecsysModel    :143-214 kwargs read at :148-152,156,167-179; forward / predict
Differences, all on purpose:
  * forward() hands ALL channels and steps to one schedule (engine.PEAEngine) instead of looping in Python;
  * graph tensors must be on the GPU; there is no CPU fallback (parity checks use oracle/ from the tests);
  * with autograd enabled (training), the conv stack runs through autograd.PEAStackFunction (HIP forward AND
    backward); the cheap fusion + scorer on top of it are plain differentiable torch ops;
  * model.shard(rank, world) (one process per GPU) shards forward, eval, loss AND the training step
    (zero_grad -> loss -> backward -> step, reference solvers.py:213-216) by destination rows; parameter gradients
    come out summed over the ranks, so every rank's optimizer takes the same step.
"""
import torch
from torch.nn import Parameter

from .. import engine as _engine
from ..autograd import PEALossFunction, PEAStackFunction, StackOptions
from ..nn.inits import glorot


class GraphRecsysModel(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self._init(**kwargs)
        self.reset_parameters()

    def _init(self, **kwargs):
        raise NotImplementedError

    def reset_parameters(self):
        raise NotImplementedError

    def update_graph_input(self, dataset):
        raise NotImplementedError

    def predict(self, unids, inids):
        raise NotImplementedError

    def loss(self, pos_neg_pair_t):
        """-sum(log(sigmoid(pos - neg))) over rows (u, i+, i-[, entity columns]); in training mode the full-graph
        forward is recomputed first, exactly like the reference (models/base.py:44-45)."""
        if self.training:
            self.cached_repr = self.forward()
        if self.cached_repr.requires_grad:      # differentiable path (same formula as the reference, models/base.py:46-48)
            pos_pred = self.predict(pos_neg_pair_t[:, 0], pos_neg_pair_t[:, 1])
            neg_pred = self.predict(pos_neg_pair_t[:, 0], pos_neg_pair_t[:, 2])
            cf_loss = -(pos_pred - neg_pred).sigmoid().log().sum()
        else:
            cf_loss = _engine.bpr_score(self.cached_repr, pos_neg_pair_t, self.fc1.weight, self.fc1.bias,
                                        self.fc2.weight, self.fc2.bias)
        if self.entity_aware and self.training:
            # entity-aware regulariser (models/base.py:50-73): one HIP launch (csrc/entity.hip)
            x = self.x if self.cached_repr.requires_grad else self.x.detach()
            return cf_loss + self.entity_aware_coff * _engine.entity_reg(x, pos_neg_pair_t)
        return cf_loss

    def eval(self, metapath_idx=None):
        """nn.Module.eval() + refresh of the cached full-graph representation.  Like the reference
        (models/base.py:88-96) the ablation index is honoured only by classes whose NAME starts with 'PEA'."""
        super().eval()
        self._repr_partial = False
        _engine.check_pending_errors()       # a bad BPR batch of the epoch raises here at the latest (IndexError)
        with torch.no_grad():
            if self.__class__.__name__[:3] == 'PEA':
                self.cached_repr = self.forward(metapath_idx)
            else:
                self.cached_repr = self.forward()
        return self


class _ShardedRows(torch.autograd.Function):
    """rows = table[ids] of a row-sharded [N, W] table: forward = the rows each rank owns summed over the ranks
    (ShardLayout.gather_rows); backward = the (replicated) gradient of those rows scattered into the rows THIS rank owns."""

    @staticmethod
    def forward(ctx, table, ids, layout):
        ctx.layout, ctx.shape = layout, table.shape
        ctx.save_for_backward(ids)
        return layout.gather_rows(table.detach(), ids)

    @staticmethod
    def backward(ctx, g):
        ids, = ctx.saved_tensors
        mine = ctx.layout.owner(ids) == ctx.layout.rank
        d = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        d.index_add_(0, ids[mine], g[mine])
        return d, None, None


class PEABaseChannel(torch.nn.Module):
    def reset_parameters(self):
        for module in self.gnn_layers:
            module.reset_parameters()

    def forward(self, x, edge_index_list):
        assert len(edge_index_list) == self.num_steps
        for step_idx in range(self.num_steps - 1):
            x = self.gnn_layers[step_idx](x, edge_index_list[step_idx], relu=True)   # conv + F.relu fused
        return self.gnn_layers[-1](x, edge_index_list[-1])


class PEABaseRecsysModel(GraphRecsysModel):
    kind = None   # 'gat' | 'gcn' | 'sage', set by the concrete model

    def _init(self, **kwargs):
        self.entity_aware = kwargs['entity_aware']
        self.entity_aware_coff = kwargs['entity_aware_coff']
        self.meta_path_steps = kwargs['meta_path_steps']
        self.if_use_features = kwargs['if_use_features']
        self.channel_aggr = kwargs['channel_aggr']
        self.gcn_deg_from = kwargs.get('gcn_deg_from', 'row')

        if not self.if_use_features:
            self.x = Parameter(torch.Tensor(kwargs['dataset']['num_nodes'], kwargs['emb_dim']))
        else:
            raise NotImplementedError('Feature not implemented!')

        meta_path_edge_index_list = self.update_graph_input(kwargs['dataset'])
        assert len(meta_path_edge_index_list) == len(kwargs['meta_path_steps'])
        self.meta_path_edge_index_list = meta_path_edge_index_list   # plain attribute, like the reference

        self.pea_channels = torch.nn.ModuleList()
        for num_steps in kwargs['meta_path_steps']:
            kwargs_cpy = kwargs.copy()
            kwargs_cpy['num_steps'] = num_steps
            self.pea_channels.append(kwargs_cpy['channel_class'](**kwargs_cpy))

        num_paths = len(kwargs['meta_path_steps'])
        if self.channel_aggr == 'att':
            self.att = Parameter(torch.Tensor(1, num_paths, kwargs['repr_dim']))
        if self.channel_aggr == 'cat':
            self.fc1 = torch.nn.Linear(2 * num_paths * kwargs['repr_dim'], kwargs['repr_dim'])
        else:
            self.fc1 = torch.nn.Linear(2 * kwargs['repr_dim'], kwargs['repr_dim'])
        self.fc2 = torch.nn.Linear(kwargs['repr_dim'], 1)
        self._dims = (kwargs['emb_dim'], kwargs['hidden_size'], kwargs['repr_dim'], kwargs.get('num_heads', 1))
        self._engine = None
        self._train_engine = None
        self._plan = None
        self._shard = (0, 1, 256)

    def shard(self, rank, world, tile=256):
        """Multi-GPU: this process computes the destination rows it owns (row i -> rank (i // tile) % world);
        torch.distributed must be initialised (backend 'nccl' = RCCL on the GPUs)."""
        self._shard = (int(rank), int(world), int(tile))
        self._engine = self._train_engine = self._plan = None
        if world > 1 and self.x.is_cuda:
            # RCCL: the in-place all-gather form of the exchanges is checked against the out-of-place one, once per process
            from ..sharding import ShardLayout
            ShardLayout.verify_inplace_all_gather(self.x.device)
        return self

    def reset_parameters(self):
        if not self.if_use_features:
            glorot(self.x)
        for module in self.pea_channels:
            module.reset_parameters()
        glorot(self.fc1.weight)
        glorot(self.fc2.weight)
        if self.channel_aggr == 'att':
            glorot(self.att)

    # ------------------------------------------------------------------ HIP schedule
    def _get_engine(self, train=False):
        if self.channel_aggr not in ('att', 'mean'):
            raise NotImplementedError('Other aggr methods not implemeted!')
        emb, hidden, repr_dim, heads = self._dims
        single = self._shard[1] == 1
        if self._plan is None:
            self._plan = _engine.GraphPlan(self.x.shape[0], self.meta_path_edge_index_list,
                                           self_loops=self.kind in ('gat', 'gcn'), shard_rank=self._shard[0],
                                           shard_world=self._shard[1], shard_tile=self._shard[2],
                                           gather_row_bytes=4 * hidden * (heads if self.kind == 'gat' else 1),
                                           with_reverse=True)         # reversed relations drive the backward gathers
        attr = '_train_engine' if train else '_engine'
        if getattr(self, attr) is None:
            setattr(self, attr, _engine.PEAEngine(self._plan, self.kind, self.meta_path_steps, emb, hidden, repr_dim,
                                                  heads=heads if self.kind == 'gat' else 1,
                                                  channel_aggr=self.channel_aggr, gcn_deg_from=self.gcn_deg_from,
                                                  enable_backward=train))
        return getattr(self, attr)

    def _layer_params(self):
        """Parameter tensors per conv layer in PARAM_SLOTS order.  The Parameter objects are looked up once (named_parameters
        over ~20 modules was 35 % of a step's host time on the launch-bound presets) and the list is rebuilt only when a
        module's parameter has been replaced by another object (every module._parameters entry is checked by identity)."""
        cache = getattr(self, '_lp_cache', None)
        if cache is not None and all(mod._parameters.get(name) is t for mod, name, t in cache[1]):
            return cache[0]
        slots = _engine.PARAM_SLOTS[self.kind]
        out, check = [], []
        for channel in self.pea_channels:
            for layer in channel.gnn_layers:
                sd = dict(layer.named_parameters())
                out.append(tuple(sd.get(name) for name in slots))
                for mod in layer.modules():                       # the layer and its Linear submodules
                    check.extend((mod, name, t) for name, t in mod._parameters.items())
        self._lp_cache = (out, check)
        return out

    def forward(self, metapath_idx=None, return_stack=False):
        if not self.x.is_cuda:
            raise RuntimeError('the HIP path needs the model on a GPU (there is no CPU fallback)')
        for channel in self.pea_channels:
            for layer in channel.gnn_layers:
                if self.training and getattr(layer, 'dropout', 0) > 0:
                    raise NotImplementedError('attention dropout > 0 is not implemented (p = 0 in every reference script)')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_autograd(metapath_idx, return_stack)
        eng = self._get_engine()
        return eng.forward(self._layer_params(), self.x.detach(), getattr(self, 'att', None), masked=metapath_idx,
                           want_stack=return_stack)

    def _forward_autograd(self, metapath_idx, return_stack):
        """Differentiable forward: conv stack forward + backward in HIP, fusion (models/base.py:194-203) in torch ops."""
        eng = self._get_engine(train=True)
        if eng.sharded:
            raise NotImplementedError('a sharded model is differentiated through model.loss(batch) (the rows the loss '
                                      'reads are exchanged there); forward() under autograd is single-GPU')
        flat = [t for lp in self._layer_params() for t in lp]
        stack = PEAStackFunction.apply(eng, self.x, eng.slots, None, *flat)
        out = self._fuse_torch(stack, metapath_idx)
        return (out, stack) if return_stack else out

    def _fuse_torch(self, x, metapath_idx=None):
        """Channel fusion (models/base.py:194-203) of [rows, P, R] in differentiable torch ops."""
        if metapath_idx is not None:
            keep = torch.ones(x.shape[1], dtype=x.dtype, device=x.device)
            keep[metapath_idx] = 0
            x = x * keep.view(1, -1, 1)
        if self.channel_aggr == 'mean':
            return x.mean(dim=1)
        atts = torch.softmax(torch.sum(x * self.att, dim=-1), dim=-1).unsqueeze(-1)
        return torch.sum(x * atts, dim=1)

    def _loss_autograd(self, t):
        """Differentiable training-step loss, single GPU.  The fusion and the scorer are row-local, so they are
        differentiated on the BATCH's rows of the channel stack only ([3B, P, R] instead of [N, P, R]: the dense torch
        fusion and its backward cost ~2 ms per step on the 25m-shaped graph); the conv stack below runs forward and
        backward in HIP over the whole graph as before.  Same loss as fusing everything and indexing afterwards
        (reference models/base.py:44-48).  cached_repr is the fused table of the same forward (HIP, detached)."""
        eng = self._get_engine(train=True)
        flat = [p for lp in self._layer_params() for p in lp]
        ids = t[:, :3].reshape(-1)                  # the stack is read at the batch's rows only: tell the backward
        opts = StackOptions(fuse_att=self.att.detach().reshape(eng.P, eng.repr_dim) if self.channel_aggr == 'att' else None,
                            read_ids=ids)
        if _engine.bpr_train_supported(eng.P, eng.repr_dim) and ids.numel() <= _engine.ROWS_SCATTER_MAX:
            # one autograd node for the whole step: conv stack + the batch's rows + fusion / scorer / loss, all HIP
            loss = PEALossFunction.apply(eng, self.x, eng.slots, opts, ids, self.att if self.channel_aggr == 'att' else None,
                                         self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, *flat)
            self.cached_repr, self._repr_partial = opts.fused, eng.sharded
            return loss
        stack = PEAStackFunction.apply(eng, self.x, eng.slots, opts, *flat)      # repr_dim > 32 or > 5461 triples: torch ops
        self.cached_repr, self._repr_partial = opts.fused, eng.sharded
        b = t.shape[0]
        if eng.sharded:
            # every rank holds the stack rows it owns: the batch's rows are summed from their owners (one all-reduce of
            # [3B, P * R], exact: x + 0), the loss is then computed replicated, and its gradient flows back into the
            # stack rows this rank owns only
            picked = _ShardedRows.apply(stack.view(stack.shape[0], -1), ids, eng.plan.layout).view(-1, eng.P, eng.repr_dim)
        else:
            picked = stack[ids]
        rows = self._fuse_torch(picked).view(b, 3, -1)

        def score(i):
            return self.fc2(torch.relu(self.fc1(torch.cat([rows[:, 0], rows[:, i]], dim=-1))))

        return -(score(1) - score(2)).sigmoid().log().sum()

    def loss(self, pos_neg_pair_t):
        """Sharded training-mode loss without autograd: every rank computes the rows it owns, then only the rows the
        batch names are exchanged (one all-reduce of [3B, repr_dim]) instead of all-gathering the [N, repr_dim] table;
        same value as the single-GPU loss.  cached_repr is completed lazily if predict() is called before eval()."""
        sharded = self._shard[1] > 1
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.training and grad and self.x.is_cuda:
            cf_loss = self._loss_autograd(pos_neg_pair_t)
            if self.entity_aware:
                return cf_loss + self.entity_aware_coff * self._entity_reg(pos_neg_pair_t, self.x)
            return cf_loss
        if not (sharded and self.training) or grad:
            return super().loss(pos_neg_pair_t)
        eng = self._get_engine()
        t = pos_neg_pair_t
        if t.dtype != torch.int64 or t.dim() != 2 or t.shape[1] < 3:
            raise ValueError('triples must be int64 [B, >=3]')
        # stage by stage with the exchanges in between; the batch's rows ride in the last stage's fusion launch (rows this
        # rank owns, zeros elsewhere), one all-reduce, then the scorer: engine.PEAEngine.sharded_loss
        cf_loss, part = eng.sharded_loss(self._layer_params(), self.x, getattr(self, 'att', None), t, self.fc1.weight,
                                         self.fc1.bias, self.fc2.weight, self.fc2.bias)
        self.cached_repr, self._repr_partial = part, True
        if self.entity_aware:
            return cf_loss + self.entity_aware_coff * self._entity_reg(t, self.x.detach())
        return cf_loss

    @staticmethod
    def _entity_reg(t, x):
        """entity-aware regulariser (reference models/base.py:50-73): squared L2 distances between raw x rows, one launch"""
        return _engine.entity_reg(x, t)

    def _complete_repr(self):
        if getattr(self, '_repr_partial', False):
            self._get_engine().plan.layout.allgather_rows(self.cached_repr)
            self._repr_partial = False

    def predict(self, unids, inids):
        self._complete_repr()
        if self.cached_repr.requires_grad:
            z = torch.cat([self.cached_repr[unids], self.cached_repr[inids]], dim=-1)
            return self.fc2(torch.relu(self.fc1(z)))
        return _engine.predict(self.cached_repr, unids, inids, self.fc1.weight, self.fc1.bias, self.fc2.weight,
                               self.fc2.bias)
