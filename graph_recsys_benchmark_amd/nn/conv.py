"""Drop-in GATConv / GCNConv / SAGEConv backed by the gfx950 HIP kernels.

Same constructor signatures, parameter names and `forward(x, edge_index) -> [N, heads*out]` contract as the
torch-geometric 1.5.0 classes the reference instantiates at
    graph_recsys_benchmark/models/peagat.py:16-21, peagcn.py:16-21, peasage.py:16-21
and calls at models/base.py:138-139, so the reference's checkpoints load with strict=True
(state_dict leaves: GAT lin.weight/att_i/att_j/bias, GCN weight/bias, SAGE lin_rel.{weight,bias}/lin_root.weight).

These per-layer modules run ONE conv per call (plan cached per edge_index tensor).  The fast path for a whole
PEA model is PEABaseRecsysModel.forward (models/base.py here), which hands all P x S layers to one schedule.

Training (reference solvers.py:213-216 with the reference's OWN channel loop, models/base.py:134-140): with autograd
enabled and any input requiring grad, forward() runs through _ConvFunction -- the HIP forward that keeps the softmax
statistics (pea_model_forward_train on a one-channel, one-step schedule over a plan that also holds the reversed
relation) and the HIP backward of autograd.backward_conv_stack (gradient gathers over the reversed relation, fixed-order
weight / bias / attention-vector reductions): gradients of x and of every parameter.
"""
import ctypes as C
import weakref

import torch
from torch.nn import Linear, Parameter

from .. import _lib
from ..engine import GraphPlan
from .inits import glorot, zeros

_plan_cache = {}


def _plan_for(edge_index, num_nodes, self_loops):
    """One GraphPlan per (edge_index tensor object, version, N, self-loop handling)."""
    key = (id(edge_index), edge_index._version, int(num_nodes), bool(self_loops))
    hit = _plan_cache.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    plan = GraphPlan(num_nodes, [[edge_index]], self_loops)
    ref = weakref.ref(edge_index, lambda _r, k=key: _plan_cache.pop(k, None))
    _plan_cache[key] = (ref, plan)
    return plan


_train_plan_cache = {}


def _train_plan_for(edge_index, num_nodes, self_loops):
    """Like _plan_for, with the reversed relation planned too (the backward's gathers walk it)."""
    key = (id(edge_index), edge_index._version, int(num_nodes), bool(self_loops))
    hit = _train_plan_cache.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    plan = GraphPlan(num_nodes, [[edge_index]], self_loops, with_reverse=True)
    ref = weakref.ref(edge_index, lambda _r, k=key: _train_plan_cache.pop(k, None))
    _train_plan_cache[key] = (ref, plan)
    return plan


def _wants_grad(module, x):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))


class _ConvFunction(torch.autograd.Function):
    """One conv layer, differentiable: forward and backward in HIP on a one-channel, one-step PEAEngine."""

    @staticmethod
    def forward(ctx, engine, x, width, *params):
        from ..autograd import _Layout, _view
        lay = getattr(engine, '_layout', None)
        if lay is None:
            lay = engine._layout = _Layout(engine)
        engine.forward([tuple(params)], x, att=None, train=True, out=engine._scratch_out)
        out = _view(engine._wsf, lay.off_x, engine.plan.num_nodes, lay.ld_x)[:, :width].clone()
        ctx.engine, ctx.width = engine, width
        ctx.save_for_backward(x, *[t for t in params if t is not None])
        ctx.present = [t is not None for t in params]
        return out

    @staticmethod
    def backward(ctx, d_out):
        from ..autograd import backward_conv_stack
        saved = list(ctx.saved_tensors)
        x, it = saved[0], iter(saved[1:])
        params = tuple(next(it) if p else None for p in ctx.present)
        n = x.shape[0]
        with torch.no_grad():
            # the forward's statistics live in the engine's workspace: one backward per forward, in order (a module
            # called twice before backward -- weight sharing across calls -- would need a second workspace)
            dx, grads = backward_conv_stack(ctx.engine, d_out.contiguous().view(n, 1, ctx.width), x, [params])
        out = [None if t is None else g.reshape(t.shape) for t, g in zip(params, grads[0])]
        return (None, dx, None, *out)


def _train_engine(module, kind, plan, in_channels, out_channels, heads, **kw):
    """One training engine per (module, plan): a single channel of a single step, mean 'fusion' (identity for P = 1)."""
    from ..engine import PEAEngine
    cache = module.__dict__.setdefault('_train_engines', {})
    eng = cache.get(id(plan))
    if eng is None or eng.plan is not plan:
        eng = PEAEngine(plan, kind, [1], in_channels, out_channels, out_channels, heads=heads, channel_aggr='mean',
                        enable_backward=True, **kw)
        eng._scratch_out = torch.empty((plan.num_nodes, out_channels), dtype=torch.float32, device=plan.device)
        cache.clear()                       # one live engine per module: its workspace holds [N, .] training buffers
        cache[id(plan)] = eng
    return eng


def _autograd_x(x, in_channels):
    if not x.is_cuda:
        raise RuntimeError('the HIP conv path needs CUDA tensors (there is no CPU fallback)')
    if x.dim() != 2 or x.shape[1] != in_channels or x.dtype != torch.float32:
        raise ValueError('x must be float32 [N, %d], got %s %s' % (in_channels, x.dtype, tuple(x.shape)))
    return x.contiguous()


def _check_x(x, in_channels):
    if not x.is_cuda:
        raise RuntimeError('the HIP conv path needs CUDA tensors (there is no CPU fallback)')
    if x.dim() != 2 or x.shape[1] != in_channels or x.dtype != torch.float32:
        raise ValueError('x must be float32 [N, %d], got %s %s' % (in_channels, x.dtype, tuple(x.shape)))
    x = x.detach()
    if x.stride(1) != 1 or x.stride(0) % 4 != 0:
        x = x.contiguous()
    return x


def _workspace(plan, kind, in_channels, heads, out_channels, device):
    nbytes = int(_lib.load().pea_conv_workspace_bytes(plan._h, kind, 0, in_channels, heads, out_channels))
    if nbytes == 0:
        raise _lib.PeaError(-1, _lib.last_error())
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def _ptr(p, keep):
    """Device pointer of a parameter; a contiguous copy of a non-contiguous one (transposed / tied weight) is parked in
    `keep`, which the caller holds until the launch has been enqueued (stream order then protects the block)."""
    if p is None:
        return None
    t = p.detach().contiguous()
    keep.append(t)
    return C.c_void_p(t.data_ptr())


class GATConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.heads, self.concat = heads, concat
        self.negative_slope, self.dropout = negative_slope, dropout
        self.lin = Linear(in_channels, heads * out_channels, bias=False)
        self.att_i = Parameter(torch.Tensor(1, heads, out_channels))
        self.att_j = Parameter(torch.Tensor(1, heads, out_channels))
        if bias and concat:
            self.bias = Parameter(torch.Tensor(heads * out_channels))
        elif bias and not concat:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.lin.weight)
        glorot(self.att_i)
        glorot(self.att_j)
        zeros(self.bias)

    def forward(self, x, edge_index, relu=False):
        if self.training and self.dropout > 0:
            raise NotImplementedError('attention dropout > 0 is not implemented (p = 0 in every reference script)')
        if _wants_grad(self, x):
            x = _autograd_x(x, self.in_channels)
            plan = _train_plan_for(edge_index, x.shape[0], True)
            eng = _train_engine(self, 'gat', plan, self.in_channels, self.out_channels, self.heads,
                                negative_slope=float(self.negative_slope))
            fused_bias = self.bias if self.concat else None
            out = _ConvFunction.apply(eng, x, self.heads * self.out_channels, self.lin.weight, self.att_i, self.att_j,
                                      fused_bias)
            if not self.concat:
                out = out.view(x.shape[0], self.heads, self.out_channels).mean(dim=1)
                if self.bias is not None:
                    out = out + self.bias
            return torch.relu(out) if relu else out
        x = _check_x(x, self.in_channels)
        n = x.shape[0]
        plan = _plan_for(edge_index, n, True)
        hf = self.heads * self.out_channels
        out = torch.empty((n, hf), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(plan, _lib.KIND_GAT, self.in_channels, self.heads, self.out_channels, x.device)
        fused_bias = self.bias if self.concat else None
        keep = []
        _lib.check(_lib.load().pea_gat_conv(plan._h, 0, self.in_channels, self.heads, self.out_channels,
                                            _lib.ptr(x), x.stride(0), _ptr(self.lin.weight, keep), _ptr(self.att_i, keep),
                                            _ptr(self.att_j, keep), _ptr(fused_bias, keep), float(self.negative_slope),
                                            1 if (relu and self.concat) else 0, _lib.ptr(out), hf, _lib.ptr(ws), nbytes,
                                            _lib.current_stream()))
        if not self.concat:
            out = out.view(n, self.heads, self.out_channels).mean(dim=1)
            if self.bias is not None:
                out = out + self.bias.detach()
            if relu:
                out = torch.relu(out)
        return out

    def __repr__(self):
        return '{}({}, {}, heads={})'.format(self.__class__.__name__, self.in_channels, self.out_channels, self.heads)


class GCNConv(torch.nn.Module):
    """gcn_deg_from='row' reproduces PyG <= 1.5.0 (degree over the SOURCE index); 'col' is PyG >= 1.6."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, bias=True, normalize=True,
                 gcn_deg_from='row'):
        super().__init__()
        if improved or not normalize:
            raise NotImplementedError('improved=True / normalize=False are not used by the reference and not implemented')
        if gcn_deg_from not in ('row', 'col'):
            raise ValueError(gcn_deg_from)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.cached = cached            # the plan always caches the (static) normalisation
        self.gcn_deg_from = gcn_deg_from
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)

    def forward(self, x, edge_index, relu=False):
        if _wants_grad(self, x):
            x = _autograd_x(x, self.in_channels)
            plan = _train_plan_for(edge_index, x.shape[0], True)
            eng = _train_engine(self, 'gcn', plan, self.in_channels, self.out_channels, 1, gcn_deg_from=self.gcn_deg_from)
            out = _ConvFunction.apply(eng, x, self.out_channels, self.weight, self.bias)
            return torch.relu(out) if relu else out
        x = _check_x(x, self.in_channels)
        n = x.shape[0]
        plan = _plan_for(edge_index, n, True)
        out = torch.empty((n, self.out_channels), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(plan, _lib.KIND_GCN, self.in_channels, 1, self.out_channels, x.device)
        keep = []
        _lib.check(_lib.load().pea_gcn_conv(plan._h, 0, self.in_channels, self.out_channels, _lib.ptr(x), x.stride(0),
                                            _ptr(self.weight, keep), _ptr(self.bias, keep), 1 if self.gcn_deg_from == 'col' else 0,
                                            1 if relu else 0, _lib.ptr(out), self.out_channels, _lib.ptr(ws), nbytes,
                                            _lib.current_stream()))
        return out

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class SAGEConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, normalize=False, bias=True):
        super().__init__()
        if normalize:
            raise NotImplementedError('normalize=True is not used by the reference and not implemented')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.normalize = normalize
        self.lin_rel = Linear(in_channels, out_channels, bias=bias)
        self.lin_root = Linear(in_channels, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    def forward(self, x, edge_index, relu=False):
        if _wants_grad(self, x):
            x = _autograd_x(x, self.in_channels)
            plan = _train_plan_for(edge_index, x.shape[0], False)
            eng = _train_engine(self, 'sage', plan, self.in_channels, self.out_channels, 1)
            out = _ConvFunction.apply(eng, x, self.out_channels, self.lin_rel.weight, self.lin_rel.bias,
                                      self.lin_root.weight)
            return torch.relu(out) if relu else out
        x = _check_x(x, self.in_channels)
        n = x.shape[0]
        plan = _plan_for(edge_index, n, False)
        out = torch.empty((n, self.out_channels), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(plan, _lib.KIND_SAGE, self.in_channels, 1, self.out_channels, x.device)
        keep = []
        _lib.check(_lib.load().pea_sage_conv(plan._h, 0, self.in_channels, self.out_channels, _lib.ptr(x), x.stride(0),
                                             _ptr(self.lin_rel.weight, keep), _ptr(self.lin_rel.bias, keep),
                                             _ptr(self.lin_root.weight, keep), 1 if relu else 0, _lib.ptr(out),
                                             self.out_channels, _lib.ptr(ws), nbytes, _lib.current_stream()))
        return out

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)
