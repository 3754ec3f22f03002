"""Drop-in GATConv / GCNConv / SAGEConv backed by the gfx950 HIP kernels.

Same constructor signatures, parameter names and `forward(x, edge_index) -> [N, heads*out]` contract as the
torch-geometric 1.5.0 classes the reference instantiates at
    graph_recsys_benchmark/models/peagat.py:16-21, peagcn.py:16-21, peasage.py:16-21
and calls at models/base.py:138-139, so the reference's checkpoints load with strict=True
(state_dict leaves: GAT lin.weight/att_i/att_j/bias, GCN weight/bias, SAGE lin_rel.{weight,bias}/lin_root.weight).

These per-layer modules run ONE conv per call (plan cached per edge_index tensor).  The fast path for a whole
PEA model is PEABaseRecsysModel.forward (models/base.py here), which hands all P x S layers to one schedule.
Forward only for now: calling them with autograd enabled on parameters that require grad raises.
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


def _check_no_grad(module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            'the HIP conv path is forward-only so far (backward is the next row of SURVEY.md section 8f); '
            'call under torch.no_grad()')


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
        _check_no_grad(self)
        if self.training and self.dropout > 0:
            raise NotImplementedError('attention dropout > 0 is not implemented (p = 0 in every reference script)')
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
        _check_no_grad(self)
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
        _check_no_grad(self)
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
