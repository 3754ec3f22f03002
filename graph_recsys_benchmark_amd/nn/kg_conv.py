"""Drop-in KGATConv / KGCNConv / NGCFConv (the reference's OWN MessagePassing subclasses, used by its KGAT / KGCN /
NGCF baseline models) with the gather -> weighted-sum half on the gfx950 kernels.

    graph_recsys_benchmark/nn/kgat_conv.py:10-54   forward(x, edge_index, att_map)
    graph_recsys_benchmark/nn/kgcn_conv.py:10-44   forward(x, edge_index, att_map)
    graph_recsys_benchmark/nn/ngcf_conv.py:10-48   forward(x, edge_index)

Same constructors, parameter names (weight_add / weight_bi / bias; weight / bias; W_1 / W_2) and initialisers.
The sparse half  aggr_i = sum_{e: j->i} w_e x_j  is pea_weighted_aggregate (w = att_map, or NGCF's degree
coefficient); the dense update that follows (two small GEMMs + elementwise) is ordinary torch, so autograd works:
the aggregate's backward is the same kernel over the reversed relation.
"""
import ctypes as C
import weakref

import torch
import torch.nn.functional as F
from torch.nn import Parameter

from .. import _lib
from .inits import glorot, zeros

_plans = {}


class _EdgePlan:
    """Forward + reversed CSR (with original edge ids) of one edge_index tensor."""

    def __init__(self, edge_index, num_nodes):
        lib = _lib.require_device()
        if not edge_index.is_cuda or edge_index.dtype != torch.int64 or edge_index.dim() != 2:
            raise ValueError('edge_index must be a CUDA int64 [2, E] tensor')
        if bool((edge_index[0] == edge_index[1]).any()):
            # the reference removes self loops from edge_index but not from att_map, so such input cannot work there
            raise ValueError('edge_index must not contain self loops')
        self.fwd = edge_index.contiguous()
        self.rev = torch.flip(edge_index, dims=[0]).contiguous()
        self.num_nodes, self.num_edges = int(num_nodes), int(edge_index.shape[1])
        ptrs = (C.c_void_p * 2)(self.fwd.data_ptr(), self.rev.data_ptr())
        nedge = (C.c_int64 * 2)(self.num_edges, self.num_edges)
        h = C.c_void_p()
        _lib.check(lib.pea_plan_create(self.num_nodes, 2, ptrs, nedge, _lib.PLAN_EDGE_IDS, 0, 0, 1, 256,
                                       _lib.current_stream(), C.byref(h)))
        self._h = h

    def aggregate(self, x, w, reverse=False):
        lib = _lib.load()
        x = x.contiguous()
        n, width = x.shape
        out = torch.empty_like(x)
        rel = 1 if reverse else 0
        nbytes = int(lib.pea_weighted_aggregate_workspace_bytes(self._h, rel, width))
        ws = torch.empty(max(nbytes, 512), dtype=torch.uint8, device=x.device)
        _lib.check(lib.pea_weighted_aggregate(self._h, rel, width, _lib.ptr(x), width, _lib.ptr(w.contiguous()),
                                              _lib.ptr(out), width, _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return out

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.load().pea_plan_destroy(h)
            except Exception:
                pass


def _plan_for(edge_index, num_nodes):
    key = (id(edge_index), edge_index._version, int(num_nodes))
    hit = _plans.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    plan = _EdgePlan(edge_index, num_nodes)
    _plans[key] = (weakref.ref(edge_index, lambda _r, k=key: _plans.pop(k, None)), plan)
    return plan


class _WeightedAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, plan):
        ctx.plan = plan
        ctx.save_for_backward(x, w)
        return plan.aggregate(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = ctx.plan.aggregate(g, w, reverse=True) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:      # d w_e = x_j . g_i  (rarely needed: att_map is computed without grad)
            ei = ctx.plan.fwd
            dw = (x.index_select(0, ei[0]) * g.index_select(0, ei[1])).sum(-1)
        return dx, dw, None


def weighted_aggregate(x, edge_index, w):
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.shape[1] % 4:
        raise ValueError('x must be a CUDA float32 [N, F] tensor with F a multiple of 4')
    w = w.reshape(-1).to(torch.float32)
    if w.numel() != edge_index.shape[1]:
        raise ValueError('one weight per edge expected')
    return _WeightedAggregate.apply(x, w, _plan_for(edge_index, x.shape[0]))


class KGATConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, negative_slope=0.2, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.negative_slope = in_channels, out_channels, negative_slope
        self.weight_add = Parameter(torch.Tensor(in_channels, out_channels))
        self.weight_bi = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight_add)
        glorot(self.weight_bi)
        zeros(self.bias)

    def forward(self, x, edge_index, att_map, size=None):
        aggr = weighted_aggregate(x, edge_index, att_map)
        add_aggr = F.leaky_relu(torch.mm(x + aggr, self.weight_add), negative_slope=self.negative_slope)
        bi_aggr = F.leaky_relu(torch.mm(x * aggr, self.weight_bi), negative_slope=self.negative_slope)
        out = add_aggr + bi_aggr
        if self.bias is not None:
            out = out + self.bias
        return out


class KGCNConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, negative_slope=0.2, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.negative_slope = in_channels, out_channels, negative_slope
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)

    def forward(self, x, edge_index, att_map, size=None):
        aggr = weighted_aggregate(x, edge_index, att_map)
        return F.relu(torch.mm(aggr + x, self.weight) + self.bias)


class NGCFConv(torch.nn.Module):
    """deg_div='true' divides the occurrence count by 2 as a float (torch >= 1.6); 'floor' reproduces torch 1.5.1,
    the reference's pin, where `long_tensor / 2` was an integer division (nn/ngcf_conv.py:39)."""

    def __init__(self, in_channels, out_channels, negative_slope=0.2, deg_div='true', **kwargs):
        super().__init__()
        if deg_div not in ('true', 'floor'):
            raise ValueError(deg_div)
        self.in_channels, self.out_channels, self.negative_slope = in_channels, out_channels, negative_slope
        self.deg_div = deg_div
        self.W_1 = Parameter(torch.Tensor(in_channels, out_channels))
        self.W_2 = Parameter(torch.Tensor(in_channels, out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.W_1)
        glorot(self.W_2)

    def _without_self_loops(self, edge_index):
        """remove_self_loops(edge_index), as the reference does before counting degrees and propagating
        (nn/ngcf_conv.py:33-34).  The filtered tensor is cached per input tensor so the plan cache keeps hitting."""
        key = (id(edge_index), edge_index._version)
        hit = getattr(self, '_noloop', None)
        if hit is not None and hit[0] == key and hit[1]() is edge_index:
            return hit[2]
        keep = edge_index[0] != edge_index[1]
        filtered = edge_index if bool(keep.all()) else edge_index[:, keep].contiguous()
        self._noloop = (key, weakref.ref(edge_index), filtered)
        return filtered

    def forward(self, x, edge_index, size=None):
        edge_index = self._without_self_loops(edge_index)
        if not hasattr(self, 'deg'):     # cached on first use like the reference (which counts in an O(N*E) loop)
            cnt = torch.bincount(edge_index.reshape(-1), minlength=x.shape[0])
            self.deg = (cnt // 2 if self.deg_div == 'floor' else cnt / 2).view(-1, 1)
        coff = 1 / torch.sqrt((self.deg[edge_index[1]] * self.deg[edge_index[0]]).float())
        s = weighted_aggregate(x, edge_index, coff.view(-1))
        return F.leaky_relu(torch.mm(x, self.W_1) + torch.mm(s, self.W_1) + torch.mm(x * s, self.W_2),
                            negative_slope=self.negative_slope)
