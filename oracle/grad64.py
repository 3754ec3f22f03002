"""ORACLE / TEST INFRASTRUCTURE ONLY -- float64 gradients of one training-step loss at FULL graph size.

The reference's training step is `loss = model.loss(batch); loss.backward()` (solvers.py:213-215; loss =
models/base.py:43-48 on the full-graph forward :191-206).  The loss of a batch depends on the batch's rows of the fused
table only, i.e. on the complete 2-hop in-neighbourhood of those rows under every 2-step metapath: float64 torch autograd
through oracle/pyg_restatement.py on that induced subgraph therefore yields the EXACT full-size gradient of every
parameter (rows of x outside the neighbourhood have gradient zero), at a cost set by the batch, not by the graph.
GAT and SAGE only: a row's result depends on its own in-edges alone (GCN's PyG-1.5.0 degree runs over the SOURCE index,
i.e. over edges the induced subgraph drops).  Used by tests/ and by bench.py's checker leg; never imported by the product.
"""
import numpy as np


def _conv(R, kind, params, pre, in_w, heads):
    """A float64 restatement conv whose parameters ARE the given leaf tensors (so autograd accumulates into them)."""
    if kind == 'gat':
        conv = R.GATConv(in_w, params[pre + 'lin.weight'].shape[0] // heads, heads=heads).double()
        del conv._parameters['att_i'], conv._parameters['att_j'], conv._parameters['bias'], conv.lin._parameters['weight']
        conv.lin.weight = params[pre + 'lin.weight']
        conv.att_i, conv.att_j, conv.bias = params[pre + 'att_i'], params[pre + 'att_j'], params[pre + 'bias']
    else:
        conv = R.SAGEConv(in_w, params[pre + 'lin_rel.weight'].shape[0]).double()
        del conv.lin_rel._parameters['weight'], conv.lin_rel._parameters['bias'], conv.lin_root._parameters['weight']
        conv.lin_rel.weight, conv.lin_rel.bias = params[pre + 'lin_rel.weight'], params[pre + 'lin_rel.bias']
        conv.lin_root.weight = params[pre + 'lin_root.weight']
    return conv


def f64_subgraph_loss_and_grads(kind, sd, edges, batch, heads=1, aggr='att'):
    """(loss, {parameter name: float64 gradient}, rows of x that can carry a gradient) for a model of 2-step channels.
    sd: state_dict as numpy; edges: P lists of two int64 [2, E] arrays; batch: int64 [B, 3]."""
    import torch
    from oracle import pyg_restatement as R
    assert kind in ('gat', 'sage')
    params = {k: torch.from_numpy(np.asarray(v)).double().requires_grad_(True) for k, v in sd.items()}
    x = params['x']
    rows = np.unique(np.asarray(batch)[:, :3].reshape(-1)).astype(np.int64)
    n_nodes = x.shape[0]
    outs, touched = [], np.zeros(n_nodes, bool)
    for p, (rel1, rel2) in enumerate(edges):
        in_r0 = np.zeros(n_nodes, bool)
        in_r0[rows] = True
        e2 = rel2[:, in_r0[rel2[1]]]
        in_s1 = in_r0.copy()
        in_s1[e2[0]] = True                                    # rows whose layer-1 output is read
        e1 = rel1[:, in_s1[rel1[1]]]
        in_s0 = in_s1.copy()
        in_s0[e1[0]] = True                                    # rows of x that are read
        touched |= in_s0
        s0 = np.flatnonzero(in_s0)
        remap = np.full(n_nodes, -1, dtype=np.int64)
        remap[s0] = np.arange(s0.size)
        h = x.index_select(0, torch.from_numpy(s0))
        c0 = _conv(R, kind, params, 'pea_channels.%d.gnn_layers.0.' % p, h.shape[1], heads)
        h = torch.relu(c0(h, torch.from_numpy(remap[e1])))    # exact on S1 (all their in-edges are present)
        c1 = _conv(R, kind, params, 'pea_channels.%d.gnn_layers.1.' % p, h.shape[1], 1)
        h = c1(h, torch.from_numpy(remap[e2]))                # exact on the batch's rows
        outs.append(h.index_select(0, torch.from_numpy(remap[rows])))
    stack = torch.stack(outs, dim=1)                           # [len(rows), P, R]
    if aggr == 'att':
        w = torch.softmax((stack * params['att']).sum(-1), dim=-1).unsqueeze(-1)
        fused = (stack * w).sum(1)
    else:
        fused = stack.mean(1)
    pos = np.full(n_nodes, -1, dtype=np.int64)
    pos[rows] = np.arange(rows.size)
    b = torch.from_numpy(pos[np.asarray(batch)[:, :3]])

    def pred(u, i):
        z = torch.cat([fused[u], fused[i]], dim=-1)
        hdn = torch.relu(z @ params['fc1.weight'].t() + params['fc1.bias'])
        return hdn @ params['fc2.weight'].t() + params['fc2.bias']

    loss = -(pred(b[:, 0], b[:, 1]) - pred(b[:, 0], b[:, 2])).sigmoid().log().sum()
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in params.items()}
    return float(loss), grads, np.flatnonzero(touched)
