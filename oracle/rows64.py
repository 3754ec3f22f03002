"""ORACLE / TEST INFRASTRUCTURE ONLY -- float64 recomputation of sampled destination rows of a 2-step PEA channel.

SURVEY.md 8, config 5: "parity spot-checked on sampled destination rows recomputed on CPU".  Used by tests/ and by
bench.py's checker legs where a full-size CPU oracle run is out of reach (stress preset) or is itself the less
accurate side (rows with millions of messages: the reference's fp32 scatter sums them sequentially).  Follows
oracle/pyg_restatement.py (the torch restatement of the PyG 1.5.0 convs) in float64 on the complete 2-hop
in-neighbourhood of the sampled rows; never imported by the product package.
"""
import numpy as np


def in_edges_of(edge_index, nodes):
    """Columns of the int64 [2, E] COO whose destination is in `nodes` (original order kept)."""
    mask = np.isin(edge_index[1], nodes)
    return edge_index[:, mask]


def f64_rows_two_step(kind, sd, p, rel1, rel2, rows, heads=1):
    """float64 value of channel p's output at the destination rows `rows`, computed on the 2-hop in-neighbourhood only
    ("parity spot-checked on sampled destination rows recomputed on CPU").  GAT and SAGE only:
    their per-row result depends on the complete in-neighbourhood of the row and nothing else (GCN's 1.5.0 degree is
    over the SOURCE index, i.e. global).  Returns [len(rows), R]."""
    import torch
    from oracle import pyg_restatement as R
    assert kind in ('gat', 'sage')
    rows = np.asarray(rows, dtype=np.int64)
    e2 = in_edges_of(rel2, rows)
    s1 = np.union1d(rows, e2[0])                              # rows whose layer-1 output is read
    e1 = in_edges_of(rel1, s1)
    s0 = np.union1d(s1, e1[0])                                # rows of x that are read
    remap = -np.ones(int(max(s0.max(), rows.max())) + 1, dtype=np.int64)
    remap[s0] = np.arange(s0.size)
    x = torch.from_numpy(sd['x'][s0]).double()

    def conv(step, h, ei, last):
        pre = 'pea_channels.%d.gnn_layers.%d.' % (p, step)
        lp = {k[len(pre):]: torch.from_numpy(v).double() for k, v in sd.items() if k.startswith(pre)}
        if kind == 'gat':
            hh = 1 if last else heads
            c = R.GATConv(h.shape[1], lp['lin.weight'].shape[0] // hh, heads=hh)
        else:
            c = R.SAGEConv(h.shape[1], lp['lin_rel.weight'].shape[0])
        c = c.double()
        c.load_state_dict(lp, strict=True)
        with torch.no_grad():
            return c(h, torch.from_numpy(remap[ei]))

    h1 = torch.relu(conv(0, x, e1, False))                    # exact on s1 (all their in-edges are present)
    out = conv(1, h1, e2, True)                               # exact on rows
    return out[torch.from_numpy(remap[rows])].numpy()


def f64_rows_one_step(kind, lp, x, rel, rows, heads=1):
    """float64 value of ONE conv layer (parameters `lp`: numpy arrays named like the state_dict leaves) at the destination
    rows `rows`, from the fp32 input `x` [N, F] and the complete in-neighbourhood of those rows under `rel`."""
    import torch
    from oracle import pyg_restatement as R
    assert kind in ('gat', 'sage')
    rows = np.asarray(rows, dtype=np.int64)
    e = in_edges_of(rel, rows)
    s0 = np.union1d(rows, e[0])
    remap = -np.ones(int(s0.max()) + 1, dtype=np.int64)
    remap[s0] = np.arange(s0.size)
    h = torch.from_numpy(np.asarray(x)[s0]).double()
    lp = {k: torch.from_numpy(np.asarray(v)).double() for k, v in lp.items()}
    if kind == 'gat':
        c = R.GATConv(h.shape[1], lp['lin.weight'].shape[0] // heads, heads=heads)
    else:
        c = R.SAGEConv(h.shape[1], lp['lin_rel.weight'].shape[0])
    c = c.double()
    c.load_state_dict(lp, strict=True)
    with torch.no_grad():
        out = c(h, torch.from_numpy(remap[e]))
    return out[torch.from_numpy(remap[rows])].numpy()
