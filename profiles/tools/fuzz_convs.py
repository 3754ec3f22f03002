"""Randomised check of the single-conv drop-ins (graph_recsys_benchmark_amd.nn GATConv / GCNConv / SAGEConv ->
pea_gat_conv / pea_gcn_conv / pea_sage_conv) against the torch restatement of the PyG 1.5.0 classes in float64
(oracle/pyg_restatement.py), on random graphs with hubs, self loops and multi-edges.  python profiles/tools/fuzz_convs.py [N] [seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(rng, i):
    from graph_recsys_benchmark_amd import nn as hnn
    from oracle import pyg_restatement as ref
    kind = ['gat', 'gcn', 'sage'][rng.integers(0, 3)]
    n = int(rng.integers(3, 5000))
    fin, fout = 4 * int(rng.integers(1, 40)), 4 * int(rng.integers(1, 40))
    heads = int(rng.choice([1, 1, 2, 3, 4])) if kind == 'gat' else 1
    if heads * fout > 256:
        fout = 4 * max(1, (256 // heads) // 4)
    concat = bool(rng.random() < 0.8)
    e = int(rng.choice([0, 1, 40, 2000, 60000]))
    dst = rng.integers(0, n, e)
    if rng.random() < 0.6 and e:
        dst = np.where(rng.random(e) < 0.85, rng.integers(0, max(1, n // 50), e), dst)
    src = rng.integers(0, n, e)
    if rng.random() < 0.3 and e:
        src[: e // 10] = dst[: e // 10]                    # explicit self loops
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    relu = bool(rng.random() < 0.5)
    deg_from = 'row' if rng.random() < 0.5 else 'col'
    desc = '%d: %s n %d in %d out %d heads %d concat %s e %d relu %s deg_from %s' % (i, kind, n, fin, fout, heads, concat, e, relu, deg_from)
    try:
        torch.manual_seed(int(rng.integers(0, 10000)))
        if kind == 'gat':
            a, b = hnn.GATConv(fin, fout, heads=heads, concat=concat), ref.GATConv(fin, fout, heads=heads, concat=concat)
        elif kind == 'gcn':
            a, b = hnn.GCNConv(fin, fout, gcn_deg_from=deg_from), ref.GCNConv(fin, fout, gcn_deg_from=deg_from)
        else:
            a, b = hnn.SAGEConv(fin, fout), ref.SAGEConv(fin, fout)
        with torch.no_grad():
            for p in a.parameters():
                p.uniform_(-0.4, 0.4)
        b.load_state_dict(a.state_dict())
        x = torch.randn(n, fin)
        with torch.no_grad():
            got = a.cuda().eval()(x.cuda(), ei.cuda(), relu=relu).cpu().double()
            want = b.double().eval()(x.double(), ei)
            if relu:
                want = torch.relu(want)
            w32 = b.float().eval()(x, ei)
            if relu:
                w32 = torch.relu(w32)
        scale = float(want.abs().max()) + 1e-30
        err, err32 = float((got - want).abs().max()), float((w32.double() - want).abs().max())
        assert err <= 2.0 * err32 + 2e-6 * scale + 1e-7, 'max err %.3e (torch fp32 restatement: %.3e, scale %.3e)' % (err, err32, scale)
        return True, desc
    except Exception:
        return False, desc + '\n' + traceback.format_exc(limit=2)


if __name__ == '__main__':
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(count):
        ok, desc = one(rng, i)
        if not ok:
            bad += 1
            print('FAIL', desc, flush=True)
        elif i % 10 == 0:
            print('ok  ', desc, flush=True)
    print('%d / %d failed' % (bad, count))
    sys.exit(1 if bad else 0)
