"""Randomised parity sweep (GPU box): random heterogeneous graphs (skewed degrees, hubs, multi-edges, empty relations),
random widths / heads / step counts, optionally with the source-sliced layout forced on small graphs, HIP path against
the CPU oracle (and float64 where fp32 orders legitimately differ).  python profiles/tools/fuzz_parity.py [N] [seed]
FUZZ_TWOSTEP=1: only configurations the two-step inference schedule takes (every channel 2 steps, one head, emb / hidden
64 or 128, repr <= 32 -- <= 16 for SAGE): csrc/mlp2.hip and the first-layer aggregation of x."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def one(rng, i):
    kind = ['gat', 'gcn', 'sage'][rng.integers(0, 3)]
    heads = int(rng.choice([1, 1, 2, 4])) if kind == 'gat' else 1
    n = int(rng.integers(20, 3000))
    emb = 4 * int(rng.integers(1, 33))
    hidden = 4 * int(rng.integers(1, 33))
    repr_dim = 4 * int(rng.integers(1, 9))
    n_ch = int(rng.integers(1, 6))
    twostep = os.environ.get('FUZZ_TWOSTEP') == '1'
    if twostep:
        heads, emb, hidden = 1, int(rng.choice([64, 128])), int(rng.choice([64, 128]))
        repr_dim = 4 * int(rng.integers(1, 5 if kind == 'sage' else 9))
        n_ch = int(rng.integers(1, 12))
    rels = []
    for _ in range(int(rng.integers(1, 4))):
        e = int(rng.choice([0, 5, 200, 3000, 30000]))
        hot = int(rng.integers(1, max(2, n // 4)))
        dst = rng.integers(0, n, e)
        if rng.random() < 0.6 and e:                       # hubs: most edges land on a few destinations
            dst = np.where(rng.random(e) < 0.8, rng.integers(0, hot, e), dst)
        rels.append(np.stack([rng.integers(0, n, e), dst]).astype(np.int64))
    steps, edges = [], []
    for _ in range(n_ch):
        s = 2 if twostep else int(rng.integers(1, 4))
        if kind == 'gat' and heads > 1 and s == 1:
            s = 2                                          # a 1-step GAT channel only stacks with one head (reference)
        steps.append(s)
        edges.append([rels[rng.integers(0, len(rels))] if rng.random() < 0.7 else
                      np.ascontiguousarray(rels[rng.integers(0, len(rels))][::-1]) for _ in range(s)])
    aggr = 'att' if rng.random() < 0.7 else 'mean'
    sliced = rng.random() < 0.4
    if sliced:
        os.environ['PEA_SLICE_MIN_EDGES'], os.environ['PEA_SLICE_BYTES'] = '500', str(int(rng.choice([512, 4096, 20000])))
    else:
        os.environ.pop('PEA_SLICE_MIN_EDGES', None), os.environ.pop('PEA_SLICE_BYTES', None)
    from test_gpu_edge_cases import _check
    desc = '%d: %s heads %d n %d emb %d hid %d repr %d steps %s aggr %s sliced %s edges %s' % (
        i, kind, heads, n, emb, hidden, repr_dim, steps, aggr, sliced, [r.shape[1] for r in rels])
    try:
        _check(kind, n, edges, steps, emb, hidden * 1, repr_dim, heads=heads, aggr=aggr, seed=int(rng.integers(0, 1000)))
        return True, desc
    except Exception:
        return False, desc + '\n' + traceback.format_exc(limit=3)


if __name__ == '__main__':
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 50
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
