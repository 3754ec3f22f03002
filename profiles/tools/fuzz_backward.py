"""Randomised gradient sweep (GPU box): loss.backward() of the HIP path (conv stack forward + backward in HIP, batch-row
fusion / scorer) against float64 torch autograd of the restated model, on random graphs (hubs, empty relations, forced
source slicing), widths, heads and step counts.  python profiles/tools/fuzz_backward.py [N] [seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def one(rng, i):
    from helpers import build_model, random_state_dict
    from test_gpu_backward import f64_loss_and_grads
    kind = ['gat', 'gcn', 'sage'][rng.integers(0, 3)]
    heads = int(rng.choice([1, 1, 2])) if kind == 'gat' else 1
    n = int(rng.integers(30, 1500))
    emb, hidden, repr_dim = 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 5))
    rels = []
    for _ in range(int(rng.integers(1, 4))):
        e = int(rng.choice([0, 30, 800, 12000]))
        dst = rng.integers(0, n, e)
        if rng.random() < 0.6 and e:
            dst = np.where(rng.random(e) < 0.8, rng.integers(0, max(2, n // 20), e), dst)
        rels.append(np.stack([rng.integers(0, n, e), dst]).astype(np.int64))
    steps, edges = [], []
    for _ in range(int(rng.integers(1, 5))):
        s = int(rng.integers(1, 4))
        if kind == 'gat' and heads > 1 and s == 1:
            s = 2
        steps.append(s)
        edges.append([rels[rng.integers(0, len(rels))] if rng.random() < 0.7 else
                      np.ascontiguousarray(rels[rng.integers(0, len(rels))][::-1]) for _ in range(s)])
    aggr = 'att' if rng.random() < 0.7 else 'mean'
    sliced = rng.random() < 0.4
    if sliced:
        os.environ['PEA_SLICE_MIN_EDGES'], os.environ['PEA_SLICE_BYTES'] = '500', str(int(rng.choice([512, 4096])))
    else:
        os.environ.pop('PEA_SLICE_MIN_EDGES', None), os.environ.pop('PEA_SLICE_BYTES', None)
    desc = '%d: %s heads %d n %d emb %d hid %d repr %d steps %s aggr %s sliced %s edges %s' % (
        i, kind, heads, n, emb, hidden, repr_dim, steps, aggr, sliced, [r.shape[1] for r in rels])
    try:
        model = build_model(kind, n, edges, steps, emb, hidden, repr_dim, heads=heads, channel_aggr=aggr)
        model.load_state_dict(random_state_dict(model, int(rng.integers(0, 1000)), scale=0.25))
        b = int(rng.integers(1, 300))
        batch = np.stack([rng.integers(0, n, b), rng.integers(0, n, b), rng.integers(0, n, b)], axis=1).astype(np.int64)
        model.train()
        model.zero_grad()
        loss = model.loss(torch.from_numpy(batch).cuda())
        loss.backward()
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        want_loss, want = f64_loss_and_grads(kind, sd, edges, steps, heads, aggr, batch)
        assert abs(float(loss) - want_loss) <= 5e-5 * max(abs(want_loss), 1.0), 'loss %r vs %r' % (float(loss), want_loss)
        top0 = max(float(np.abs(w).max()) for w in want.values())
        worst = max(float(np.abs(p.grad.detach().cpu().numpy().astype(np.float64) - want[n_]).max()) /
                    (5e-4 * max(float(np.abs(want[n_]).max()), 1e-12) + 1e-6 * top0 + 1e-8) for n_, p in model.named_parameters())
        if os.environ.get('FUZZ_VERBOSE') and worst > 1.0:
            print('   batch', batch[:4].tolist(), 'rows', batch.shape[0], flush=True)
            for name, p in model.named_parameters():
                g, w = p.grad.detach().cpu().numpy().astype(np.float64), want[name]
                bad_rows = np.argwhere(np.abs(g - w) > 1e-4 * max(np.abs(w).max(), 1e-12))
                print('   %-44s err %.3e scale %.3e  (%d entries off, first %s)' % (name, np.abs(g - w).max(), np.abs(w).max(),
                                                                                 len(bad_rows), bad_rows[:4].tolist()), flush=True)
        top = max(float(np.abs(w).max()) for w in want.values())     # a gradient that is exactly 0 in float64 (e.g. d att_i when
        for name, p in model.named_parameters():                       # every logit of a row has one sign) is fp32 noise here
            g, w = p.grad.detach().cpu().numpy().astype(np.float64), want[name]
            scale = max(np.abs(w).max(), 1e-12)
            err = np.abs(g - w).max()
            assert err <= 5e-4 * scale + 1e-6 * top + 1e-8, '%s: max err %.3e vs scale %.3e (largest gradient %.3e)' % (name, err, scale, top)
        return True, desc
    except Exception:
        return False, desc + '\n' + traceback.format_exc(limit=2)


if __name__ == '__main__':
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(count):
        ok, desc = one(rng, i)
        if not ok:
            bad += 1
            print('FAIL', desc, flush=True)
        elif i % 5 == 0:
            print('ok  ', desc, flush=True)
    print('%d / %d failed' % (bad, count))
    sys.exit(1 if bad else 0)
