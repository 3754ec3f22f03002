"""Randomised check of the scoring entry points (pea_bpr_score, pea_predict, pea_rank_eval) against float64 torch:
every repr width (multiples of 4 up to 64), batch sizes from 1, candidate counts across the 64-lane boundary, duplicate
ids and exact ties.  python profiles/tools/fuzz_scoring.py [N] [seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(rng, i):
    from graph_recsys_benchmark_amd import engine
    r = 4 * int(rng.integers(1, 17))
    n = int(rng.integers(2, 5000))
    b = int(rng.choice([1, 2, 63, 64, 65, 1000, 5000]))
    c = int(rng.choice([2, 3, 64, 65, 100, 129]))
    desc = '%d: R %d N %d B %d C %d' % (i, r, n, b, c)
    try:
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        rep = torch.randn(n, r, generator=g)
        if rng.random() < 0.3:
            rep[: n // 2] = rep[0]                                   # many identical rows: exact score ties
        w1, b1 = torch.randn(r, 2 * r, generator=g) * 0.3, torch.randn(r, generator=g) * 0.1
        w2, b2 = torch.randn(1, r, generator=g) * 0.3, torch.randn(1, generator=g) * 0.1
        t = torch.from_numpy(rng.integers(0, n, (b, 3)).astype(np.int64))
        cu = [x.cuda() for x in (rep, w1, b1, w2, b2)]

        def score64(u, it):
            z = torch.cat([rep[u], rep[it]], dim=-1).double()
            return (torch.relu(z @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()).squeeze(-1)

        loss, pos, neg = engine.bpr_score(cu[0], t.cuda(), *cu[1:], want_preds=True)
        p64, n64 = score64(t[:, 0], t[:, 1]), score64(t[:, 0], t[:, 2])
        want = -torch.nn.functional.logsigmoid(p64 - n64).sum()
        sc = float(max(p64.abs().max(), 1.0))
        assert float((pos.cpu().double() - p64).abs().max()) <= 2e-5 * sc and float((neg.cpu().double() - n64).abs().max()) <= 2e-5 * sc
        assert abs(float(loss) - float(want)) <= 2e-5 * max(abs(float(want)), 1.0), 'loss %r vs %r' % (float(loss), float(want))
        pr = engine.predict(cu[0], t[:, 0].cuda(), t[:, 1].cuda(), *cu[1:]).cpu().double().squeeze(-1)
        assert float((pr - p64).abs().max()) <= 2e-5 * sc
        # ranking: rank of column 0 = number of candidates scoring strictly higher in fp32 (the kernel's own scores)
        users = torch.from_numpy(rng.integers(0, n, b).astype(np.int64))
        cand = torch.from_numpy(rng.integers(0, n, (b, c)).astype(np.int64))
        scores, rank, auc, eloss = engine.rank_eval(cu[0], users.cuda(), cand.cuda(), *cu[1:])
        s = scores.cpu()
        s64 = score64(users[:, None].expand(b, c).reshape(-1), cand.reshape(-1)).view(b, c)
        assert float((s.double() - s64).abs().max()) <= 2e-5 * float(max(s64.abs().max(), 1.0))
        assert torch.equal(rank.cpu().long(), (s[:, 1:] > s[:, :1]).sum(dim=1))
        torch.testing.assert_close(auc.cpu(), (s[:, :1] > s[:, 1:]).sum(dim=1).float() / (c - 1), rtol=0, atol=1e-7)
        want_l = -torch.nn.functional.logsigmoid((s[:, :1] - s[:, 1:]).double()).sum(dim=1)
        assert float((eloss.cpu().double() - want_l).abs().max()) <= 2e-5 * float(max(want_l.abs().max(), 1.0))
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
