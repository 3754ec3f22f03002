"""How accurate is the GAT softmax aggregation alone?  python profiles/tools/softmax_error.py [N] [seed]
Single GATConv layers with lin.weight = identity (the transform is then exact: h = x), so the error against float64 is
that of the logits, the softmax weights and the weighted sum only.  Random graphs with hubs, heads 1-4, attention vectors
scaled so that logits reach |e| ~ 5 ... 60.  Prints, per case and overall, the error of the HIP path and of the fp32 CPU
oracle (oracle/pea_oracle.c, the reference's op order) against float64, as multiples of fp32 eps x the row's magnitude.
PEA_LIB=<other build> compares two builds of the library (A/B of a kernel change)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_recsys_benchmark_amd import nn as hnn  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import pyg_restatement as ref  # noqa: E402

EPS = 2.0 ** -24


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    tot_h, tot_o, worse2 = [], [], 0
    rows_total = 0
    for i in range(count):
        heads = int(rng.choice([1, 2, 4]))
        f = 4 * int(rng.integers(2, 17))
        n = int(rng.integers(200, 4000))
        e = int(rng.choice([2000, 20000, 100000]))
        dst = rng.integers(0, n, e)
        if rng.random() < 0.7:
            dst = np.where(rng.random(e) < 0.85, rng.integers(0, max(1, n // 50), e), dst)
        ei = np.stack([rng.integers(0, n, e), dst]).astype(np.int64)
        att_scale = float(rng.choice([0.3, 1.0, 3.0]))
        torch.manual_seed(int(rng.integers(0, 10000)))
        a = hnn.GATConv(heads * f, f, heads=heads)
        with torch.no_grad():
            a.lin.weight.copy_(torch.eye(heads * f))
            a.att_i.uniform_(-att_scale, att_scale)
            a.att_j.uniform_(-att_scale, att_scale)
            a.bias.zero_()
        b = ref.GATConv(heads * f, f, heads=heads)
        b.load_state_dict(a.state_dict())
        x = torch.randn(n, heads * f)
        with torch.no_grad():
            got = a.cuda().eval()(x.cuda(), torch.from_numpy(ei).cuda()).cpu().double().numpy()
            want = b.double().eval()(x.double(), torch.from_numpy(ei)).numpy()
        lp = {k: v.detach().cpu().numpy() for k, v in a.state_dict().items()}
        o32 = orc.conv('gat', x.numpy(), ei, lp, heads).astype(np.float64)
        # per (row, head) vector: max error over the vector, in units of eps * max |value| of the vector
        g, w, o = (t.reshape(n * heads, f) for t in (got, want, o32))
        scale = np.abs(w).max(axis=1) + 1e-30
        eh, eo = np.abs(g - w).max(axis=1) / (EPS * scale), np.abs(o - w).max(axis=1) / (EPS * scale)
        logit = float(np.abs((x.double().numpy().reshape(n, heads, f) * lp['att_j'].reshape(1, heads, f)).sum(-1)).max())
        tot_h.append(eh)
        tot_o.append(eo)
        bad = int((eh > 2.0 * eo + 4.0).sum())
        worse2 += bad
        rows_total += eh.size
        print('%3d heads %d f %3d n %4d e %6d att %.1f |a_src| max %5.1f: hip mean %.2f max %.1f | oracle mean %.2f max %.1f | rows hip > 2 x oracle + 4 eps: %d'
              % (i, heads, f, n, e, att_scale, logit, eh.mean(), eh.max(), eo.mean(), eo.max(), bad), flush=True)
    eh, eo = np.concatenate(tot_h), np.concatenate(tot_o)
    print('ALL (%d vectors), error / (eps x vector magnitude): hip mean %.3f p99 %.2f max %.1f | oracle mean %.3f p99 %.2f max %.1f | hip > 2 x oracle + 4 eps in %d vectors'
          % (eh.size, eh.mean(), np.percentile(eh, 99), eh.max(), eo.mean(), np.percentile(eo, 99), eo.max(), worse2))


if __name__ == '__main__':
    main()
