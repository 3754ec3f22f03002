"""Is the HIP path the looser side against the fp32 CPU oracle?  python profiles/tools/error_symmetry.py [N] [seed]
The fuzz configurations of fuzz_parity.py (random graphs / widths / heads / step counts), but instead of asserting, every
output vector (one node, one channel) of the channel stack is scored against float64 on BOTH sides:
    e_hip, e_orc = max |side - f64| over the vector;   scale = max |f64| over the vector
and the tool counts the vectors where one side is more than twice the other beyond the elementwise atol of the tests
(e_a > 2 e_b + 1e-6), in both directions, plus the error distributions in units of eps x scale.  Two equally accurate fp32
evaluation orders give equal counts; a looser kernel shows as an excess on its side."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
EPS = 2.0 ** -24


def main():
    import fuzz_parity
    import test_gpu_edge_cases as T
    from helpers import f64_forward
    from oracle import oracle as orc
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    stats = {}

    def check(kind, n, edges, steps, emb, hidden, repr_dim, heads=1, aggr='att', seed=3):
        model = T.build_model(kind, n, edges, steps, emb, hidden, repr_dim, heads=heads, channel_aggr=aggr)
        model.load_state_dict(T.random_state_dict(model, seed))
        model.eval()
        with torch.no_grad():
            _, stack = model.forward(return_stack=True)
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        cps, hls = [], []
        for p, S in enumerate(steps):
            cps.append([{k[len('pea_channels.%d.gnn_layers.%d.' % (p, s)):]: v for k, v in sd.items()
                         if k.startswith('pea_channels.%d.gnn_layers.%d.' % (p, s))} for s in range(S)])
            hls.append([1] * S if kind != 'gat' else ([heads] * (S - 1) + [1] if S > 1 else [heads]))
        _, wstack = orc.pea_forward(kind, sd['x'], edges, cps, hls, att=sd.get('att'), channel_aggr=aggr, return_stack=True)
        _, t_stack = f64_forward(kind, sd, edges, steps, heads, aggr)
        g = stack.cpu().numpy().astype(np.float64).reshape(-1, repr_dim)
        w, t = wstack.astype(np.float64).reshape(-1, repr_dim), t_stack.reshape(-1, repr_dim)
        eh, eo, sc = np.abs(g - t).max(1), np.abs(w - t).max(1), np.abs(t).max(1) + 1e-30
        st = stats.setdefault(kind, dict(n=0, hip_loose=0, orc_loose=0, eh=[], eo=[]))
        st['n'] += eh.size
        st['hip_loose'] += int((eh > 2 * eo + 1e-6).sum())
        st['orc_loose'] += int((eo > 2 * eh + 1e-6).sum())
        st['eh'].append(eh / (EPS * sc))
        st['eo'].append(eo / (EPS * sc))

    saved = T._check
    T._check = check
    try:
        for i in range(count):
            ok, desc = fuzz_parity.one(rng, i)
            if not ok:
                print('ERROR', desc, flush=True)
    finally:
        T._check = saved
    for kind, st in sorted(stats.items()):
        eh, eo = np.concatenate(st['eh']), np.concatenate(st['eo'])
        print('%-4s %9d vectors | hip > 2 x oracle + 1e-6: %6d | oracle > 2 x hip + 1e-6: %6d | err / (eps x scale): hip mean %.2f p99 %.1f p99.9 %.1f | oracle mean %.2f p99 %.1f p99.9 %.1f'
              % (kind, st['n'], st['hip_loose'], st['orc_loose'], eh.mean(), np.percentile(eh, 99), np.percentile(eh, 99.9),
                 eo.mean(), np.percentile(eo, 99), np.percentile(eo, 99.9)), flush=True)
    return stats


if __name__ == '__main__':
    main()
