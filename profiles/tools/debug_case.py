"""python profiles/tools/debug_case.py CASE [SEED]: reproduce configuration CASE of `fuzz_parity.py N SEED` and take it apart
layer by layer: every conv layer of channel 0 is run on the SAME fp32 input (the float64 truth of the layer below, rounded) on
the HIP drop-in, the fp32 CPU oracle and in float64; prints each side's error and the worst rows with their in-degree and
largest |logit|."""
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
    from oracle import oracle as orc
    from oracle import pyg_restatement as R
    case = int(sys.argv[1])
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    grabbed = {}

    def grab(kind, n, edges, steps, emb, hidden, repr_dim, heads=1, aggr='att', seed=3):
        grabbed.update(kind=kind, n=n, edges=edges, steps=steps, emb=emb, hidden=hidden, repr_dim=repr_dim, heads=heads, aggr=aggr, seed=seed)

    T._check = grab
    for i in range(case + 1):
        ok, desc = fuzz_parity.one(rng, i)
    print(desc)
    g = grabbed
    model = T.build_model(g['kind'], g['n'], g['edges'], g['steps'], g['emb'], g['hidden'], g['repr_dim'], heads=g['heads'], channel_aggr=g['aggr'])
    model.load_state_dict(T.random_state_dict(model, g['seed']))
    model.eval()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    # ---- end to end: the whole stack on each side against float64; and the HIP path once more with another (equally
    # valid) evaluation order -- source slicing off -- to tell a systematic excess from two realisations of rounding noise
    from helpers import f64_forward
    cps, hls = [], []
    for pp, SS in enumerate(g['steps']):
        cps.append([{k[len('pea_channels.%d.gnn_layers.%d.' % (pp, s_)):]: v for k, v in sd.items()
                     if k.startswith('pea_channels.%d.gnn_layers.%d.' % (pp, s_))} for s_ in range(SS)])
        hls.append([g['heads']] * (SS - 1) + [1] if SS > 1 else [g['heads']])
    _, wstack = orc.pea_forward(g['kind'], sd['x'], g['edges'], cps, hls, att=sd.get('att'), channel_aggr=g['aggr'], return_stack=True)
    _, t_stack = f64_forward(g['kind'], sd, g['edges'], g['steps'], g['heads'], g['aggr'])
    with torch.no_grad():
        _, stack_a = model.forward(return_stack=True)
    env = {k: os.environ.pop(k, None) for k in ('PEA_SLICE_MIN_EDGES', 'PEA_SLICE_BYTES')}
    model_b = T.build_model(g['kind'], g['n'], g['edges'], g['steps'], g['emb'], g['hidden'], g['repr_dim'], heads=g['heads'], channel_aggr=g['aggr'])
    model_b.load_state_dict(model.state_dict())
    model_b.eval()
    with torch.no_grad():
        _, stack_b = model_b.forward(return_stack=True)
    for k, v in env.items():
        if v is not None:
            os.environ[k] = v
    R_ = g['repr_dim']
    t2 = t_stack.reshape(-1, R_)
    sc = np.abs(t2).max(1) + 1e-30
    ea = np.abs(stack_a.cpu().numpy().astype(np.float64).reshape(-1, R_) - t2).max(1) / (EPS * sc)
    eb = np.abs(stack_b.cpu().numpy().astype(np.float64).reshape(-1, R_) - t2).max(1) / (EPS * sc)
    eo = np.abs(wstack.astype(np.float64).reshape(-1, R_) - t2).max(1) / (EPS * sc)
    for nm, e in (('hip (sliced: %s)' % bool(env.get('PEA_SLICE_BYTES')), ea), ('hip (unsliced)', eb), ('oracle', eo)):
        print('END TO END %-20s err / (eps x row): mean %.2f p90 %.1f p99 %.1f max %.1f' % (nm, e.mean(), np.percentile(e, 90), np.percentile(e, 99), e.max()))
    print('rows with hip > 2 oracle + 16: sliced %d, unsliced %d; oracle > 2 hip + 16: %d / %d;  corr(log hip_sliced, log hip_unsliced) %.3f, corr(log hip, log oracle) %.3f'
          % (int((ea > 2 * eo + 16).sum()), int((eb > 2 * eo + 16).sum()), int((eo > 2 * ea + 16).sum()), int((eo > 2 * eb + 16).sum()),
             np.corrcoef(np.log(ea + 1e-3), np.log(eb + 1e-3))[0, 1], np.corrcoef(np.log(ea + 1e-3), np.log(eo + 1e-3))[0, 1]))
    p, S = 0, g['steps'][0]
    h64 = torch.from_numpy(sd['x']).double()
    for s in range(S):
        pre = 'pea_channels.%d.gnn_layers.%d.' % (p, s)
        lp = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        ei = g['edges'][p][s]
        last = s == S - 1
        hh = 1 if (S > 1 and last) else g['heads']
        x32 = h64.float()
        conv64 = R.GATConv(x32.shape[1], lp['lin.weight'].shape[0] // hh, heads=hh).double()
        conv64.load_state_dict({k: torch.from_numpy(v).double() for k, v in lp.items()})
        with torch.no_grad():
            t = conv64(x32.double(), torch.from_numpy(ei))
            layer = model.pea_channels[p].gnn_layers[s]
            got = layer(x32.cuda(), torch.from_numpy(ei).cuda()).cpu().double()
        o32 = torch.from_numpy(orc.conv('gat', x32.numpy(), ei, lp, hh)).double()
        f = t.shape[1] // hh
        tt, gg, oo = (v.reshape(-1, f).numpy() for v in (t, got, o32))
        sc = np.abs(tt).max(1) + 1e-30
        eh, eo = np.abs(gg - tt).max(1) / (EPS * sc), np.abs(oo - tt).max(1) / (EPS * sc)
        deg = np.bincount(ei[1][ei[0] != ei[1]], minlength=g['n'])
        hlin = (x32.double() @ torch.from_numpy(lp['lin.weight']).double().t()).reshape(g['n'], hh, f)
        a_src = (hlin * torch.from_numpy(lp['att_j']).double().reshape(1, hh, f)).sum(-1).abs().max().item()
        a_dst = (hlin * torch.from_numpy(lp['att_i']).double().reshape(1, hh, f)).sum(-1).abs().max().item()
        print('layer %d: in %d heads %d f %d | input max %.2f | max |a_src| %.1f |a_dst| %.1f | err / (eps x row): hip mean %.2f p99 %.1f max %.1f | oracle mean %.2f p99 %.1f max %.1f'
              % (s, x32.shape[1], hh, f, float(x32.abs().max()), a_src, a_dst, eh.mean(), np.percentile(eh, 99), eh.max(), eo.mean(), np.percentile(eo, 99), eo.max()))
        worst = np.argsort(-eh)[:6]
        for w in worst:
            print('    vector %d (node %d head %d): hip %.1f oracle %.1f eps, in-degree %d, row scale %.3f' % (w, w // hh, w % hh, eh[w], eo[w], deg[w // hh], sc[w]))
        h64 = torch.relu(t) if not last else t


if __name__ == '__main__':
    main()
