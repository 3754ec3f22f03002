"""CPU: the induced-subgraph float64 gradient checker (oracle/grad64.py, used at full graph size by the GPU tests and by
bench.py's training leg) equals float64 autograd over the WHOLE graph on a graph small enough to run both."""
import numpy as np
import pytest
import torch

from helpers import random_hin
from oracle.grad64 import f64_subgraph_loss_and_grads


def _state(kind, n, P, emb, hid, rep, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.25).numpy()
    sd = {'x': r(n, emb) * 0.6, 'att': r(1, P, rep), 'fc1.weight': r(rep, 2 * rep), 'fc1.bias': r(rep) * 0.4,
          'fc2.weight': r(1, rep), 'fc2.bias': r(1) * 0.4}
    for p in range(P):
        for s, (i, o) in enumerate([(emb, hid), (hid, rep)]):
            pre = 'pea_channels.%d.gnn_layers.%d.' % (p, s)
            if kind == 'gat':
                sd.update({pre + 'lin.weight': r(o, i), pre + 'att_i': r(1, 1, o), pre + 'att_j': r(1, 1, o), pre + 'bias': r(o) * 0.4})
            else:
                sd.update({pre + 'lin_rel.weight': r(o, i), pre + 'lin_rel.bias': r(o) * 0.4, pre + 'lin_root.weight': r(o, i)})
    return sd


@pytest.mark.parametrize('kind', ['gat', 'sage'])
def test_subgraph_gradients_equal_whole_graph_autograd(kind):
    from test_gpu_backward import f64_loss_and_grads
    n, blocks, rel = random_hin(7, n_user=300, n_item=120, n_attr=12, e_u2i=3000, e_attr=300)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)]]
    sd = _state(kind, n, 3, 16, 16, 8, 3)
    rng = np.random.default_rng(1)
    batch = np.stack([rng.integers(*blocks['u'], size=24), rng.integers(*blocks['i'], size=24),
                      rng.integers(*blocks['i'], size=24)], axis=1).astype(np.int64)
    want_loss, want = f64_loss_and_grads(kind, sd, edges, [2, 2, 2], 1, 'att', batch)
    loss, got, touched = f64_subgraph_loss_and_grads(kind, sd, edges, batch)
    assert abs(loss - want_loss) <= 1e-10 * abs(want_loss)
    assert set(got) == set(want) or set(want) <= set(got)
    for k, w in want.items():
        np.testing.assert_allclose(got[k], w, rtol=1e-9, atol=1e-12, err_msg=k)
    outside = np.setdiff1d(np.arange(n), touched)
    assert outside.size > 0 and not want['x'][outside].any()      # rows outside the 2-hop neighbourhood carry no gradient
