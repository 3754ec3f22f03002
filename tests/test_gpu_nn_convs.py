"""GPU: the drop-in KGATConv / KGCNConv / NGCFConv against outputs of the REFERENCE's own classes
(graph_recsys_benchmark/nn/*.py run by oracle/make_golden.py::make_nn_convs), plus the aggregate's backward."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(GOLDEN, 'nn_convs.npz'))


def _make(name):
    from graph_recsys_benchmark_amd import nn
    cls = {'kgat': nn.KGATConv, 'kgcn': nn.KGCNConv, 'ngcf': nn.NGCFConv}[name]
    m = cls(32, 16)
    sd = {k[len(name + '/param/'):]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(name + '/param/')}
    m.load_state_dict(sd, strict=True)
    return m.cuda()


@pytest.mark.parametrize('name', ['kgat', 'kgcn', 'ngcf'])
def test_baseline_convs_match_reference_classes(name):
    m = _make(name)
    x = torch.from_numpy(Z['x']).cuda()
    ei = torch.from_numpy(Z['edge_index']).cuda()
    att = torch.from_numpy(Z['att_map']).cuda()
    with torch.no_grad():
        out = m(x, ei) if name == 'ngcf' else m(x, ei, att)
    np.testing.assert_allclose(out.cpu().numpy(), Z[name + '/out'], rtol=2e-5, atol=2e-6)


def test_ngcf_strips_self_loops_like_the_reference():
    """nn/ngcf_conv.py:33-34: remove_self_loops before the degree count and the propagation."""
    from graph_recsys_benchmark_amd import nn
    m = nn.NGCFConv(32, 16)
    m.load_state_dict({k[len('ngcf_selfloop/param/'):]: torch.from_numpy(Z[k]) for k in Z.files
                       if k.startswith('ngcf_selfloop/param/')}, strict=True)
    m = m.cuda()
    x = torch.from_numpy(Z['x']).cuda()
    ei = torch.from_numpy(Z['ngcf_selfloop/edge_index']).cuda()
    assert bool((ei[0] == ei[1]).any())
    with torch.no_grad():
        out = m(x, ei)
        again = m(x, ei)                      # second call: cached filtered tensor, cached plan
    np.testing.assert_allclose(out.cpu().numpy(), Z['ngcf_selfloop/out'], rtol=2e-5, atol=2e-6)
    assert torch.equal(out, again)


def test_weighted_aggregate_forward_backward():
    from graph_recsys_benchmark_amd.nn import weighted_aggregate
    rng = np.random.default_rng(5)
    n, e, f = 3000, 60000, 64
    src = rng.integers(0, n, e)
    dst = np.where(rng.random(e) < 0.3, 7, rng.integers(0, n, e))          # node 7 is a hub (> 512 in-edges)
    keep = src != dst
    ei = np.stack([src[keep], dst[keep]]).astype(np.int64)
    w = rng.random(ei.shape[1]).astype(np.float32)
    x = rng.normal(size=(n, f)).astype(np.float32)
    gout = rng.normal(size=(n, f)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    wt = torch.from_numpy(w).cuda().requires_grad_(True)
    eit = torch.from_numpy(ei).cuda()
    out = weighted_aggregate(xt, eit, wt)
    out.backward(torch.from_numpy(gout).cuda())
    # float64 reference with index ops
    xd = torch.from_numpy(x).double().requires_grad_(True)
    wd = torch.from_numpy(w).double().requires_grad_(True)
    ref = torch.zeros(n, f, dtype=torch.float64).index_add_(0, torch.from_numpy(ei[1]), xd[torch.from_numpy(ei[0])] * wd[:, None])
    ref.backward(torch.from_numpy(gout).double())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), xd.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(wt.grad.cpu().numpy(), wd.grad.numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        weighted_aggregate(xt.detach(), torch.tensor([[1, 2], [1, 3]]).cuda(), torch.ones(2).cuda())   # self loop
