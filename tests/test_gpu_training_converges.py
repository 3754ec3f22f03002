"""End-to-end training sanity (reference loop solvers.py:211-218: zero_grad, loss, backward, step): the loss of a fixed
batch goes down under Adam for all three model kinds, and the first steps' losses agree between the single-node HIP path
(autograd.PEALossFunction) and the stack function + torch-op head it replaced."""
import numpy as np
import pytest
import torch

from helpers import build_model, random_hin, random_state_dict

pytestmark = pytest.mark.gpu


def _setup(kind, aggr):
    n, blocks, rel = random_hin(61, n_user=800, n_item=260, n_attr=18, e_u2i=9000, e_attr=700)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)]]
    model = build_model(kind, n, edges, [2, 2, 2], 32, 32, 16, channel_aggr=aggr)
    model.load_state_dict(random_state_dict(model, 13, scale=0.2))
    rng = np.random.default_rng(4)
    batch = torch.from_numpy(np.stack([rng.integers(*blocks['u'], size=256), rng.integers(*blocks['i'], size=256),
                                       rng.integers(*blocks['i'], size=256)], axis=1).astype(np.int64)).cuda()
    return model, batch


def _train(model, batch, steps):
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    model.train()
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = model.loss(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses


@pytest.mark.parametrize('kind,aggr', [('gat', 'att'), ('gcn', 'att'), ('sage', 'mean')])
def test_loss_goes_down_and_matches_the_torch_head(kind, aggr):
    from graph_recsys_benchmark_amd import engine
    model, batch = _setup(kind, aggr)
    hip = _train(model, batch, 40)
    assert np.all(np.isfinite(hip))
    assert hip[-1] < 0.6 * hip[0], hip[::8]
    model2, _ = _setup(kind, aggr)
    orig = engine.bpr_train_supported
    engine.bpr_train_supported = lambda *_: False          # PEAStackFunction + fusion / scorer / loss in torch ops
    try:
        ref = _train(model2, batch, 6)
    finally:
        engine.bpr_train_supported = orig
    # The first two steps see the same parameters up to one rounding of the gradients: tight.  After that Adam has divided
    # rounding noise by sqrt(v) (an entry whose gradient is noise still moves by ~lr per step), and at this learning rate
    # the GCN loss swings 282 -> 181 -> 265: the SAME path re-run with only the fp32 summation order of the weight-gradient
    # reduction changed (PEA_GW_PARTS=64 vs 128, profiles/tools/adam_sensitivity.py, profiles/r03/adam_sensitivity_r03.txt)
    # is 9e-6 apart at step 3, 1.9e-4 at step 5 and 2.6e-4 at step 6.  The later steps are therefore held to 2e-3.
    np.testing.assert_allclose(hip[:2], ref[:2], rtol=2e-5)
    np.testing.assert_allclose(hip[2:6], ref[2:6], rtol=2e-3)
