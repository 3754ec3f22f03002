"""CPU: the C oracle (oracle/pea_oracle.c) against the golden vectors produced by the
reference's own models/base.py (oracle/make_golden.py).  Tolerance: fp32, rtol 1e-5 /
atol 1e-6 (two fp32 implementations with different summation order in the GEMMs)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, GoldenCase, golden_cases
from oracle import oracle as orc

RTOL, ATOL = 1e-5, 1e-6


@pytest.mark.parametrize('name', golden_cases())
def test_oracle_matches_reference_fixture(name):
    g = GoldenCase(name)
    sd = g.state_dict
    att = sd.get('att')
    fused, stack = orc.pea_forward(g.kind, sd['x'], g.edges, g.channel_params(), g.heads_lists(), att=att,
                                   channel_aggr=g.meta['channel_aggr'], return_stack=True)
    for p in range(g.P):
        np.testing.assert_allclose(stack[:, p], g.out['channel/%d' % p], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(fused, g.out['repr'], rtol=RTOL, atol=ATOL)
    masked = orc.fuse(stack, att if g.meta['channel_aggr'] == 'att' else None, masked=1)
    np.testing.assert_allclose(masked, g.out['repr_mask1'], rtol=RTOL, atol=ATOL)
    loss, pos, neg = orc.pea_loss(g.out['repr'], g.batch, sd['fc1.weight'], sd['fc1.bias'],
                                  sd['fc2.weight'], sd['fc2.bias'])
    np.testing.assert_allclose(pos, g.out['pos'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(neg, g.out['neg'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(loss, g.out['loss_eval'], rtol=RTOL)
    if g.batch9 is None:
        np.testing.assert_allclose(loss, g.out['loss_train'], rtol=RTOL)
    else:
        reg = orc.entity_reg(sd['x'], g.batch9)
        np.testing.assert_allclose(loss + 0.1 * reg, g.out['loss_train'], rtol=RTOL)


def test_oracle_rejects_out_of_range_ids():
    x = np.zeros((4, 8), np.float32)
    ei = np.array([[0, 5], [1, 2]], np.int64)
    w = np.zeros((8, 8), np.float32)
    with pytest.raises(RuntimeError):
        orc.gcn_conv(x, ei, w, None)


def test_oracle_empty_relation():
    """E = 0: GAT/GCN reduce to the self loop, SAGE to bias + root term."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 8)).astype(np.float32)
    ei = np.zeros((2, 0), np.int64)
    w = rng.normal(size=(4, 8)).astype(np.float32)
    ai = rng.normal(size=(1, 1, 4)).astype(np.float32)
    aj = rng.normal(size=(1, 1, 4)).astype(np.float32)
    b = rng.normal(size=(4,)).astype(np.float32)
    out = orc.gat_conv(x, ei, w, ai, aj, b)
    np.testing.assert_allclose(out, x @ w.T + b, rtol=1e-5, atol=1e-6)
    out = orc.gcn_conv(x, ei, w.T.copy(), b)
    np.testing.assert_allclose(out, x @ w.T + b, rtol=1e-5, atol=1e-6)
    out = orc.sage_conv(x, ei, w, b, w)
    np.testing.assert_allclose(out, b + x @ w.T, rtol=1e-5, atol=1e-6)


def test_gcn_degree_side_switch_differs():
    g = GoldenCase('pea_gcn_p5s2_att')
    lp = g.layer_params(2, 0)
    a = orc.gcn_conv(g.state_dict['x'], g.edges[2][0], lp['weight'], lp['bias'], 'row')
    b = orc.gcn_conv(g.state_dict['x'], g.edges[2][0], lp['weight'], lp['bias'], 'col')
    assert np.abs(a - b).max() > 1e-3


def test_rng_streams_known_answers():
    """Legacy numpy / Python RNG streams the reference's samplers draw from
    (datasets/movielens.py:920-937, solvers.py:29) -- version-frozen, must be bit-exact."""
    import random
    with open(os.path.join(GOLDEN, 'rng_streams.json')) as f:
        ks = json.load(f)
    np.random.seed(2020)
    assert np.random.randint(low=608, high=608 + 2121, size=(32, 1)).reshape(-1).tolist() == ks['randint_608_2729_x32']
    np.random.seed(2020)
    assert np.random.choice(list(range(100, 200)), size=(5,)).tolist() == ks['choice_100_200_x5']
    random.seed(2020)
    assert random.choices(list(range(100, 200)), k=4) == ks['choices_100_200_k4']
