"""Sampling (bit-exact) and evaluation mirrors against fixtures produced by the REFERENCE's own code
(datasets/movielens.py cf_negative_sampling, solvers.py BaseSolver.metrics, utils/rec_utils.py), see
oracle/make_golden.py::make_sampling_and_metrics."""
import json
import os
import random

import numpy as np
import pytest
import torch

from helpers import GOLDEN, build_model

Z = np.load(os.path.join(GOLDEN, 'sampling_metrics.npz'))


class FakeDataset:
    """Same construction as oracle/make_golden.py::_FakeDataset (inputs of the fixture)."""

    def __init__(self, seed, n_user=40, n_item=70, e=600):
        rng = np.random.default_rng(seed)
        self.type_accs = {'uid': 0, 'iid': n_user}
        self.num_uids, self.num_iids = n_user, n_item
        u = rng.integers(0, n_user, size=e)
        i = n_user + rng.integers(0, n_item, size=e)
        pairs = np.unique(np.stack([u, i]), axis=1)
        self.edge_index_nps = {'user2item': pairs.astype(np.float64)}
        seen = {int(x): set() for x in range(n_user)}
        for a, b in pairs.T:
            seen[int(a)].add(int(b))
        self.test_pos_unid_inid_map, self.neg_unid_inid_map = {}, {}
        for x in range(n_user):
            unseen = [j for j in range(n_user, n_user + n_item) if j not in seen[x]]
            k = int(rng.integers(0, len(unseen)))
            self.test_pos_unid_inid_map[x] = [unseen[k]]
            self.neg_unid_inid_map[x] = unseen[:k] + unseen[k + 1:]
        self.cf_loss_type, self.entity_aware = 'BPR', False
        self.num_negative_samples = 4


@pytest.mark.parametrize('strategy', ['random', 'unseen'])
def test_negative_sampling_is_bit_exact(strategy):
    from graph_recsys_benchmark_amd.utils import cf_negative_sampling
    ds = FakeDataset(11)
    ds.sampling_strategy = strategy
    random.seed(2020)
    np.random.seed(2020)
    torch.manual_seed(2020)
    got = cf_negative_sampling(ds).numpy()
    np.testing.assert_array_equal(got, Z['train_data_' + strategy])
    assert got.dtype == np.int64 and ds.train_data_length == got.shape[0]


def _entity_dataset(seed=13):
    """Same construction as oracle/make_golden.py::entity_fake_dataset."""
    fake = FakeDataset(seed)
    rng = np.random.default_rng(seed + 100)
    base = fake.num_uids + fake.num_iids
    fake.type_accs.update({'genre': base, 'tid': base + 5})
    fake.num_genres, fake.num_tids = 5, 9
    fake.nid2e_dict = {base + k: ('genre', k) for k in range(5)}
    fake.nid2e_dict.update({base + 5 + k: ('tid', k) for k in range(9)})
    fake.iid_feat_nids = [[int(base + v) for v in rng.integers(0, 14, size=int(rng.integers(0, 4)))]
                          for _ in range(fake.num_iids)]
    fake.uid_feat_nids = [[int(base + 5 + v) for v in rng.integers(0, 9, size=int(rng.integers(0, 3)))]
                          for _ in range(fake.num_uids)]
    fake.entity_aware = True
    return fake


def test_entity_aware_rows_are_bit_exact():
    """The six entity columns against rows produced by the reference's own Dataset.__getitem__
    (datasets/movielens.py:1147-1181; oracle/make_golden.py::make_entity_rows)."""
    from graph_recsys_benchmark_amd.utils import cf_negative_sampling, entity_aware_row
    z = np.load(os.path.join(GOLDEN, 'entity_rows.npz'))
    ds = _entity_dataset()
    ds.sampling_strategy = 'random'
    random.seed(2020)
    np.random.seed(2020)
    torch.manual_seed(2020)
    cf_negative_sampling(ds)
    np.testing.assert_array_equal(ds.train_data.numpy(), z['train_data'])
    random.seed(77)
    rows = torch.stack([entity_aware_row(ds, ds.train_data[i]) for i in range(200)]).numpy()
    np.testing.assert_array_equal(rows, z['rows'])
    assert rows.shape == (200, 9) and (rows[:, 5] == 0).any() and (rows[:, 8] == 1).any()


def test_entity_aware_sampling_without_entity_lists_raises():
    from graph_recsys_benchmark_amd.utils import cf_negative_sampling
    ds = FakeDataset(11)
    ds.sampling_strategy, ds.entity_aware = 'random', True
    with pytest.raises(NotImplementedError):
        cf_negative_sampling(ds)


def test_rank_metrics_match_rec_utils_definitions():
    from graph_recsys_benchmark_amd.solvers import metrics_from_ranks
    with open(os.path.join(GOLDEN, 'rec_utils.json')) as f:
        cases = json.load(f)
    for c in cases:
        r = int(np.argmax(c['hit_vec']))
        hr, nd = metrics_from_ranks([r])
        np.testing.assert_array_equal(hr[0], np.array(c['hit'], dtype=np.float64))
        np.testing.assert_allclose(nd[0], np.array(c['ndcg']), rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_batched_evaluator_matches_reference_metrics_loop():
    """Same model, same dataset maps, same np.random seed as the reference's BaseSolver.metrics run."""
    from graph_recsys_benchmark_amd.solvers import metrics
    ds = FakeDataset(12)
    n = int(Z['metrics_num_nodes'])
    sd = {k[len('metrics_param/'):]: Z[k] for k in Z.files if k.startswith('metrics_param/')}
    edges = [[Z['metrics_edge/%d/%d' % (p, s)] for s in range(2)] for p in range(3)]
    model = build_model('gat', n, edges, [2, 2, 2], 32, 24, 16, state_dict=sd)
    model.eval()
    np.random.seed(2021)
    hr, nd, auc, loss = metrics(model, ds, num_neg_candidates=99)
    np.testing.assert_allclose(hr, Z['metrics_HR'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(nd, Z['metrics_NDCG'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(auc, Z['metrics_AUC'], rtol=1e-6)
    np.testing.assert_allclose(loss, Z['metrics_loss'], rtol=1e-5)
    # the host RNG stream is left where the reference leaves it
    np.random.seed(2021)
    from graph_recsys_benchmark_amd.utils import generate_candidates
    for u in ds.test_pos_unid_inid_map:
        generate_candidates(ds, u, 99)
    a = np.random.randint(0, 1 << 30)
    np.random.seed(2021)
    metrics(model, ds, num_neg_candidates=99)
    assert np.random.randint(0, 1 << 30) == a
