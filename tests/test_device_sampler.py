"""Device-side negative sampler (an addition next to the bit-exact host mirror).
CPU: the numpy restatement of its stream (oracle/philox.py) against the published Philox4x32-10 known-answer vectors.
GPU: the kernel against that restatement bit for bit, plus the properties the reference's pool has."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import philox                                            # noqa: E402
from test_sampling_metrics import FakeDataset                        # noqa: E402


def test_philox_known_answer_vectors():
    ones, zeros = philox.known_answer()
    assert ones == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]   # Random123 kat_vectors, philox4x32 10 rounds
    assert zeros == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def _inputs(ds):
    pos = ds.edge_index_nps['user2item'].astype(np.int64)
    lo, n_items = ds.type_accs['iid'], ds.num_iids
    keys = np.unique(pos[0] * n_items + (pos[1] - lo))
    return pos, lo, n_items, keys


def test_oracle_sampler_properties():
    ds = FakeDataset(11)
    pos, lo, n_items, keys = _inputs(ds)
    out, exhausted = philox.sample_negatives(pos[0], pos[1], 4, lo, n_items, keys, seed=2020)
    assert exhausted == 0 and out.shape == (pos.shape[1] * 4, 3)
    np.testing.assert_array_equal(out[:, :2], np.repeat(pos.T, 4, axis=0))
    assert ((out[:, 2] >= lo) & (out[:, 2] < lo + n_items)).all()
    assert not np.isin(out[:, 0] * n_items + (out[:, 2] - lo), keys).any()          # never a training positive
    again, _ = philox.sample_negatives(pos[0], pos[1], 4, lo, n_items, keys, seed=2020)
    other, _ = philox.sample_negatives(pos[0], pos[1], 4, lo, n_items, keys, seed=2021)
    np.testing.assert_array_equal(out, again)
    assert (out[:, 2] != other[:, 2]).mean() > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize('strategy', ['random', 'unseen'])
def test_device_sampler_matches_the_restatement(strategy):
    from graph_recsys_benchmark_amd.utils import device_negative_sampling
    ds = FakeDataset(11)
    ds.sampling_strategy = strategy
    pos, lo, n_items, keys = _inputs(ds)
    got = device_negative_sampling(ds, seed=(7 << 32) | 2020, epoch=3, shuffle=False).cpu().numpy()
    want, _ = philox.sample_negatives(pos[0], pos[1], ds.num_negative_samples, lo, n_items,
                                      keys if strategy == 'unseen' else None, seed=(7 << 32) | 2020, offset=3)
    np.testing.assert_array_equal(got, want)
    assert got.dtype == np.int64 and ds.train_data_length == got.shape[0]
    shuffled = device_negative_sampling(ds, seed=(7 << 32) | 2020, epoch=3).cpu().numpy()   # same multiset of rows
    np.testing.assert_array_equal(np.sort(shuffled.view('i8,i8,i8'), axis=0), np.sort(got.view('i8,i8,i8'), axis=0))


@pytest.mark.gpu
def test_device_sampler_distribution_and_exhaustion():
    from graph_recsys_benchmark_amd import _lib
    lib = _lib.require_device()
    n_items, lo, n_pos, k = 50, 1000, 20000, 5
    pos_u = torch.zeros(n_pos, dtype=torch.int64, device='cuda')
    pos_i = torch.full((n_pos,), lo, dtype=torch.int64, device='cuda')
    seen = torch.arange(0, 10, dtype=torch.int64, device='cuda')                  # user 0 has seen items 0..9
    out = torch.empty((n_pos * k, 3), dtype=torch.int64, device='cuda')
    ex = torch.zeros(1, dtype=torch.int32, device='cuda')
    _lib.check(lib.pea_sample_negatives(n_pos, k, _lib.ptr(pos_u), _lib.ptr(pos_i), lo, n_items, _lib.ptr(seen), 10, 99, 0,
                                        _lib.ptr(out), 3, _lib.ptr(ex), _lib.current_stream()))
    neg = (out[:, 2] - lo).cpu().numpy()
    assert int(ex.item()) == 0 and neg.min() >= 10 and neg.max() < n_items
    counts = np.bincount(neg, minlength=n_items)[10:]
    expect = n_pos * k / 40.0
    assert np.abs(counts - expect).max() < 6 * np.sqrt(expect)                   # uniform over the 40 unseen items
    # a user who has seen everything: the attempts run out, every row is flagged, ids stay inside the item block
    seen_all = torch.arange(0, n_items, dtype=torch.int64, device='cuda')
    _lib.check(lib.pea_sample_negatives(64, 1, _lib.ptr(pos_u), _lib.ptr(pos_i), lo, n_items, _lib.ptr(seen_all), n_items, 1, 0,
                                        _lib.ptr(out), 3, _lib.ptr(ex), _lib.current_stream()))
    assert int(ex.item()) == 64 and bool(((out[:64, 2] >= lo) & (out[:64, 2] < lo + n_items)).all())
    with pytest.raises(_lib.PeaError):
        _lib.check(lib.pea_sample_negatives(4, 0, _lib.ptr(pos_u), _lib.ptr(pos_i), lo, n_items, None, 0, 1, 0, _lib.ptr(out), 3,
                                            _lib.ptr(ex), _lib.current_stream()))
