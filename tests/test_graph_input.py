"""A8: the metapath tables / edge-list wire format of graph_recsys_benchmark_amd.utils.graph_input against a fixture
recorded from the reference's own update_pea_graph_input (utils/general_utils.py:280-395) on marker relations
(oracle/make_golden.py::make_graph_input_tables)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN

with open(os.path.join(GOLDEN, 'graph_input_tables.json')) as f:
    TABLES = json.load(f)

CASES = [('Movielens/latest-small', {'dataset': 'Movielens', 'name': 'latest-small'}),
         ('Movielens/25m', {'dataset': 'Movielens', 'name': '25m'}),
         ('Yelp', {'dataset': 'Yelp', 'name': ''})]


@pytest.mark.parametrize('tag,dargs', CASES)
def test_metapath_tables_match_the_reference(tag, dargs):
    from graph_recsys_benchmark_amd.utils import metapath_table
    got = [[[rel, int(flip)] for rel, flip in steps] for steps in metapath_table(dargs)]
    assert got == TABLES[tag]


@pytest.mark.parametrize('tag,dargs', CASES)
def test_edge_lists_have_the_reference_wire_format(tag, dargs):
    """int64 [2, E] tensors, reversed relations = flipped rows; an unflipped relation is the SAME tensor object wherever it
    is reused, every flip is a fresh equal copy (reference utils/general_utils.py:300-307)."""
    from graph_recsys_benchmark_amd.utils import update_pea_graph_input
    names = sorted({rel for steps in TABLES[tag] for rel, _ in steps})
    rng = np.random.default_rng(0)
    nps = {n: rng.integers(0, 50, size=(2, 7)).astype(np.float64) for n in names}     # float64 like the reference's arrays
    lists = update_pea_graph_input(dargs, {'device': 'cpu'}, types.SimpleNamespace(edge_index_nps=nps))
    assert len(lists) == len(TABLES[tag])
    seen = {}
    for steps, want in zip(lists, TABLES[tag]):
        assert len(steps) == len(want)
        for t, (rel, flip) in zip(steps, want):
            base = torch.from_numpy(nps[rel]).long()
            assert t.dtype == torch.int64 and torch.equal(t, torch.flip(base, dims=[0]) if flip else base)
            if not flip:
                assert seen.setdefault(rel, t) is t
    n = 5 if tag != 'Yelp' else 4
    assert len(update_pea_graph_input(dargs, {'device': 'cpu', 'num_metapaths': n}, types.SimpleNamespace(edge_index_nps=nps))) == n
