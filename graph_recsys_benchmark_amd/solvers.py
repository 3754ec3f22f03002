"""Batched evaluation with the reference's definitions.

Mirror of BaseSolver.metrics (graph_recsys_benchmark/solvers.py:33-104): the reference walks the test users one by
one (candidate draw, pandas merge, 4 H2D copies, 3 MLP launches, a sort and 3 D2H copies per user).  Here the
candidate ids are drawn on the host with the very same np.random.choice call per user (bit-exact ids, same stream
position afterwards), then ALL users are scored by one launch of pea_rank_eval; HR@5..20 / NDCG@5..20 / AUC / eval
loss follow graph_recsys_benchmark/utils/rec_utils.py:7-30 (hit :7, ndcg :18, auc :28) and solvers.py:72,93-96.
"""
import numpy as np
import torch

from . import engine
from .utils.sampling import generate_candidates

NUM_RECS_RANGE = 20   # utils/rec_utils.py:4


def metrics_from_ranks(rank):
    """rank [U]: number of negatives placed before the (single) positive.  Returns HR [U,16], NDCG [U,16] for
    k = 5..20 exactly as hit() / ndcg() compute them from a hit vector with one positive."""
    rank = np.asarray(rank).reshape(-1, 1)
    ks = np.arange(5, NUM_RECS_RANGE + 1).reshape(1, -1)
    inside = rank < ks
    hr = inside.astype(np.float64)
    ndcg = np.where(inside, 1.0 / np.log2(rank + 2.0), 0.0)
    return hr, ndcg


def metrics(model, dataset, num_neg_candidates=99):
    """Returns (HR[16], NDCG[16], AUC[1], eval_loss[1]) means over the test users, like BaseSolver.metrics.
    `model` must be in eval() mode (cached_repr refreshed)."""
    u_nids = list(dataset.test_pos_unid_inid_map.keys())
    cand = np.empty((len(u_nids), 1 + num_neg_candidates), dtype=np.int64)
    for idx, u_nid in enumerate(u_nids):
        pos_i_nids, neg_i_nids = generate_candidates(dataset, u_nid, num_neg_candidates)
        if len(pos_i_nids) == 0 or len(neg_i_nids) == 0:
            raise ValueError("No pos or neg samples found in evaluation!")
        if len(pos_i_nids) != 1:
            raise NotImplementedError('the batched evaluator expects the leave-one-out protocol (one positive per user, '
                                      'datasets/movielens.py:304-308)')
        cand[idx, 0] = pos_i_nids[0]
        cand[idx, 1:] = neg_i_nids
    dev = model.cached_repr.device
    scores, rank, auc, loss = engine.rank_eval(model.cached_repr, torch.as_tensor(np.asarray(u_nids, dtype=np.int64), device=dev),
                                               torch.from_numpy(cand).to(dev), model.fc1.weight, model.fc1.bias,
                                               model.fc2.weight, model.fc2.bias)
    hr, ndcg = metrics_from_ranks(rank.cpu().numpy())
    return hr.mean(axis=0), ndcg.mean(axis=0), np.array([auc.double().mean().item()]), np.array([loss.double().mean().item()])
