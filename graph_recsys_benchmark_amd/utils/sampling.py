"""Host-side samplers with the reference's exact RNG call sequence, so sampled ids are BIT-EXACT given the same
seeds (`random.seed / np.random.seed / torch.manual_seed`, reference solvers.py:123-127) and the same call order.

    cf_negative_sampling   graph_recsys_benchmark/datasets/movielens.py:879-997 (BPR branch :920-940, shuffle :994-997)
                           (Yelp twin: datasets/yelp.py:746-760)
    generate_candidates    graph_recsys_benchmark/solvers.py:21-31

They stay on the host on purpose: numpy's legacy RandomState stream and Python's `random` are version-frozen, a
device-side Philox stream would not reproduce the reference's ids.
"""
import random as rd

import numpy as np
import torch


def cf_negative_sampling(dataset):
    """BPR training triples: every positive (u, i+) repeated `num_negative_samples` times + one sampled negative
    column, then a torch.randperm shuffle.  Sets dataset.train_data / train_data_length like the reference and
    returns the int64 [M*neg, 3] tensor."""
    if dataset.cf_loss_type != 'BPR':
        raise NotImplementedError('only the BPR branch is on the accelerated path')
    pos = dataset.edge_index_nps['user2item'].T
    num_interactions = pos.shape[0]
    k = dataset.num_negative_samples
    train_data_np = np.repeat(pos, repeats=k, axis=0)
    if dataset.sampling_strategy == 'random':
        lo = dataset.type_accs['iid']
        neg = np.random.randint(low=lo, high=lo + dataset.num_iids, size=(num_interactions * k, 1))
    elif dataset.sampling_strategy == 'unseen':
        parts = []
        for u_nid in pos[:, 0]:
            pool = dataset.test_pos_unid_inid_map[u_nid] + dataset.neg_unid_inid_map[u_nid]
            parts.append(np.array(rd.choices(pool, k=k), dtype=np.int64).reshape(-1, 1))
        neg = np.vstack(parts)
    else:
        raise NotImplementedError
    train_data_np = np.hstack([train_data_np, neg])
    train_data_t = torch.from_numpy(train_data_np).long()
    shuffle_idx = torch.randperm(train_data_t.shape[0])
    dataset.train_data = train_data_t[shuffle_idx]
    dataset.train_data_length = train_data_t.shape[0]
    return dataset.train_data


def generate_candidates(dataset, u_nid, num_neg_candidates=99):
    """(positives, negatives) for one test user: the held-out positives and `num_neg_candidates` negatives drawn
    WITH replacement by np.random.choice (reference solvers.py:28-29)."""
    pos_i_nids = dataset.test_pos_unid_inid_map[u_nid]
    neg_i_nids = list(np.random.choice(dataset.neg_unid_inid_map[u_nid], size=(num_neg_candidates,)))
    return pos_i_nids, neg_i_nids
