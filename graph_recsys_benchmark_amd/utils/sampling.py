"""Host-side samplers with the reference's exact RNG call sequence, so sampled ids are BIT-EXACT given the same
seeds (`random.seed / np.random.seed / torch.manual_seed`, reference solvers.py:123-127) and the same call order.

    cf_negative_sampling   graph_recsys_benchmark/datasets/movielens.py:879-997 (BPR branch :920-940, shuffle :994-997)
                           (Yelp twin: datasets/yelp.py:746-760)
    generate_candidates    graph_recsys_benchmark/solvers.py:21-31
    entity_aware_row       graph_recsys_benchmark/datasets/movielens.py:1147-1181 (the six entity columns of a row)

They stay on the host on purpose: numpy's legacy RandomState stream and Python's `random` are version-frozen, a
device-side Philox stream would not reproduce the reference's ids.

    device_negative_sampling   the ADDITIONAL device-side sampler: same strategies and output
                               layout, its own Philox4x32-10 stream (csrc/sampler.hip); never a replacement for the
                               bit-exact path above.
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
    if getattr(dataset, 'entity_aware', False) and not hasattr(dataset, 'iid_feat_nids'):
        # the reference builds the per-item / per-user entity lists here from its processed CSV files
        # (datasets/movielens.py:941-991): dataset ETL, outside the accelerated path.  Without them the six
        # entity columns __getitem__ appends (see entity_aware_row) cannot be drawn, and a model with
        # entity_aware=True would index columns 3..8 of a 3-column batch.
        raise NotImplementedError('entity_aware=True needs dataset.iid_feat_nids / uid_feat_nids / nid2e_dict '
                                  '(built by the reference\'s dataset preprocessing); attach them to the dataset')
    train_data_t = torch.from_numpy(train_data_np).long()
    shuffle_idx = torch.randperm(train_data_t.shape[0])
    dataset.train_data = train_data_t[shuffle_idx]
    dataset.train_data_length = train_data_t.shape[0]
    return dataset.train_data


def entity_aware_row(dataset, train_data_t):
    """One training row as the reference's Dataset.__getitem__ returns it (datasets/movielens.py:1147-1181): with
    dataset.entity_aware the (u, i+, i-) row gets six more columns -- positive / negative entity of the item, mask,
    positive / negative entity of the user, mask -- drawn with Python's `random` in the reference's call order (bit-exact
    when called from the seeded main process; the reference's DataLoader workers make its own stream irreproducible,
    SURVEY.md appendix C.10).  `train_data_t`: int64 [3] tensor = dataset.train_data[idx]."""
    if not getattr(dataset, 'entity_aware', False):
        return train_data_t
    out = []
    for col, feats_of, acc in ((1, dataset.iid_feat_nids, dataset.type_accs['iid']),
                               (0, dataset.uid_feat_nids, dataset.type_accs['uid'])):
        nid = int(train_data_t[col])
        feat_nids = feats_of[int(nid - acc)]
        if len(feat_nids) == 0:
            out += [0, 0, 0]
        else:
            pos_entity = rd.choice(feat_nids)
            entity_type = dataset.nid2e_dict[pos_entity][0]
            lower = dataset.type_accs.get(entity_type)
            upper = lower + getattr(dataset, 'num_' + entity_type + 's')
            out += [pos_entity, rd.choice(range(lower, upper)), 1]
    return torch.cat([train_data_t, torch.tensor(out, dtype=torch.long)], dim=-1)


def generate_candidates(dataset, u_nid, num_neg_candidates=99):
    """(positives, negatives) for one test user: the held-out positives and `num_neg_candidates` negatives drawn
    WITH replacement by np.random.choice (reference solvers.py:28-29)."""
    pos_i_nids = dataset.test_pos_unid_inid_map[u_nid]
    neg_i_nids = list(np.random.choice(dataset.neg_unid_inid_map[u_nid], size=(num_neg_candidates,)))
    return pos_i_nids, neg_i_nids


def device_negative_sampling(dataset, seed, epoch=0, device='cuda', shuffle=True):
    """BPR triples sampled ON THE GPU: int64 [M * num_negative_samples, 3] CUDA tensor (u, i+, i-), optionally shuffled
    with torch.randperm on the device.  'random': uniform items; 'unseen': uniform over the items the user has no
    training interaction with (the pool the reference draws from, datasets/movielens.py:929-937).  `epoch` selects an
    independent stream per call (the Philox counter's fourth word)."""
    from .. import _lib
    lib = _lib.require_device()
    if dataset.cf_loss_type != 'BPR':
        raise NotImplementedError('only the BPR branch is on the accelerated path')
    pos = torch.as_tensor(dataset.edge_index_nps['user2item']).to(device=device, dtype=torch.int64)
    pos_u, pos_i = pos[0].contiguous(), pos[1].contiguous()
    k = int(dataset.num_negative_samples)
    lo, n_items = int(dataset.type_accs['iid']), int(dataset.num_iids)
    keys = None
    if dataset.sampling_strategy == 'unseen':
        keys = torch.unique(pos_u * n_items + (pos_i - lo))        # ascending
    elif dataset.sampling_strategy != 'random':
        raise NotImplementedError
    out = torch.empty((pos_u.numel() * k, 3), dtype=torch.int64, device=device)
    exhausted = torch.zeros(1, dtype=torch.int32, device=device)
    _lib.check(lib.pea_sample_negatives(pos_u.numel(), k, _lib.ptr(pos_u), _lib.ptr(pos_i), lo, n_items, _lib.ptr(keys),
                                        0 if keys is None else keys.numel(), int(seed) & (2 ** 64 - 1),
                                        int(epoch) & 0xFFFFFFFF, _lib.ptr(out), out.stride(0), _lib.ptr(exhausted),
                                        _lib.current_stream()))
    n_exhausted = int(exhausted.item())
    if n_exhausted:
        # 'unseen': 64 rejection attempts all hit seen items (a user who has rated nearly every item); the row keeps
        # a seen item as its negative.  Loud, because such a row is a false negative.
        import warnings
        warnings.warn('device_negative_sampling: %d of %d negatives are seen items (rejection sampling exhausted)'
                      % (n_exhausted, out.shape[0]), RuntimeWarning)
    dataset.negatives_exhausted = n_exhausted
    if shuffle:
        out = out[torch.randperm(out.shape[0], device=device)]
    dataset.train_data, dataset.train_data_length = out, out.shape[0]
    return out
