"""update_pea_graph_input: the P x S list of int64 COO [2, E] device tensors per dataset.

Mirror of graph_recsys_benchmark/utils/general_utils.py:280-395 (same signature, same metapath tables, same
torch.flip idiom so reversed relations are fresh equal-content copies, same NotImplementedError for unknown
datasets).  `train_args['num_metapaths']` (optional) keeps only the first n metapaths -- BASELINE.json's
"MovieLens-25m, 9 metapaths" is the first nine of the reference's thirteen.  A 'Synthetic' dataset branch
carries the stress preset of SURVEY.md 8(d).
"""
import torch

# (relation, flipped) per step; tables transcribed from the reference's metapath definitions
_ML_SMALL = [
    [('user2item', 0), ('user2item', 1)], [('user2item', 1), ('user2item', 0)],
    [('year2item', 0), ('user2item', 1)], [('actor2item', 0), ('user2item', 1)],
    [('writer2item', 0), ('user2item', 1)], [('director2item', 0), ('user2item', 1)],
    [('genre2item', 0), ('user2item', 1)], [('tag2item', 0), ('user2item', 1)],
    [('tag2user', 0), ('user2item', 0)],
]
_ML_25M = [
    [('user2item', 0), ('user2item', 1)], [('year2item', 0), ('user2item', 1)],
    [('actor2item', 0), ('user2item', 1)], [('writer2item', 0), ('user2item', 1)],
    [('director2item', 0), ('user2item', 1)], [('genre2item', 0), ('user2item', 1)],
    [('genome_tag2item', 0), ('user2item', 1)], [('tag2user', 1), ('tag2user', 0)],
    [('tag2item', 1), ('tag2user', 0)], [('user2item', 1), ('user2item', 0)],
    [('tag2user', 0), ('user2item', 0)], [('tag2item', 1), ('tag2item', 0)],
    [('tag2user', 1), ('tag2item', 0)],
]
_YELP = [
    [('user2item', 0), ('user2item', 1)], [('user2item', 1), ('user2item', 0)],
    [('stars2item', 0), ('user2item', 1)], [('reviewcount2item', 0), ('user2item', 1)],
    [('attributes2item', 0), ('user2item', 1)], [('categories2item', 0), ('user2item', 1)],
    [('checkincount2item', 0), ('user2item', 1)], [('reviewcount2user', 0), ('user2item', 0)],
    [('friendcount2user', 0), ('user2item', 0)], [('fans2user', 0), ('user2item', 0)],
    [('stars2user', 0), ('user2item', 0)],
]
_STRESS = ([[('user2item', 0), ('user2item', 1)], [('user2item', 1), ('user2item', 0)]] +
           [[('attr_%d2item' % k, 0), ('user2item', 1)] for k in range(8)] +
           [[('attr_02user', 0), ('user2item', 0)], [('attr_02user', 1), ('attr_02user', 0)]] +
           [[('attr_%d2item' % k, 1), ('attr_%d2item' % k, 0)] for k in range(4)])


def metapath_table(dataset_args):
    if dataset_args['dataset'] == 'Movielens':
        if dataset_args['name'] == 'latest-small':
            return _ML_SMALL
        if dataset_args['name'] == '25m':
            return _ML_25M
    elif dataset_args['dataset'] == 'Yelp':
        return _YELP
    elif dataset_args['dataset'] == 'Synthetic':
        return _STRESS
    raise NotImplementedError


def update_pea_graph_input(dataset_args, train_args, dataset):
    table = metapath_table(dataset_args)
    n = train_args.get('num_metapaths')
    if n is not None:
        table = table[:n]
    device = train_args['device']
    base = {}
    for steps in table:
        for rel, _ in steps:
            if rel not in base:
                base[rel] = torch.from_numpy(dataset.edge_index_nps[rel]).long().to(device)
    return [[torch.flip(base[rel], dims=[0]) if flipped else base[rel] for rel, flipped in steps]
            for steps in table]
