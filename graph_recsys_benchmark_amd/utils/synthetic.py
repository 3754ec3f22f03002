"""Synthetic heterogeneous information networks with the SHAPE of the reference's datasets.

The reference's processed datasets are absent (and cannot be rebuilt offline), so benchmarks and size-dependent
tests run on seeded synthetic HINs laid out like the reference's: node ids in contiguous per-type blocks in the
order of graph_recsys_benchmark/datasets/movielens.py:184-227 (yelp.py:250-254), one float64 [2, E] numpy array
per relation in `edge_index_nps` (datasets/movielens.py:294,318), `type_accs` = first id of each type.
user->item pairs are unique per user, item popularity is Zipf(1.0), user degree is log-normal clipped to
[11, 299] (the reference's core filter, datasets/movielens.py:693-694); attribute relations draw uniformly and
keep duplicate pairs (tag relations are built from every tagging row, datasets/movielens.py:282-288).
"""
import numpy as np

PRESETS = {
    'ml_small': dict(
        dataset='Movielens', name='latest-small', emb_dim=64, hidden_size=64, repr_dim=16, batch_size=1024,
        types=[('uid', 608), ('iid', 2121), ('genre', 19), ('year', 8), ('director', 30), ('actor', 70),
               ('writer', 35), ('tid', 42)],
        relations={'user2item': ('uid', 'iid', 79000), 'year2item': ('year', 'iid', 2121),
                   'genre2item': ('genre', 'iid', 5400), 'director2item': ('director', 'iid', 600),
                   'actor2item': ('actor', 'iid', 1500), 'writer2item': ('writer', 'iid', 700),
                   'tag2item': ('tid', 'iid', 1900), 'tag2user': ('tid', 'uid', 1900)},
        num_metapaths=9),
    'ml25m_shaped': dict(
        dataset='Movielens', name='25m', emb_dim=64, hidden_size=64, repr_dim=16, batch_size=4096,
        types=[('uid', 162541), ('iid', 59047), ('genre', 20), ('year', 8), ('director', 6000), ('actor', 20000),
               ('writer', 10000), ('tid', 15000), ('genome_tid', 1128)],
        relations={'user2item': ('uid', 'iid', 24800000), 'year2item': ('year', 'iid', 59047),
                   'genre2item': ('genre', 'iid', 110000), 'director2item': ('director', 'iid', 45000),
                   'actor2item': ('actor', 'iid', 160000), 'writer2item': ('writer', 'iid', 75000),
                   'genome_tag2item': ('genome_tid', 'iid', 600000), 'tag2user': ('tid', 'uid', 1000000),
                   'tag2item': ('tid', 'iid', 1000000)},
        num_metapaths=9),       # BASELINE.json's "9 metapaths" = the first nine of utils/general_utils.py:335-343
    'yelp_shaped': dict(
        dataset='Yelp', name='yelp', emb_dim=128, hidden_size=128, repr_dim=16, batch_size=1024,
        types=[('uid', 30000), ('iid', 10000), ('user_reviewcount', 20), ('user_friendcount', 20), ('user_fan', 20),
               ('user_star', 10), ('item_star', 9), ('item_reviewcount', 20), ('item_attribute', 80),
               ('item_categorie', 500), ('item_checkincount', 20)],
        relations={'user2item': ('uid', 'iid', 400000), 'stars2item': ('item_star', 'iid', 10000),
                   'reviewcount2item': ('item_reviewcount', 'iid', 10000),
                   'attributes2item': ('item_attribute', 'iid', 150000),
                   'categories2item': ('item_categorie', 'iid', 40000),
                   'checkincount2item': ('item_checkincount', 'iid', 10000),
                   'reviewcount2user': ('user_reviewcount', 'uid', 30000),
                   'friendcount2user': ('user_friendcount', 'uid', 30000),
                   'fans2user': ('user_fan', 'uid', 30000), 'stars2user': ('user_star', 'uid', 30000)},
        num_metapaths=11),
    'stress_10m': dict(
        dataset='Synthetic', name='stress_10m', emb_dim=128, hidden_size=128, repr_dim=16, batch_size=4096,
        types=[('uid', 4000000), ('iid', 2000000)] + [('attr_%d' % k, 500000) for k in range(8)],
        relations=dict([('user2item', ('uid', 'iid', 160000000))] +
                       [('attr_%d2item' % k, ('attr_%d' % k, 'iid', 4000000)) for k in range(8)] +
                       [('attr_02user', ('attr_0', 'uid', 8000000))]),
        num_metapaths=16),
}


class SyntheticHIN:
    """Quacks like the reference's dataset object for the path: .num_nodes, .type_accs, .edge_index_nps,
    .num_uids / .num_iids, plus .scale (edge-count multiplier used by reduced-size tests)."""

    def __init__(self, preset, seed=2019, scale=1.0):
        if preset not in PRESETS:
            raise KeyError('unknown preset %r (have %s)' % (preset, sorted(PRESETS)))
        self.preset, self.spec, self.seed, self.scale = preset, PRESETS[preset], seed, scale
        rng = np.random.default_rng(seed)
        self.type_accs, acc = {}, 0
        self.type_sizes = {}
        for t, n in self.spec['types']:
            n = max(4, int(round(n * scale))) if scale != 1.0 and n > 64 else n
            self.type_accs[t] = acc
            self.type_sizes[t] = n
            acc += n
        self.num_nodes = acc
        self.num_uids, self.num_iids = self.type_sizes['uid'], self.type_sizes['iid']
        self.edge_index_nps = {}
        for name, (src_t, dst_t, e) in self.spec['relations'].items():
            e = max(8, int(round(e * scale))) if scale != 1.0 else e
            if name == 'user2item':
                ei = self._user_item(rng, e)
            else:
                ei = self._attribute(rng, src_t, dst_t, e)
            self.edge_index_nps[name] = ei

    def _user_item(self, rng, e):
        nu, ni = self.num_uids, self.num_iids
        lo, hi = 11, min(299, ni)
        mean = e / nu
        sigma = 0.8
        deg = rng.lognormal(np.log(max(mean, 1.0)) - 0.5 * sigma * sigma, sigma, size=nu)
        for _ in range(20):                                   # rescale into [lo, hi] with the requested total
            deg = np.clip(deg * (e / deg.sum()), lo, hi)
        deg = np.floor(deg).astype(np.int64)
        rest = int(e - deg.sum())
        while rest != 0:                                      # spread the rounding residue
            room = np.flatnonzero(deg < hi) if rest > 0 else np.flatnonzero(deg > lo)
            if room.size == 0:
                break
            pick = rng.choice(room, size=min(abs(rest), room.size), replace=False)
            deg[pick] += 1 if rest > 0 else -1
            rest = int(e - deg.sum())
        users = np.repeat(np.arange(nu, dtype=np.int64), deg)
        w = 1.0 / np.arange(1, ni + 1)
        cdf = np.cumsum(w / w.sum())
        perm = rng.permutation(ni)                            # popularity rank -> item id
        items = perm[np.minimum(np.searchsorted(cdf, rng.random(users.size)), ni - 1)]
        for it in range(12):                                  # make (user, item) pairs unique per user
            key = users * ni + items
            order = np.argsort(key, kind='stable')
            dup = np.zeros(users.size, bool)
            dup[order[1:]] = key[order[1:]] == key[order[:-1]]
            n_dup = int(dup.sum())
            if n_dup == 0:
                break
            if it < 6:
                items[dup] = perm[np.minimum(np.searchsorted(cdf, rng.random(n_dup)), ni - 1)]
            else:
                items[dup] = rng.integers(0, ni, size=n_dup)
        else:
            keep = ~dup
            users, items = users[keep], items[keep]
        shuffle = rng.permutation(users.size)                 # interaction order is not sorted in the reference
        ei = np.empty((2, users.size), dtype=np.float64)
        ei[0] = users[shuffle] + self.type_accs['uid']
        ei[1] = items[shuffle] + self.type_accs['iid']
        return ei

    def _attribute(self, rng, src_t, dst_t, e):
        ns, nd = self.type_sizes[src_t], self.type_sizes[dst_t]
        if e == nd:                                           # exactly one attribute per target (year -> item)
            dst = np.arange(nd, dtype=np.int64)
        else:
            dst = rng.integers(0, nd, size=e)
        src = rng.integers(0, ns, size=e)
        ei = np.empty((2, e), dtype=np.float64)
        ei[0] = src + self.type_accs[src_t]
        ei[1] = dst + self.type_accs[dst_t]
        return ei

    def __getitem__(self, key):
        """dataset['num_nodes'] -- the reference's Dataset returns attributes for str keys (datasets/movielens.py:1141-1142)
        and the models read kwargs['dataset']['num_nodes'] (models/base.py:156)."""
        if isinstance(key, str):
            return getattr(self, key, None)
        raise TypeError('SyntheticHIN is indexed by attribute name only')

    # ---- what the reference's experiment scripts would pass around (experiments/peagat_solver_bpr.py:70-104)
    def dataset_args(self):
        return {'dataset': self.spec['dataset'], 'name': self.spec['name']}

    def model_args(self, kind='gat', num_heads=1, channel_aggr='att', entity_aware=False):
        p = self.spec['num_metapaths']
        return dict(model_type='Graph', if_use_features=False, emb_dim=self.spec['emb_dim'],
                    hidden_size=self.spec['hidden_size'], repr_dim=self.spec['repr_dim'], dropout=0,
                    num_heads=num_heads, meta_path_steps=[2] * p, channel_aggr=channel_aggr,
                    entity_aware=entity_aware, entity_aware_coff=0.1, num_nodes=self.num_nodes,
                    dataset={'num_nodes': self.num_nodes})

    def bpr_batch(self, batch_size=None, seed=2020):
        """B rows (u, i+, i-): random training edges + negatives drawn like the reference's 'random' strategy
        (np.random.randint over the item block, datasets/movielens.py:922-927; may hit seen items)."""
        b = batch_size or self.spec['batch_size']
        rs = np.random.RandomState(seed)
        u2i = self.edge_index_nps['user2item']
        pick = rs.randint(0, u2i.shape[1], size=b)
        neg = rs.randint(low=self.type_accs['iid'], high=self.type_accs['iid'] + self.num_iids, size=(b, 1))
        return np.hstack([u2i[:, pick].T.astype(np.int64), neg.astype(np.int64)])

    def eval_split(self, num_users=None, seed=2021):
        """Leave-one-out evaluation maps in the reference's layout (datasets/movielens.py:304-308, read by
        solvers.py:21-31,50-54): test_pos_unid_inid_map[u] = [one item u has not rated], neg_unid_inid_map[u] = every
        other unrated item.  Per-user Python lists, so only for the first `num_users` users (all by default; the
        25m-shaped preset would need 162 k x 59 k entries: use eval_candidates there).  Sets and returns both maps."""
        rng = np.random.default_rng(seed)
        u2i = self.edge_index_nps['user2item'].astype(np.int64)
        u0, i0 = self.type_accs['uid'], self.type_accs['iid']
        nu = self.num_uids if num_users is None else min(int(num_users), self.num_uids)
        order = np.argsort(u2i[0], kind='stable')
        starts = np.searchsorted(u2i[0][order], np.arange(u0, u0 + nu + 1))
        self.test_pos_unid_inid_map, self.neg_unid_inid_map = {}, {}
        items = np.arange(i0, i0 + self.num_iids)
        for k in range(nu):
            seen = u2i[1][order[starts[k]:starts[k + 1]]]
            unseen = np.setdiff1d(items, seen, assume_unique=False)
            j = int(rng.integers(0, unseen.size))
            self.test_pos_unid_inid_map[u0 + k] = [int(unseen[j])]
            self.neg_unid_inid_map[u0 + k] = [int(v) for v in np.delete(unseen, j)]
        return self.test_pos_unid_inid_map, self.neg_unid_inid_map

    def eval_candidates(self, num_users=None, num_neg=99, seed=2021):
        """(u_nids [U], cand [U, 1 + num_neg]) int64 for the batched evaluator: column 0 the held-out positive, the rest
        negatives drawn WITH replacement (solvers.py:29), all uniform over the item block -- the vectorised stand-in for
        eval_split + generate_candidates at sizes where per-user Python lists are impractical (benchmarks)."""
        rs = np.random.RandomState(seed)
        nu = self.num_uids if num_users is None else min(int(num_users), self.num_uids)
        lo = self.type_accs['iid']
        cand = rs.randint(lo, lo + self.num_iids, size=(nu, 1 + num_neg)).astype(np.int64)
        return np.arange(self.type_accs['uid'], self.type_accs['uid'] + nu, dtype=np.int64), cand
