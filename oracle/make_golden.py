"""ORACLE / TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz|json.

Runs in the AUTHORING container only (needs /root/reference; the GPU box has no
reference tree and only consumes the committed fixtures).

What is executed for real from the reference (read-only, never copied):
    graph_recsys_benchmark/models/base.py      GraphRecsysModel.loss / .eval,
                                               PEABaseChannel.forward, PEABaseRecsysModel.*
    graph_recsys_benchmark/models/pea{gat,gcn,sage}.py   channel construction
    graph_recsys_benchmark/utils/rec_utils.py  hit / ndcg / auc
They are loaded file-by-file through importlib under hand-made *package shells*
(so the reference's package __init__ chain, which needs torch_geometric datasets,
pandas pickles etc., never runs).  The only third-party names those files touch
are torch_geometric.nn.inits.{glorot,zeros} and torch_geometric.nn.{GATConv,
GCNConv,SAGEConv}; torch-geometric is not installed here, so those names are
bound to oracle/pyg_restatement.py (spec restatement, parity of the conv
arithmetic unpinned -- see its header).  Everything downstream of the convs in
the fixtures (relu chain, channel stack, ablation mask, att/mean fusion,
predict MLP, BPR loss, entity-aware term, eval() cache) is the reference's own
code acting on those conv outputs.

Fixtures are numbers only: inputs (x, parameters, edge lists, batches) and
outputs (per-channel reps, fused cached_repr, predictions, losses).
"""
import importlib
import json
import os
import random
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import pyg_restatement as R  # noqa: E402


def load_reference_models():
    tg = types.ModuleType('torch_geometric')
    tgnn = types.ModuleType('torch_geometric.nn')
    inits = types.ModuleType('torch_geometric.nn.inits')
    inits.glorot, inits.zeros = R.glorot, R.zeros
    tgnn.inits = inits
    tgnn.GATConv, tgnn.GCNConv, tgnn.SAGEConv = R.GATConv, R.GCNConv, R.SAGEConv
    tg.nn = tgnn
    sys.modules.update({'torch_geometric': tg, 'torch_geometric.nn': tgnn,
                        'torch_geometric.nn.inits': inits})
    pkg = types.ModuleType('graph_recsys_benchmark')
    pkg.__path__ = [os.path.join(REF, 'graph_recsys_benchmark')]
    mp = types.ModuleType('graph_recsys_benchmark.models')
    mp.__path__ = [os.path.join(REF, 'graph_recsys_benchmark', 'models')]
    up = types.ModuleType('graph_recsys_benchmark.utils')
    up.__path__ = [os.path.join(REF, 'graph_recsys_benchmark', 'utils')]
    sys.modules.update({'graph_recsys_benchmark': pkg, 'graph_recsys_benchmark.models': mp,
                        'graph_recsys_benchmark.utils': up})
    mods = {k: importlib.import_module('graph_recsys_benchmark.models.' + k)
            for k in ('base', 'peagat', 'peagcn', 'peasage')}
    if not hasattr(np, 'int'):
        np.int = int      # utils/rec_utils.py:21 uses the alias removed in numpy 1.24
    mods['rec_utils'] = importlib.import_module('graph_recsys_benchmark.utils.rec_utils')
    return mods


def tiny_hin(seed, n_user=60, n_item=90, n_attr=12, n_tag=14, e_u2i=900):
    """A small HIN in the reference's id layout (contiguous per-type blocks) with the
    features that matter for parity: multi-edges (tag relations), a hub item, isolated
    nodes, and explicit self loops in one relation (dropped by GAT/GCN, kept by SAGE)."""
    rng = np.random.default_rng(seed)
    u0, i0, a0, t0 = 0, n_user, n_user + n_item, n_user + n_item + n_attr
    n = t0 + n_tag + 5                     # 5 isolated trailing nodes
    users = rng.integers(u0, i0, size=e_u2i)
    pop = rng.zipf(1.3, size=e_u2i) % n_item
    items = i0 + pop
    hub = np.stack([np.arange(u0, i0), np.full(n_user, i0)])          # every user -> item 0
    u2i = np.concatenate([np.stack([users, items]), hub], axis=1)
    attr2item = np.stack([a0 + rng.integers(0, n_attr, size=150), i0 + rng.integers(0, n_item, size=150)])
    tag2item = np.stack([t0 + rng.integers(0, n_tag, size=200), i0 + rng.integers(0, n_item, size=200)])
    tag2item = np.concatenate([tag2item, tag2item[:, :40]], axis=1)    # repeated pairs
    tag2user = np.stack([t0 + rng.integers(0, n_tag, size=120), u0 + rng.integers(0, n_user, size=120)])
    loops = np.stack([np.arange(i0, i0 + 7), np.arange(i0, i0 + 7)])
    tag2user = np.concatenate([tag2user, loops], axis=1)               # explicit self loops
    rel = {'user2item': u2i, 'attr2item': attr2item, 'tag2item': tag2item, 'tag2user': tag2user}
    rel = {k: torch.from_numpy(v.astype(np.float64)).long() for k, v in rel.items()}
    return n, dict(u=(u0, i0), i=(i0, a0)), rel


def metapaths(rel, which):
    f = lambda t: torch.flip(t, dims=[0])       # utils/general_utils.py:300-308 idiom
    u2i, a2i, t2i, t2u = rel['user2item'], rel['attr2item'], rel['tag2item'], rel['tag2user']
    full = [[u2i, f(u2i)], [f(u2i), u2i], [a2i, f(u2i)], [t2i, f(u2i)], [t2u, u2i]]
    if which == 'p5s2':
        return full, [2, 2, 2, 2, 2]
    if which == 'p3mixed':
        return [[u2i, f(u2i)], [f(t2i), t2i, f(u2i)], [t2u]], [2, 3, 1]
    if which == 'p3deep':
        # no 1-step channel: with num_heads > 1 the reference's 1-step GAT channel emits
        # repr_dim*heads columns (models/peagat.py:16) and torch.cat at models/base.py:196 fails
        return [[u2i, f(u2i)], [f(t2i), t2i, f(u2i)], [t2u, u2i]], [2, 3, 2]
    raise ValueError(which)


def make_case(mods, name, kind, which, heads, channel_aggr, seed, entity_aware=False):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    n, blocks, rel = tiny_hin(seed)
    mpl, steps = metapaths(rel, which)
    ModelBase = {'gat': mods['peagat'].PEAGATRecsysModel, 'gcn': mods['peagcn'].PEAGCNRecsysModel,
                 'sage': mods['peasage'].PEASageRecsysModel}[kind]

    class PEAModel(ModelBase):                  # name starts with 'PEA' -> eval(metapath_idx) honoured
        def update_graph_input(self, dataset):  # (models/base.py:91)
            return mpl

    kw = dict(entity_aware=entity_aware, entity_aware_coff=0.1, meta_path_steps=steps,
              if_use_features=False, channel_aggr=channel_aggr, dataset={'num_nodes': n},
              num_nodes=n, emb_dim=32, hidden_size=24, repr_dim=16, num_heads=heads, dropout=0)
    model = PEAModel(**kw)
    # trained-like magnitudes and NON-ZERO biases (fresh init has bias == 0)
    with torch.no_grad():
        for pname, p in model.named_parameters():
            if pname.endswith('bias'):
                p.copy_(torch.randn_like(p) * 0.1)
            elif pname != 'x':
                p.mul_(1.5)
    fx = {}
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        fx['param/' + k] = v
    for p, eil in enumerate(mpl):
        for s, ei in enumerate(eil):
            fx['edge/%d/%d' % (p, s)] = ei.numpy()
    meta = dict(kind=kind, heads=heads, channel_aggr=channel_aggr, steps=steps, num_nodes=n,
                emb_dim=32, hidden_size=24, repr_dim=16, entity_aware=entity_aware,
                gcn_deg_from='row', seed=seed)

    # per-channel outputs straight from the reference channel loop
    with torch.no_grad():
        for p, ch in enumerate(model.pea_channels):
            fx['out/channel/%d' % p] = ch(model.x, mpl[p]).numpy()
        # eval(): cached_repr, with and without ablation mask (models/base.py:88-96,194-195)
        model.eval()
        fx['out/repr'] = model.cached_repr.numpy().copy()
        model.eval(1)
        fx['out/repr_mask1'] = model.cached_repr.numpy().copy()
        model.eval()
        rng = np.random.default_rng(seed + 1)
        B = 64
        u = rng.integers(*blocks['u'], size=B)
        ip = rng.integers(*blocks['i'], size=B)
        ineg = rng.integers(*blocks['i'], size=B)
        batch = np.stack([u, ip, ineg], axis=1).astype(np.int64)
        bt = torch.from_numpy(batch)
        fx['in/batch'] = batch
        fx['out/pos'] = model.predict(bt[:, 0], bt[:, 1]).numpy().reshape(-1)
        fx['out/neg'] = model.predict(bt[:, 0], bt[:, 2]).numpy().reshape(-1)
        fx['out/loss_eval'] = np.float32(model.loss(bt).item())
    # training-mode loss recomputes the forward (models/base.py:44-45)
    model.train()
    if entity_aware:
        ent = rng.integers(0, n, size=(B, 6))
        mask = rng.integers(0, 2, size=(B, 2))
        batch9 = np.concatenate([batch, ent[:, 0:2], mask[:, 0:1], ent[:, 2:4], mask[:, 1:2]], axis=1)
        fx['in/batch9'] = batch9.astype(np.int64)
        fx['out/loss_train'] = np.float32(model.loss(torch.from_numpy(batch9.astype(np.int64))).item())
    else:
        fx['out/loss_train'] = np.float32(model.loss(bt).item())
    fx['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **fx)
    print(name, 'N=%d' % n, 'loss_train=%.6f' % float(fx['out/loss_train']),
          '%.0f kB' % (os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e3))


def make_rec_utils(mods):
    ru = mods['rec_utils']
    rng = np.random.default_rng(7)
    cases = []
    for pos_rank in [0, 3, 4, 5, 9, 10, 19, 20, 57, 99]:
        hv = np.zeros(100, dtype=bool)
        hv[pos_rank] = True
        pos = rng.normal(size=1).astype(np.float32)
        neg = rng.normal(size=99).astype(np.float32)
        cases.append(dict(hit_vec=hv.astype(int).tolist(), hit=[int(v) for v in ru.hit(hv)],
                          ndcg=[float(v) for v in ru.ndcg(hv)],
                          pos=pos.tolist(), neg=neg.tolist(), auc=float(ru.auc(pos, neg))))
    with open(os.path.join(OUT, 'rec_utils.json'), 'w') as f:
        json.dump(cases, f)
    print('rec_utils.json', len(cases), 'cases')


def make_rng_streams():
    """Legacy host RNG streams the reference's negative sampling draws from
    (datasets/movielens.py:920-937, solvers.py:29,123-127); stable across numpy versions."""
    out = {}
    np.random.seed(2020)
    out['randint_608_2729_x32'] = np.random.randint(low=608, high=608 + 2121, size=(32, 1)).reshape(-1).tolist()
    np.random.seed(2020)
    out['choice_100_200_x5'] = np.random.choice(list(range(100, 200)), size=(5,)).tolist()
    random.seed(2020)
    out['choices_100_200_k4'] = random.choices(list(range(100, 200)), k=4)
    # survey-recorded known answers as a cross-check of this container
    assert out['randint_608_2729_x32'][:8] == [1472, 1000, 2269, 2584, 765, 1168, 1792, 1528]
    assert out['choice_100_200_x5'] == [196, 108, 167, 167, 191]
    assert out['choices_100_200_k4'] == [161, 117, 176, 194]
    with open(os.path.join(OUT, 'rng_streams.json'), 'w') as f:
        json.dump(out, f)
    print('rng_streams.json ok')


def make_checkpoint_manifest():
    """Key / shape / dtype / checksum manifest of the six shipped checkpoints.
    Read with weights_only=True (nothing in the pickle is executed)."""
    root = os.path.join(REF, 'experiments', 'checkpoint', 'weights', 'Movielenslatest-small')
    import numpy._core.multiarray as ncm
    allow = [(ncm._reconstruct, 'numpy.core.multiarray._reconstruct'),
             (ncm.scalar, 'numpy.core.multiarray.scalar'),
             np.ndarray, np.dtype, type(np.dtype(np.float64))]
    man = {}
    for model in ('PEAGAT', 'PEAGCN', 'PEASage'):
        base = os.path.join(root, model, 'BPR')
        for d in sorted(os.listdir(base)):
            path = os.path.join(base, d, 'run_1', 'latest.pkl')
            with torch.serialization.safe_globals(allow):
                ck = torch.load(path, map_location='cpu', weights_only=True)
            sd = ck['model_states']['model']
            ea = "'entity_aware': True" in d
            man['%s/entity_aware=%s' % (model, ea)] = dict(
                epoch=int(ck['epoch']),
                keys={k: dict(shape=list(v.shape), dtype=str(v.dtype),
                              sum=float(v.double().sum()), abs_sum=float(v.double().abs().sum()))
                      for k, v in sd.items()})
    with open(os.path.join(OUT, 'checkpoint_manifest.json'), 'w') as f:
        json.dump(man, f, indent=0)
    print('checkpoint_manifest.json', {k: len(v['keys']) for k, v in man.items()})


class _FakeDataset:
    """The attributes the reference's sampling / evaluation code reads from a dataset object
    (datasets/movielens.py:879-997, solvers.py:21-31,50-54), filled from a tiny synthetic HIN."""

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
            self.test_pos_unid_inid_map[x] = [unseen[k]]           # leave-one-out positive
            self.neg_unid_inid_map[x] = unseen[:k] + unseen[k + 1:]
        self.cf_loss_type, self.entity_aware = 'BPR', False
        self.num_negative_samples = 4


def load_reference_sampling_and_solver(mods):
    """datasets/movielens.py (cf_negative_sampling) and solvers.py (generate_candidates, metrics) loaded file-by-file;
    the only foreign names they need at import time are torch_geometric.data.{download_url,extract_zip} (never called)
    and the parser entry points (never called)."""
    tgd = types.ModuleType('torch_geometric.data')
    tgd.download_url = tgd.extract_zip = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('offline'))
    sys.modules['torch_geometric.data'] = tgd
    sys.modules['torch_geometric'].data = tgd
    par = types.ModuleType('graph_recsys_benchmark.parser')
    par.parse_ml25m = par.parse_mlsmall = par.parse_yelp = None
    sys.modules['graph_recsys_benchmark.parser'] = par
    ds = types.ModuleType('graph_recsys_benchmark.datasets')
    ds.__path__ = [os.path.join(REF, 'graph_recsys_benchmark', 'datasets')]
    sys.modules['graph_recsys_benchmark.datasets'] = ds
    for alias, typ in (('long', np.int64), ('str', str), ('int', int)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)          # aliases removed in numpy 1.24 (datasets/movielens.py:35-54,935)
    ml = importlib.import_module('graph_recsys_benchmark.datasets.movielens')
    up = sys.modules['graph_recsys_benchmark.utils']
    up.hit, up.ndcg, up.auc = mods['rec_utils'].hit, mods['rec_utils'].ndcg, mods['rec_utils'].auc
    solvers = importlib.import_module('graph_recsys_benchmark.solvers')
    return ml, solvers


def make_sampling_and_metrics(mods):
    ml, solvers = load_reference_sampling_and_solver(mods)
    out = {}
    # --- BPR negative sampling, both strategies (datasets/movielens.py:920-940,994-997)
    for strategy in ('random', 'unseen'):
        fake = _FakeDataset(11)
        fake.sampling_strategy = strategy
        random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)
        ml.MovieLens.cf_negative_sampling(fake)
        out['train_data_' + strategy] = fake.train_data.numpy()
    # --- evaluation loop (solvers.py:33-104) on a CPU reference model
    fake = _FakeDataset(12)
    n = fake.num_uids + fake.num_iids + 6
    u2i = torch.from_numpy(fake.edge_index_nps['user2item']).long()
    attr = torch.stack([torch.randint(n - 6, n, (80,), generator=torch.Generator().manual_seed(3)),
                        fake.num_uids + torch.randint(0, fake.num_iids, (80,), generator=torch.Generator().manual_seed(4))])
    mpl = [[u2i, torch.flip(u2i, dims=[0])], [torch.flip(u2i, dims=[0]), u2i], [attr, torch.flip(u2i, dims=[0])]]

    class PEAModel(mods['peagat'].PEAGATRecsysModel):
        def update_graph_input(self, dataset):
            return mpl

    torch.manual_seed(5)
    model = PEAModel(entity_aware=False, entity_aware_coff=0.1, meta_path_steps=[2, 2, 2], if_use_features=False,
                     channel_aggr='att', dataset={'num_nodes': n}, num_nodes=n, emb_dim=32, hidden_size=24, repr_dim=16,
                     num_heads=1, dropout=0)
    with torch.no_grad():
        for pname, p in model.named_parameters():
            if pname.endswith('bias'):
                p.copy_(torch.randn_like(p) * 0.1)
            elif pname != 'x':
                p.mul_(2.0)
    solver = solvers.BaseSolver(PEAModel, {}, {'model_type': 'Graph'}, {'num_neg_candidates': 99, 'device': 'cpu'})
    model.eval()
    np.random.seed(2021)
    with torch.no_grad():
        hr, nd, auc, loss = solver.metrics(1, 1, model, fake)
    out.update(metrics_HR=hr, metrics_NDCG=nd, metrics_AUC=auc, metrics_loss=loss)
    for k, v in model.state_dict().items():
        out['metrics_param/' + k] = v.numpy()
    for p, eil in enumerate(mpl):
        for s_, ei in enumerate(eil):
            out['metrics_edge/%d/%d' % (p, s_)] = ei.numpy()
    out['metrics_num_nodes'] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, 'sampling_metrics.npz'), **out)
    print('sampling_metrics.npz', {k: np.asarray(v).shape for k, v in out.items() if not k.startswith('metrics_param') and not k.startswith('metrics_edge')},
          'HR@10 %.4f' % hr[5])


def entity_fake_dataset(seed=13):
    """_FakeDataset + the entity lists the reference builds from its CSV files (datasets/movielens.py:941-991):
    per item / per user a list of entity node ids (some empty), nid -> (type, entity) and per-type id blocks."""
    fake = _FakeDataset(seed)
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


def make_entity_rows(mods):
    """The six entity columns of a training row: the reference's own Dataset.__getitem__
    (datasets/movielens.py:1147-1181) on a fake dataset, main process, seeded `random`."""
    ml, _ = load_reference_sampling_and_solver(mods)
    fake = entity_fake_dataset()
    fake.sampling_strategy = 'random'
    random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)
    ml.MovieLens.cf_negative_sampling(fake)           # entity lists already attached: the CSV branch is skipped
    random.seed(77)
    rows = torch.stack([ml.MovieLens.__getitem__(fake, i) for i in range(200)])
    np.savez_compressed(os.path.join(OUT, 'entity_rows.npz'), rows=rows.numpy(), train_data=fake.train_data.numpy())
    print('entity_rows.npz', tuple(rows.shape), 'masked rows', int((rows[:, 5] == 0).sum()), int((rows[:, 8] == 0).sum()))


def make_nn_convs():
    """The reference's OWN conv classes (graph_recsys_benchmark/nn/{kgat,kgcn,ngcf}_conv.py) run for real; the only
    foreign pieces are the MessagePassing base / remove_self_loops (restated, oracle/pyg_restatement.py)."""
    conv = types.ModuleType('torch_geometric.nn.conv')
    conv.MessagePassing = R.MessagePassing
    utils = types.ModuleType('torch_geometric.utils')
    utils.remove_self_loops = R.remove_self_loops
    sys.modules.update({'torch_geometric.nn.conv': conv, 'torch_geometric.utils': utils})
    nnp = types.ModuleType('graph_recsys_benchmark.nn')
    nnp.__path__ = [os.path.join(REF, 'graph_recsys_benchmark', 'nn')]
    sys.modules['graph_recsys_benchmark.nn'] = nnp
    kgat = importlib.import_module('graph_recsys_benchmark.nn.kgat_conv').KGATConv
    kgcn = importlib.import_module('graph_recsys_benchmark.nn.kgcn_conv').KGCNConv
    ngcf = importlib.import_module('graph_recsys_benchmark.nn.ngcf_conv').NGCFConv
    rng = np.random.default_rng(77)
    n, e = 150, 1500
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    keep = src != dst
    half = np.stack([src[keep], dst[keep]])
    half = np.concatenate([half, np.stack([np.arange(1, n), np.zeros(n - 1, np.int64)])], axis=1)   # node 0 is a hub
    ei = np.concatenate([half, half[::-1]], axis=1).astype(np.int64)            # both directions, like the KG
    x = rng.normal(size=(n, 32)).astype(np.float32) * 0.3
    att = rng.random(ei.shape[1]).astype(np.float32)
    out = {'x': x, 'edge_index': ei, 'att_map': att}
    torch.manual_seed(9)
    eit, xt, attt = torch.from_numpy(ei), torch.from_numpy(x), torch.from_numpy(att)
    for name, cls, args in (('kgat', kgat, (xt, eit, attt)), ('kgcn', kgcn, (xt, eit, attt)), ('ngcf', ngcf, (xt, eit))):
        m = cls(32, 16)
        with torch.no_grad():
            for pn, p_ in m.named_parameters():
                if pn == 'bias':
                    p_.copy_(torch.randn_like(p_) * 0.1)
            out[name + '/out'] = m(*args).numpy()
        for pn, p_ in m.named_parameters():
            out[name + '/param/' + pn] = p_.detach().numpy()
    # NGCF only: an edge list WITH self loops -- the reference strips them (nn/ngcf_conv.py:33-34) before it counts
    # degrees and propagates (KGAT / KGCN cannot take such input: att_map would no longer line up with the edges)
    loops = np.stack([np.arange(0, n, 7), np.arange(0, n, 7)]).astype(np.int64)
    ei_loop = np.concatenate([ei[:, :700], loops, ei[:, 700:]], axis=1)
    out['ngcf_selfloop/edge_index'] = ei_loop
    m = ngcf(32, 16)
    with torch.no_grad():
        out['ngcf_selfloop/out'] = m(xt, torch.from_numpy(ei_loop)).numpy()
    for pn, p_ in m.named_parameters():
        out['ngcf_selfloop/param/' + pn] = p_.detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'nn_convs.npz'), **out)
    print('nn_convs.npz', {k: v.shape for k, v in out.items() if k.endswith('/out')})


def make_graph_input_tables():
    """A8: the reference's own update_pea_graph_input (utils/general_utils.py:280-395) on a fake dataset whose relations
    are tiny marker arrays: which relation, flipped or not, at every (metapath, step), for each dataset branch.  The
    module's `from ..datasets import MovieLens, Yelp` is satisfied by two placeholder names (never used by this function)."""
    import json
    ds_stub = types.ModuleType('graph_recsys_benchmark.datasets')
    ds_stub.MovieLens, ds_stub.Yelp = type('MovieLens', (), {}), type('Yelp', (), {})
    sys.modules['graph_recsys_benchmark.datasets'] = ds_stub
    gu = importlib.import_module('graph_recsys_benchmark.utils.general_utils')

    class Markers(dict):
        """edge_index_nps: every relation asked for gets its own 2-edge marker array."""
        def __init__(self):
            super().__init__()
            self.names = []

        def __missing__(self, key):
            r = len(self.names)
            self.names.append(key)
            self[key] = np.array([[1000 * r + 1, 1000 * r + 2], [1000 * r + 501, 1000 * r + 502]], dtype=np.float64)
            return self[key]

    out = {}
    for tag, dargs in (('Movielens/latest-small', {'dataset': 'Movielens', 'name': 'latest-small'}),
                       ('Movielens/25m', {'dataset': 'Movielens', 'name': '25m'}),
                       ('Yelp', {'dataset': 'Yelp', 'name': ''})):
        fake = types.SimpleNamespace(edge_index_nps=Markers())
        lists = gu.update_pea_graph_input(dargs, {'device': 'cpu'}, fake)
        table = []
        for steps in lists:
            row = []
            for t in steps:
                a = t.numpy()
                flipped = bool(a[0, 0] % 1000 > 500)
                r = int(a[1 if flipped else 0, 0] // 1000)
                assert a.dtype == np.int64 and a.shape == (2, 2)
                row.append([fake.edge_index_nps.names[r], int(flipped)])
            table.append(row)
        out[tag] = table
    with open(os.path.join(OUT, 'graph_input_tables.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('graph_input_tables.json', {k: len(v) for k, v in out.items()})


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    mods = load_reference_models()
    make_case(mods, 'pea_gat_p5s2_h1_att', 'gat', 'p5s2', 1, 'att', 2020)
    make_case(mods, 'pea_gat_p3deep_h2_att', 'gat', 'p3deep', 2, 'att', 2021)
    make_case(mods, 'pea_gat_p3mixed_h1_att', 'gat', 'p3mixed', 1, 'att', 2028)
    make_case(mods, 'pea_gat_p5s2_h1_mean', 'gat', 'p5s2', 1, 'mean', 2022)
    make_case(mods, 'pea_gcn_p5s2_att', 'gcn', 'p5s2', 1, 'att', 2023)
    make_case(mods, 'pea_gcn_p3mixed_mean', 'gcn', 'p3mixed', 1, 'mean', 2024)
    make_case(mods, 'pea_sage_p5s2_att', 'sage', 'p5s2', 1, 'att', 2025)
    make_case(mods, 'pea_sage_p3mixed_att', 'sage', 'p3mixed', 1, 'att', 2026)
    make_case(mods, 'pea_gat_p5s2_h1_att_ea', 'gat', 'p5s2', 1, 'att', 2027, entity_aware=True)
    make_rec_utils(mods)
    make_rng_streams()
    make_checkpoint_manifest()
    make_sampling_and_metrics(mods)
    make_entity_rows(mods)
    make_nn_convs()
    make_graph_input_tables()
