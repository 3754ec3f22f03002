"""GPU: loss.backward() through the HIP conv stack (csrc/agg_bwd.hip + model_bwd.hip + autograd.py) against torch
autograd on the float64 torch restatement of the same model (oracle/pyg_restatement.py assembled like the reference's
models/base.py).  Tolerance: fp32 gradients vs float64 autograd, rtol 2e-4 of the gradient's max magnitude."""
import numpy as np
import pytest
import torch

from helpers import build_model, random_hin, random_state_dict

pytestmark = pytest.mark.gpu


def f64_loss_and_grads(kind, sd, edges, steps, heads, aggr, batch):
    from oracle import pyg_restatement as R
    params = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in sd.items()}
    x = params['x']
    outs = []
    for p, S in enumerate(steps):
        h = x
        for s in range(S):
            pre = 'pea_channels.%d.gnn_layers.%d.' % (p, s)
            ei = torch.from_numpy(np.ascontiguousarray(edges[p][s]))
            last = s == S - 1
            if kind == 'gat':
                hh = 1 if (S > 1 and last) else heads
                conv = R.GATConv(h.shape[1], sd[pre + 'lin.weight'].shape[0] // hh, heads=hh).double()
                conv.lin.weight, conv.att_i, conv.att_j, conv.bias = None, None, None, None
                del conv._parameters['att_i'], conv._parameters['att_j'], conv._parameters['bias']
                del conv.lin._parameters['weight']
                conv.lin.weight = params[pre + 'lin.weight']
                conv.att_i, conv.att_j, conv.bias = params[pre + 'att_i'], params[pre + 'att_j'], params[pre + 'bias']
            elif kind == 'gcn':
                conv = R.GCNConv(h.shape[1], sd[pre + 'weight'].shape[1]).double()
                del conv._parameters['weight'], conv._parameters['bias']
                conv.weight, conv.bias = params[pre + 'weight'], params[pre + 'bias']
            else:
                conv = R.SAGEConv(h.shape[1], sd[pre + 'lin_rel.weight'].shape[0]).double()
                del conv.lin_rel._parameters['weight'], conv.lin_rel._parameters['bias'], conv.lin_root._parameters['weight']
                conv.lin_rel.weight, conv.lin_rel.bias = params[pre + 'lin_rel.weight'], params[pre + 'lin_rel.bias']
                conv.lin_root.weight = params[pre + 'lin_root.weight']
            h = conv(h, ei)
            if not last:
                h = torch.relu(h)
        outs.append(h)
    stack = torch.stack(outs, dim=1)
    if aggr == 'att':
        w = torch.softmax((stack * params['att']).sum(-1), dim=-1).unsqueeze(-1)
        fused = (stack * w).sum(1)
    else:
        fused = stack.mean(1)
    b = torch.from_numpy(batch)

    def pred(u, i):
        z = torch.cat([fused[u], fused[i]], dim=-1)
        hdn = torch.relu(z @ params['fc1.weight'].t() + params['fc1.bias'])
        return hdn @ params['fc2.weight'].t() + params['fc2.bias']

    loss = -(pred(b[:, 0], b[:, 1]) - pred(b[:, 0], b[:, 2])).sigmoid().log().sum()
    loss.backward()
    return float(loss), {k: v.grad.numpy() for k, v in params.items() if v.grad is not None}


@pytest.mark.parametrize('kind,heads,aggr', [('gcn', 1, 'att'), ('sage', 1, 'att'), ('gat', 1, 'att'), ('gat', 2, 'mean')])
def test_backward_matches_float64_autograd(kind, heads, aggr):
    n, blocks, rel = random_hin(41, n_user=1500, n_item=400, n_attr=30, e_u2i=20000, e_attr=1500)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)], [flip(a2i), a2i, flip(u2i)]]
    steps = [2, 2, 2, 3]
    model = build_model(kind, n, edges, steps, 32, 32, 16, heads=heads, channel_aggr=aggr)
    model.load_state_dict(random_state_dict(model, 8, scale=0.25))
    rng = np.random.default_rng(2)
    batch = np.stack([rng.integers(*blocks['u'], size=512), rng.integers(*blocks['i'], size=512),
                      rng.integers(*blocks['i'], size=512)], axis=1).astype(np.int64)
    model.train()
    model.zero_grad()
    loss = model.loss(torch.from_numpy(batch).cuda())
    loss.backward()
    with torch.no_grad():                       # training-mode loss() leaves the fused table of the same forward behind
        assert not model.cached_repr.requires_grad
        torch.testing.assert_close(model.cached_repr, model.forward(), rtol=1e-5, atol=1e-7)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    want_loss, want = f64_loss_and_grads(kind, sd, edges, steps, heads, aggr, batch)
    np.testing.assert_allclose(float(loss), want_loss, rtol=2e-5)
    checked = 0
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        g, w = p.grad.detach().cpu().numpy().astype(np.float64), want[name]
        scale = max(np.abs(w).max(), 1e-12)
        err = np.abs(g - w).max()
        assert err <= 2e-4 * scale + 1e-9, '%s: max err %.3e vs scale %.3e' % (name, err, scale)
        checked += 1
    assert checked == len(want)
    # an optimizer step runs end to end (solvers.py:213-216)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    opt.step()
    with torch.no_grad():
        l2 = float(model.loss(torch.from_numpy(batch).cuda()))
    assert np.isfinite(l2)


@pytest.mark.parametrize('kind,heads,aggr', [('gcn', 1, 'att'), ('sage', 1, 'att'), ('gat', 1, 'att'), ('gat', 2, 'mean')])
def test_conv_dropins_train_through_the_reference_channel_loop(kind, heads, aggr):
    """INTEGRATION.md's first switch: keep the reference's OWN model code (channel loop models/base.py:134-140, stack and
    fusion :191-203, predict :208-214, loss :43-48 -- restated here with plain torch ops exactly as the reference writes
    them) and only swap the conv classes.  loss.backward() then runs through the per-layer drop-ins' HIP backward;
    gradients of every parameter against float64 autograd."""
    n, blocks, rel = random_hin(43, n_user=1200, n_item=350, n_attr=25, e_u2i=16000, e_attr=1200)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [a2i, flip(u2i)], [flip(a2i), a2i, flip(u2i)], [flip(u2i)]]
    steps = [2, 2, 3, 1] if not (kind == 'gat' and heads > 1) else [2, 2, 3]   # a 1-step multi-head channel cannot be stacked
    edges = edges[:len(steps)]
    model = build_model(kind, n, edges, steps, 32, 32, 16, heads=heads, channel_aggr=aggr)
    model.load_state_dict(random_state_dict(model, 9, scale=0.25))
    rng = np.random.default_rng(4)
    batch = np.stack([rng.integers(*blocks['u'], size=384), rng.integers(*blocks['i'], size=384),
                      rng.integers(*blocks['i'], size=384)], axis=1).astype(np.int64)
    bt = torch.from_numpy(batch).cuda()
    model.train()
    model.zero_grad()
    # --- the reference's forward / predict / loss, line for line in torch ops, over the drop-in conv modules
    x = [channel(model.x, eil).unsqueeze(1)                                   # PEABaseChannel.forward: relu between steps
         for channel, eil in zip(model.pea_channels, model.meta_path_edge_index_list)]
    x = torch.cat(x, dim=1)
    if aggr == 'att':
        atts = torch.softmax(torch.sum(x * model.att, dim=-1), dim=-1).unsqueeze(-1)
        cached = torch.sum(x * atts, dim=1)
    else:
        cached = x.mean(dim=1)

    def predict(u, i):
        z = torch.cat([cached[u], cached[i]], dim=-1)
        return model.fc2(torch.relu(model.fc1(z)))

    loss = -(predict(bt[:, 0], bt[:, 1]) - predict(bt[:, 0], bt[:, 2])).sigmoid().log().sum()
    loss.backward()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    want_loss, want = f64_loss_and_grads(kind, sd, edges, steps, heads, aggr, batch)
    np.testing.assert_allclose(float(loss), want_loss, rtol=2e-5)
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        g, w = p.grad.detach().cpu().numpy().astype(np.float64), want[name]
        scale = max(np.abs(w).max(), 1e-12)
        err = np.abs(g - w).max()
        assert err <= 2e-4 * scale + 1e-9, '%s: max err %.3e vs scale %.3e' % (name, err, scale)
    # same loss as the whole-model schedule on the same parameters
    with torch.no_grad():
        np.testing.assert_allclose(float(model.loss(bt)), float(loss), rtol=2e-5)


@pytest.mark.parametrize('kind,width', [('gat', 64), ('gat', 128), ('gcn', 64), ('gcn', 128), ('sage', 64), ('sage', 128)])
def test_two_step_training_schedule_matches_float64_and_the_levelwise_schedule(kind, width, monkeypatch):
    """GAT (one head), GCN and SAGE models of 2-step channels with emb == hidden in {64, 128} train on the two-step schedule
    (csrc/model.h: fused2_train): the first layer aggregates x, csrc/mlp2.hip chains both transforms (keeping the hidden tile),
    and the first layer's backward runs in x space.  Every gradient against float64 autograd, and against the level-wise
    schedule (PEA_FUSED2_TRAIN=0) on the same parameters; the model must really be on the two-step path."""
    n, blocks, rel = random_hin(47, n_user=1400, n_item=380, n_attr=25, e_u2i=18000, e_attr=1400)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)], [flip(a2i), a2i]]
    steps = [2, 2, 2, 2]
    rng = np.random.default_rng(5)
    batch = np.stack([rng.integers(*blocks['u'], size=400), rng.integers(*blocks['i'], size=400),
                      rng.integers(*blocks['i'], size=400)], axis=1).astype(np.int64)
    bt = torch.from_numpy(batch).cuda()
    results = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('PEA_FUSED2_TRAIN', mode)
        model = build_model(kind, n, edges, steps, width, width, 16)
        model.load_state_dict(random_state_dict(model, 12, scale=0.2))
        model.train()
        model.zero_grad()
        loss = model.loss(bt)
        loss.backward()
        from graph_recsys_benchmark_amd.autograd import _Layout
        assert _Layout(model._train_engine).two_step_train == (mode == '1')
        results[mode] = (float(loss), {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in model.named_parameters()})
        if mode == '1':
            with torch.no_grad():               # the fused table left behind by the training forward
                torch.testing.assert_close(model.cached_repr, model.forward(), rtol=1e-5, atol=1e-7)
            sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    want_loss, want = f64_loss_and_grads(kind, sd, edges, steps, 1, 'att', batch)
    for mode, (loss, grads) in results.items():
        np.testing.assert_allclose(loss, want_loss, rtol=2e-5)
        g_max = max(np.abs(w).max() for w in want.values())
        for name, w in want.items():
            err = np.abs(grads[name] - w).max()
            assert err <= 2e-4 * np.abs(w).max() + 1e-6 * g_max, 'schedule %s, %s: max err %.3e vs scale %.3e' % (mode, name, err, np.abs(w).max())


@pytest.mark.parametrize('kind', ['gat', 'gcn', 'sage'])
def test_sparse_backward_equals_the_dense_one_over_several_steps(kind, monkeypatch):
    """The training backward walks the gradient's support (rows with a non-zero dT_1 -- SAGE: or in the batch --, compacted on
    the device: csrc/rows.hip) and keeps dA_0 (SAGE: dM_0 and the root blocks) zero outside it from step to step.  Three steps
    with DIFFERENT batches on one model (so rows enter and leave the support) against the same steps with PEA_SPARSE_BWD=0:
    the skipped rows contribute exact zeros, so every gradient agrees to fp32 summation order (1e-5 of the tensor's scale;
    gradients that vanish analytically: of the largest gradient)."""
    n, blocks, rel = random_hin(53, n_user=1600, n_item=420, n_attr=30, e_u2i=12000, e_attr=900)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [a2i, flip(u2i)], [flip(a2i), a2i]]
    rng = np.random.default_rng(11)
    batches = [torch.from_numpy(np.stack([rng.integers(*blocks['u'], size=24), rng.integers(*blocks['i'], size=24),
                                          rng.integers(*blocks['i'], size=24)], axis=1).astype(np.int64)).cuda() for _ in range(3)]
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('PEA_SPARSE_BWD', mode)
        model = build_model(kind, n, edges, [2, 2, 2], 64, 64, 16)
        model.load_state_dict(random_state_dict(model, 21, scale=0.2))
        model.train()
        steps = []
        for bt in batches:
            model.zero_grad()
            loss = model.loss(bt)
            loss.backward()
            steps.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
        out[mode] = steps
        if mode == '1':
            live = model._train_engine._live_rows
            torch.cuda.synchronize()
            assert 0 < int(live.count.item()) < n           # the support is a proper subset of the nodes here
    for (l1, g1), (l0, g0) in zip(out['1'], out['0']):
        assert l1 == l0
        g_max = max(float(v.abs().max()) for v in g0.values())
        for k in g0:
            scale = float(g0[k].abs().max())
            assert float((g1[k] - g0[k]).abs().max()) <= 1e-5 * scale + 1e-7 * g_max + 1e-12, k
