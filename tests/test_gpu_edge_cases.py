"""GPU: edge cases of the path -- empty and degenerate relations, tiny graphs, every lane-group width, error codes."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import assert_fp32_close, assert_fused_close, build_model, f64_forward, random_state_dict
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _check(kind, n, edges, steps, emb, hidden, repr_dim, heads=1, aggr='att', seed=3):
    model = build_model(kind, n, edges, steps, emb, hidden, repr_dim, heads=heads, channel_aggr=aggr)
    model.load_state_dict(random_state_dict(model, seed))
    model.eval()
    with torch.no_grad():
        fused, stack = model.forward(return_stack=True)
    sd = {k: _np(v) for k, v in model.state_dict().items()}
    cps, hls = [], []
    for p, S in enumerate(steps):
        cps.append([{k[len('pea_channels.%d.gnn_layers.%d.' % (p, s)):]: v for k, v in sd.items()
                     if k.startswith('pea_channels.%d.gnn_layers.%d.' % (p, s))} for s in range(S)])
        hls.append([1] * S if kind != 'gat' else ([heads] * (S - 1) + [1] if S > 1 else [heads]))
    want, wstack = orc.pea_forward(kind, sd['x'], edges, cps, hls, att=sd.get('att'), channel_aggr=aggr, return_stack=True)
    t_fused, t_stack = f64_forward(kind, sd, edges, steps, heads, aggr)
    assert_fp32_close(_np(stack), wstack, t_stack, what='stack')
    assert_fused_close(_np(fused), _np(stack), want, t_fused, sd.get('att'), aggr, truth_stack=t_stack)
    return model


@pytest.mark.parametrize('kind', ['gat', 'gcn', 'sage'])
def test_empty_and_self_loop_only_relations(kind):
    n = 37
    rng = np.random.default_rng(0)
    empty = np.zeros((2, 0), np.int64)
    loops = np.stack([np.arange(10), np.arange(10)]).astype(np.int64)        # only self loops (dropped by GAT/GCN)
    some = np.stack([rng.integers(0, n, 90), rng.integers(0, n, 90)]).astype(np.int64)
    _check(kind, n, [[empty, some], [loops, empty], [some, loops]], [2, 2, 2], 16, 8, 8)


@pytest.mark.parametrize('kind', ['gat', 'gcn', 'sage'])
def test_tiny_graph_and_one_step_channels(kind):
    ei = np.array([[0, 1, 2, 2, 4], [1, 0, 0, 3, 4]], np.int64)
    _check(kind, 5, [[ei], [np.ascontiguousarray(ei[::-1])]], [1, 1], 8, 8, 4)


@pytest.mark.parametrize('emb,hidden,repr_dim,heads', [(4, 4, 4, 1), (8, 12, 8, 1), (128, 128, 16, 1), (64, 256, 16, 1),
                                                       (16, 8, 4, 4), (32, 64, 16, 4)])
def test_every_lane_group_width(emb, hidden, repr_dim, heads):
    """G = 4 ... 64 lanes per row chunk, head widths 4 ... 256 (incl. non-power-of-two 12), up to 4 heads."""
    rng = np.random.default_rng(1)
    n = 300
    a = np.stack([rng.integers(0, n, 4000), rng.integers(0, 40, 4000)]).astype(np.int64)   # 40 hot destinations
    b = np.ascontiguousarray(a[::-1])
    for kind in ('gat', 'sage'):
        _check(kind, n, [[a, b], [b, a], [a, b]], [2, 2, 2], emb, hidden, repr_dim, heads=heads if kind == 'gat' else 1)


def test_error_codes_and_messages():
    from graph_recsys_benchmark_amd import _lib
    from graph_recsys_benchmark_amd.engine import GraphPlan, PEAEngine
    lib = _lib.load()
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]], dtype=torch.int64).cuda()
    plan = GraphPlan(4, [[ei, ei]], True)
    with pytest.raises(_lib.PeaError) as e:            # widths must be multiples of 4
        PEAEngine(plan, 'gat', [2], 6, 8, 4)
    assert e.value.code == -1 and 'multiples of 4' in str(e.value)
    eng = PEAEngine(plan, 'gat', [2], 8, 8, 4)
    x = torch.zeros(4, 8, device='cuda')
    params = [(torch.zeros(8, 8, device='cuda'), torch.zeros(1, 1, 8, device='cuda'), torch.zeros(1, 1, 8, device='cuda'),
               torch.zeros(8, device='cuda')),
              (torch.zeros(4, 8, device='cuda'), torch.zeros(1, 1, 4, device='cuda'), torch.zeros(1, 1, 4, device='cuda'),
               torch.zeros(4, device='cuda'))]
    out = eng.forward(params, x, att=torch.zeros(1, 1, 4, device='cuda'))
    assert out.shape == (4, 4) and torch.isfinite(out).all()
    with pytest.raises(_lib.PeaError):                  # masked channel out of range
        eng.forward(params, x, att=torch.zeros(1, 1, 4, device='cuda'), masked=3)
    ptrs = (C.c_void_p * 8)()
    rc = lib.pea_model_forward(eng._h, ptrs, _lib.ptr(x), None, -1, _lib.ptr(eng._ws), 16, _lib.ptr(out), None, None)
    assert rc == -4 and b'workspace' in lib.pea_last_error()      # PEA_ERR_NOMEM
    rc = lib.pea_model_forward(eng._h, ptrs, _lib.ptr(x), _lib.ptr(x), -1, _lib.ptr(eng._ws), eng.workspace_bytes,
                               _lib.ptr(out), None, None)
    assert rc == -1 and b'null weight' in lib.pea_last_error()    # null parameter pointer
    with pytest.raises(ValueError):
        GraphPlan(4, [[ei.to(torch.int32)]], True)
    with pytest.raises(RuntimeError):
        GraphPlan(4, [[ei.cpu()]], True)


def test_graphed_forward_replays_the_same_result():
    """PEAEngine.forward_graphed: the schedule captured into a hipGraph gives the eager result bit for bit, and keeps
    doing so after the parameters were changed in place (what an optimizer step does)."""
    from helpers import GoldenCase, model_from_golden
    g = GoldenCase('pea_gat_p5s2_h1_att')
    model = model_from_golden(g)
    model.eval()
    eng = model._get_engine()
    with torch.no_grad():
        params, x, att = model._layer_params(), model.x.detach(), model.att
        want = eng.forward(params, x, att=att).clone()
        got = eng.forward_graphed(params, x, att=att)
        assert torch.equal(got, want)
        for p in model.parameters():
            p.mul_(1.01)
        want2 = eng.forward(params, x, att=att).clone()
        got2 = eng.forward_graphed(params, x, att=att)
        assert got2.data_ptr() == got.data_ptr() and torch.equal(got2, want2) and not torch.equal(want2, want)


def _kernel_names_of_one_forward(model):
    import ctypes as C
    from graph_recsys_benchmark_amd import _lib
    lib = _lib.load()
    lib.pea_profile_enable(1)
    with torch.no_grad():
        model.forward()
    torch.cuda.synchronize()
    lib.pea_profile_enable(0)
    cap = 4096
    names, cnt = C.create_string_buffer(cap * 32), C.c_int()
    lib.pea_profile_read(cap, names, None, None, C.byref(cnt))
    return {names.raw[i * 32:(i + 1) * 32].split(b'\0')[0].decode() for i in range(cnt.value)}


@pytest.mark.parametrize('kind', ['gat', 'gcn', 'sage'])
def test_lds_staged_hot_sources_are_bitwise_the_plain_kernel(kind, monkeypatch):
    """Relations whose few hottest sources carry a large share of the edges (Zipf item popularity) get those rows staged in
    an LDS image by the long-row kernel (csrc/agg.hip: agg_long_hot_kernel).  Same edges, same order, same values: the
    result must be BITWISE the plain kernel's (PEA_HOT=0), on a graph with hub rows, multi-edges and every row bin."""
    from helpers import random_hin
    monkeypatch.setenv('PEA_HOT_MIN_EDGES', '1000')
    monkeypatch.setenv('PEA_FAT', '0')      # compare like with like: the LDS variant is built on the thin (4 columns per lane) kernel
    n, blocks, rel = random_hin(17, n_user=6000, n_item=500, n_attr=30, e_u2i=150000, e_attr=2500)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [a2i, flip(u2i)], [flip(u2i), u2i], [flip(a2i), flip(u2i)]]
    steps = [2, 2, 2, 2]
    model = build_model(kind, n, edges, steps, 64, 64, 16)
    model.load_state_dict(random_state_dict(model, 12))
    model.eval()
    monkeypatch.setenv('PEA_HOT', '1')
    names = _kernel_names_of_one_forward(model)
    assert any(nm.startswith('agg_longhot_') for nm in names), names       # the item -> user relation took the LDS path
    with torch.no_grad():
        hot_fused, hot_stack = model.forward(return_stack=True)
    monkeypatch.setenv('PEA_HOT', '0')
    assert not any(nm.startswith('agg_longhot_') for nm in _kernel_names_of_one_forward(model))
    with torch.no_grad():
        fused, stack = model.forward(return_stack=True)
    assert torch.equal(hot_stack, stack) and torch.equal(hot_fused, fused)
    # and both agree with the CPU oracle
    sd = {k: _np(v) for k, v in model.state_dict().items()}
    cps = [[{k[len('pea_channels.%d.gnn_layers.%d.' % (p, s)):]: v for k, v in sd.items()
             if k.startswith('pea_channels.%d.gnn_layers.%d.' % (p, s))} for s in range(2)] for p in range(4)]
    want, wstack = orc.pea_forward(kind, sd['x'], edges, cps, [[1, 1]] * 4, att=sd.get('att'), return_stack=True)
    t_fused, t_stack = f64_forward(kind, sd, edges, steps, 1, 'att')
    assert_fp32_close(_np(hot_stack), wstack, t_stack, what='stack')
    assert_fused_close(_np(hot_fused), _np(hot_stack), want, t_fused, sd.get('att'), truth_stack=t_stack)


@pytest.mark.parametrize('kind', ['gat', 'gcn', 'sage'])
def test_fat_lane_long_rows_match_the_thin_kernel_and_the_oracle(kind, monkeypatch):
    """Groups whose heads are >= 64 columns wide run their long rows on the fat-lane kernel (16 columns per lane, chunks
    of a head interleaved over its lanes; csrc/agg.hip: agg_long_fat_kernel).  Same edges, same order, same softmax
    batches; only the grouping of columns inside a lane's dot product differs, so it agrees with the thin kernel
    (PEA_FAT=0) to fp32 rounding and with the oracle to the usual tolerance -- on hub rows, multi-edges, 1 and 2 heads."""
    from helpers import random_hin
    n, blocks, rel = random_hin(19, n_user=5000, n_item=450, n_attr=30, e_u2i=120000, e_attr=2500)
    u2i, a2i = rel['u2i'], rel['a2i']
    flip = lambda e: np.ascontiguousarray(e[::-1])
    edges = [[u2i, flip(u2i)], [a2i, flip(u2i)], [flip(u2i), u2i]]
    steps = [2, 2, 2]
    heads = 2 if kind == 'gat' else 1
    model = build_model(kind, n, edges, steps, 64, 64, 16, heads=heads)
    model.load_state_dict(random_state_dict(model, 13))
    model.eval()
    monkeypatch.setenv('PEA_FAT', '1')
    names = _kernel_names_of_one_forward(model)
    assert any(nm.startswith('agg_long_fat_') for nm in names), names
    with torch.no_grad():
        fat_fused, fat_stack = model.forward(return_stack=True)
    monkeypatch.setenv('PEA_FAT', '0')
    assert not any(nm.startswith('agg_long_fat_') for nm in _kernel_names_of_one_forward(model))
    with torch.no_grad():
        fused, stack = model.forward(return_stack=True)
    scale = float(stack.abs().max())
    assert float((fat_stack - stack).abs().max()) <= 2e-6 * scale
    sd = {k: _np(v) for k, v in model.state_dict().items()}
    cps = [[{k[len('pea_channels.%d.gnn_layers.%d.' % (p, s)):]: v for k, v in sd.items()
             if k.startswith('pea_channels.%d.gnn_layers.%d.' % (p, s))} for s in range(2)] for p in range(3)]
    hls = [[heads, 1]] * 3
    want, wstack = orc.pea_forward(kind, sd['x'], edges, cps, hls, att=sd.get('att'), return_stack=True)
    t_fused, t_stack = f64_forward(kind, sd, edges, steps, heads, 'att')
    assert_fp32_close(_np(fat_stack), wstack, t_stack, what='stack')
    assert_fused_close(_np(fat_fused), _np(fat_stack), want, t_fused, sd.get('att'), truth_stack=t_stack)
