"""GPU: the dense half of the backward (csrc/dense_bwd.hip) against float64 torch products -- the GEMMs the reference
leaves to autograd (torch.nn.Linear / matmul inside the PyG convs, models/base.py:138-139 under loss.backward())."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(got, want64, rtol=2e-5):
    scale = float(want64.abs().max()) + 1e-30
    err = float((got.double() - want64).abs().max())
    assert err <= rtol * scale, 'max err %.3e vs scale %.3e' % (err, scale)


@pytest.mark.parametrize('n', [1, 37, 4099, 150001])
def test_grad_weight_matches_float64(n, monkeypatch):
    monkeypatch.setenv('PEA_GW128', '1' if n % 2 else '0')       # the (off by default) 128 x 128 kernel on the odd sizes
    from graph_recsys_benchmark_amd.engine import grad_weight
    g = torch.Generator(device='cuda').manual_seed(n)
    big = torch.randn(n, 200, generator=g, device='cuda')
    other = torch.randn(n, 150, generator=g, device='cuda')
    pairs = [(big[:, 0:16], other[:, 0:64]),          # last-layer shape [16, N] x [N, 64]
             (big[:, 16:80], other[:, 64:128]),       # first-layer shape [64, N] x [N, 64]
             (big[:, 80:128], other[:, 128:148]),     # 48 x 20: partial tiles on both sides
             (big[:, 0:200], other[:, 0:4]),          # 200 rows of output: cut into 64-row blocks
             (other[:, 3:4], big[:, 5:135])]          # 1 x 130 from odd column offsets (unaligned views)
    outs = grad_weight(pairs)
    for (a, b), out in zip(pairs, outs):
        assert out.shape == (a.shape[1], b.shape[1])
        _close(out, a.double().t() @ b.double())
    again = grad_weight(pairs)                        # fixed reduction order: bitwise reproducible
    assert all(torch.equal(x, y) for x, y in zip(outs, again))
    # width-128 models: whole 128 x 128 blocks take their own kernel (csrc/dense_bwd.hip: gw_stage1_lds128), with and without the
    # alternative operand; a 256 x 128 job is two of them
    wide_a = torch.randn(n, 256, generator=g, device='cuda')
    wide_b = torch.randn(n, 132, generator=g, device='cuda')
    alt = torch.randn(n, 128, generator=g, device='cuda')
    scale = torch.rand(n, generator=g, device='cuda')
    mask = (torch.rand(n, generator=g, device='cuda') < 0.4).to(torch.uint8)
    got = grad_weight([(wide_a[:, :128], wide_b[:, :128]), (wide_a, wide_b[:, :128], mask, alt, scale)])
    _close(got[0], wide_a[:, :128].double().t() @ wide_b[:, :128].double())
    bb = torch.where(mask.bool().unsqueeze(1), alt.double() * scale.double().unsqueeze(1), wide_b[:, :128].double())
    _close(got[1], wide_a.double().t() @ bb)


def test_dense_batch_matches_float64():
    from graph_recsys_benchmark_amd.engine import dense_batch
    g = torch.Generator(device='cuda').manual_seed(3)
    n = 20011
    dT = torch.randn(n, 160, generator=g, device='cuda')
    out = torch.full((n, 9 * 64 + 4), float('nan'), device='cuda')
    ws = [torch.randn(16, 64, generator=g, device='cuda') for _ in range(9)]
    jobs = [(dT[:, 16 * p:16 * p + 16], ws[p], out[:, 64 * p:64 * p + 64]) for p in range(9)]
    wide = torch.randn(128, 36, generator=g, device='cuda')        # k = 128, 36 output columns (narrow-ish, two tiles)
    res = torch.empty(n, 36, device='cuda')
    jobs.append((dT[:, 32:160], wide, res))
    dense_batch(jobs)
    for p in range(9):
        _close(out[:, 64 * p:64 * p + 64], dT[:, 16 * p:16 * p + 16].double() @ ws[p].double())
    _close(res, dT[:, 32:160].double() @ wide.double())
    assert torch.isnan(out[:, 576:]).all()                         # nothing written outside the job's columns


def test_dense_half_rejects_bad_arguments():
    from graph_recsys_benchmark_amd import _lib
    from graph_recsys_benchmark_amd.engine import dense_batch, grad_weight
    a = torch.randn(64, 8, device='cuda')
    with pytest.raises(ValueError):
        grad_weight([(a, torch.randn(63, 8, device='cuda'))])      # row counts differ
    with pytest.raises(ValueError):
        grad_weight([(a.t(), a.t())])                              # column stride != 1
    with pytest.raises(ValueError):
        dense_batch([(a, torch.randn(4, 8, device='cuda'), torch.empty(64, 8, device='cuda'))])   # k mismatch
    with pytest.raises(_lib.PeaError):                             # k = 6 is not a multiple of 4: refused by the library
        dense_batch([(torch.randn(64, 6, device='cuda'), torch.randn(6, 8, device='cuda'), torch.empty(64, 8, device='cuda'))])
    assert grad_weight([]) == []


@pytest.mark.parametrize('emb,hid,out', [(64, 64, 16), (128, 128, 16), (64, 64, 8)])
@pytest.mark.parametrize('form', ['gat', 'gcn', 'sage'])
def test_fused_dense_backward_matches_float64(form, emb, hid, out):
    """csrc/mlp2_bwd.hip on its own: dZ = (dT_1 W_1 [+ dR_1 W_1root]) gated by H > 0, dA = dZ W_0 [, dXr = dZ W_0root] for three
    channels laid out side by side like the training workspace, all rows and a listed subset (rows outside the list untouched)."""
    from graph_recsys_benchmark_amd.engine import RowSet, mlp2_backward_data, mlp2_backward_data_sage
    g = torch.Generator(device='cuda').manual_seed(emb + out)
    n, p = 3001, 3
    dt1 = torch.randn(n, p * out + 4, generator=g, device='cuda')
    dr1 = torch.randn(n, p * out, generator=g, device='cuda')
    h = torch.randn(n, p * hid, generator=g, device='cuda')          # about half of the gates closed
    w0 = [torch.randn(hid, emb, generator=g, device='cuda') * 0.1 for _ in range(p)]
    w0r = [torch.randn(hid, emb, generator=g, device='cuda') * 0.1 for _ in range(p)]
    w1 = [torch.randn(out, hid, generator=g, device='cuda') * 0.1 for _ in range(p)]
    w1r = [torch.randn(out, hid, generator=g, device='cuda') * 0.1 for _ in range(p)]
    mark = torch.zeros(n, 4, device='cuda')
    mark[torch.randperm(n, generator=torch.Generator().manual_seed(1))[:700].cuda(), 2] = 1.0
    live = RowSet(n, torch.device('cuda')).fill_from(mark, 4)
    for rows in (None, live):
        dz = torch.full((n, p * hid), float('nan'), device='cuda')
        da = torch.full((n, p * emb), float('nan'), device='cuda')
        dx = torch.full((n, p * emb), float('nan'), device='cuda')
        if form == 'sage':
            chans = [(w0[c], w0r[c], w1[c], w1r[c], c * out, c * out, c * hid, c * hid, c * emb, c * emb) for c in range(p)]
            mlp2_backward_data_sage(chans, emb, hid, out, dt1, dr1, h, dz, da, dx, rows=rows)
        else:
            gcn = form == 'gcn'          # GCNConv.weight is [in, out]
            chans = [((w0[c].t().contiguous() if gcn else w0[c]), (w1[c].t().contiguous() if gcn else w1[c]),
                      c * out, c * hid, c * hid, c * emb) for c in range(p)]
            mlp2_backward_data(chans, emb, hid, out, dt1, h, dz, da, rows=rows, weights_in_out=gcn)
        sel = torch.arange(n, device='cuda') if rows is None else torch.nonzero(mark[:, 2]).flatten()
        rest = torch.ones(n, dtype=torch.bool, device='cuda')
        rest[sel] = False
        for c in range(p):
            d = dt1[sel, c * out:(c + 1) * out].double() @ w1[c].double()
            if form == 'sage':
                d = d + dr1[sel, c * out:(c + 1) * out].double() @ w1r[c].double()
            z = d * (h[sel, c * hid:(c + 1) * hid] > 0).double()
            _close(dz[sel, c * hid:(c + 1) * hid], z)
            _close(da[sel, c * emb:(c + 1) * emb], z @ w0[c].double())
            if form == 'sage':
                _close(dx[sel, c * emb:(c + 1) * emb], z @ w0r[c].double())
        assert torch.isnan(dz[rest]).all() and torch.isnan(da[rest]).all()      # listed rows only


def test_row_sets_and_masked_weight_gradient():
    """csrc/rows.hip + pea_gw_job's alternative operand: the rows of a table that hold a non-zero (or are flagged), in ascending
    order with a device-side count; zeroing them again; a weight gradient over the list whose flagged rows read `scale * alt`."""
    from graph_recsys_benchmark_amd.engine import RowSet, grad_weight
    g = torch.Generator(device='cuda').manual_seed(9)
    n = 5003
    table = torch.zeros(n, 24, device='cuda')
    hot = torch.tensor([0, 7, 64, 4095, 5002], device='cuda')
    table[hot, torch.tensor([0, 23, 5, 11, 19], device='cuda')] = torch.tensor([1.0, -2.0, 1e-30, 3.0, -0.0], device='cuda')
    also = torch.zeros(n, dtype=torch.uint8, device='cuda')
    also[torch.tensor([7, 100], device='cuda')] = 1
    rs = RowSet(n, torch.device('cuda')).fill_from(table, 24, also=also)
    want = [0, 7, 64, 100, 4095]                       # -0.0 is a zero; row 100 comes from the flags
    assert int(rs.count.item()) == len(want) and rs.ids[:len(want)].tolist() == want
    assert rs.flags.nonzero().flatten().tolist() == want
    filled = torch.ones(n, 24, device='cuda')
    rs.zero_rows_of(filled, 16)
    assert float(filled[want, :16].abs().sum()) == 0.0 and float(filled.sum()) == n * 24 - len(want) * 16
    a = torch.randn(n, 64, generator=g, device='cuda')
    b = torch.randn(n, 64, generator=g, device='cuda')
    alt = torch.randn(n, 64, generator=g, device='cuda')
    scale = torch.rand(n, generator=g, device='cuda')
    mask = torch.zeros(n, dtype=torch.uint8, device='cuda')
    mask[torch.tensor([7, 4095, 33], device='cuda')] = 1
    got = grad_weight([(a, b, mask, alt, scale), (a[:, :16], b)], rows=rs)
    idx = torch.tensor(want, device='cuda')
    bb = torch.where(mask[idx].bool().unsqueeze(1), alt[idx].double() * scale[idx].double().unsqueeze(1), b[idx].double())
    _close(got[0], a[idx].double().t() @ bb)
    _close(got[1], a[idx, :16].double().t() @ b[idx].double())
