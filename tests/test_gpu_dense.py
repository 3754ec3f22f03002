"""Dense half of the backward (csrc/dense_bwd.hip, csrc/gemm.hip) on shapes the model tests do not reach: deep-k input
gradients (K > 128: gemm_deep_kernel with 1 / 2 / 4 column tiles) and 64 x 64 weight-gradient blocks (gw_stage1_lds),
against float64 matmuls."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,k,n_out', [(1000, 160, 32), (777, 192, 64), (1500, 320, 128), (513, 136, 100), (64, 576, 64),
                                       (3, 132, 4)])
def test_dense_batch_deep_k_matches_float64(n, k, n_out):
    from graph_recsys_benchmark_amd import engine
    g = torch.Generator().manual_seed(n + k + n_out)
    a = torch.randn(n, k, generator=g).cuda()
    w = (torch.randn(k, n_out, generator=g) * 0.2).cuda()
    out = torch.full((n, n_out), 9.0, device='cuda')
    engine.dense_batch([(a, w, out)])
    want = a.double() @ w.double()
    err = float((out.double() - want).abs().max())
    assert err <= 2e-6 * float(want.abs().max()) * (k ** 0.5), err
    # a row subset: only the listed rows are written
    rows = torch.arange(0, n, 3, dtype=torch.int32, device='cuda')
    out2 = torch.full((n, n_out), 9.0, device='cuda')
    engine.dense_batch([(a, w, out2)], rows=rows)
    keep = torch.ones(n, dtype=torch.bool, device='cuda')
    keep[rows.long()] = False
    assert torch.equal(out2[rows.long()], out[rows.long()])
    assert bool((out2[keep] == 9.0).all())


@pytest.mark.parametrize('n,ma,nb', [(5000, 64, 64), (4097, 128, 64), (300, 576, 64), (2000, 64, 100), (10, 64, 64)])
def test_grad_weight_blocks_match_float64(n, ma, nb):
    from graph_recsys_benchmark_amd import engine
    g = torch.Generator().manual_seed(n + ma + nb)
    a = torch.randn(n, ma, generator=g).cuda()
    b = torch.randn(n, nb, generator=g).cuda()
    got = engine.grad_weight([(a, b)])[0]
    want = a.double().t() @ b.double()
    err = float((got.double() - want).abs().max())
    assert err <= 3e-6 * float(want.abs().max()) * max(1.0, (n / 1000.0) ** 0.5) + 1e-4, err
    # views with a row stride (column blocks of a wider buffer), as the backward passes them
    wide_a = torch.randn(n, ma + 64, generator=g).cuda()
    wide_b = torch.randn(n, nb + 32, generator=g).cuda()
    got2 = engine.grad_weight([(wide_a[:, 64:], wide_b[:, :nb])])[0]
    want2 = wide_a[:, 64:].double().t() @ wide_b[:, :nb].double()
    assert float((got2.double() - want2).abs().max()) <= 3e-6 * float(want2.abs().max()) * max(1.0, (n / 1000.0) ** 0.5) + 1e-4


@pytest.mark.parametrize('n,k,n_out,listed', [(1000, 16, 64, False), (777, 32, 64, True), (515, 64, 128, False), (300, 128, 32, True),
                                              (33, 16, 20, False)])
def test_dense_batch_gate_applies_the_relu_mask(n, k, n_out, listed):
    """out = gate > 0 ? a w : 0 in the product's epilogue, bit-identical to the ungated product followed by the mask."""
    from graph_recsys_benchmark_amd import engine
    g = torch.Generator().manual_seed(7 * n + k + n_out)
    a = torch.randn(n, k, generator=g).cuda()
    w = (torch.randn(k, n_out, generator=g) * 0.3).cuda()
    wide = torch.randn(n, n_out + 24, generator=g).cuda()      # the gate is a column block of a wider buffer
    gate = wide[:, 8:8 + n_out]
    rows = torch.arange(1, n, 2, dtype=torch.int32, device='cuda') if listed else None
    plain = torch.full((n, n_out), 5.0, device='cuda')
    gated = torch.full((n, n_out), 5.0, device='cuda')
    engine.dense_batch([(a, w, plain)], rows=rows)
    engine.dense_batch([(a, w, gated, gate)], rows=rows)
    want = torch.where(gate > 0, plain, torch.zeros_like(plain))
    if listed:
        sel = rows.long()
        assert torch.equal(gated[sel], want[sel])
        keep = torch.ones(n, dtype=torch.bool, device='cuda')
        keep[sel] = False
        assert bool((gated[keep] == 5.0).all())
    else:
        assert torch.equal(gated, want)
