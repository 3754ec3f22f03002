"""GPU, BASELINE.json's full size (the MovieLens-25m-shaped preset of bench.py: 273,744 nodes, user2item 24.8 M edges):
properties of the path that do not need an oracle run at that size -- CSR structure against the COO it came from,
degree checksums against an independent torch scatter, convex-combination / linearity identities of the three conv
kinds, order invariance, determinism, and the sharded step against the single-GPU one.  (bench.py additionally compares
the fused table with the CPU oracle at this size on every run.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hin():
    from graph_recsys_benchmark_amd.utils.synthetic import SyntheticHIN
    ds = SyntheticHIN('ml25m_shaped', seed=2019)
    u2i = torch.from_numpy(ds.edge_index_nps['user2item'].astype(np.int64)).cuda()
    return ds, u2i


def test_csr_is_the_coo_sorted_by_destination_stably(hin):
    from graph_recsys_benchmark_amd.engine import GraphPlan
    ds, u2i = hin
    n = ds.num_nodes
    e = u2i.shape[1]
    order = torch.sort(u2i[1], stable=True).indices
    for row_bytes in (0, 256):                                # 0: never source-sliced; 256: the GAT layer-1 hint
        plan = GraphPlan(n, [[u2i]], False, gather_row_bytes=row_bytes)   # SAGE-style: edge list as given
        rowptr, col = plan.export_csr(0)
        assert int(rowptr[0]) == 0 and int(rowptr[-1]) == e
        deg = (rowptr[1:] - rowptr[:-1]).long()
        assert bool((deg >= 0).all())
        assert torch.equal(deg, torch.bincount(u2i[1], minlength=n))
        info = plan.relation_info(0)
        assert info['edges'] == e and info['max_degree'] == int(deg.max())
        if row_bytes == 0:
            # stable: inside a destination row the sources keep their COO order
            assert info['slices'] == 1 and torch.equal(col.long(), u2i[0][order])
        else:
            # big relation, 42 MB of gathered rows: edges inside a row are grouped by source slice (placement only) --
            # every row still holds exactly its COO edges, and the order inside a slice is still the COO order
            assert info['slices'] > 1
            dst = torch.repeat_interleave(torch.arange(n, device='cuda'), deg)
            got = torch.sort(dst * n + col.long()).values
            want = torch.sort(u2i[1] * n + u2i[0]).values
            assert torch.equal(got, want)


def test_degree_checksums_and_convex_combinations(hin):
    from graph_recsys_benchmark_amd.nn import GATConv, GCNConv, SAGEConv
    ds, u2i = hin
    n, f = ds.num_nodes, 64
    deg_in = torch.bincount(u2i[1], minlength=n).float()
    ones = torch.ones(n, f, device='cuda')
    with torch.no_grad():
        # SAGE: mean of ones over the in-neighbours = 1 where a row has any, else 0; root term switched off
        sage = SAGEConv(f, f).cuda()
        sage.lin_rel.weight.copy_(torch.eye(f)); sage.lin_rel.bias.zero_(); sage.lin_root.weight.zero_()
        out = sage(ones, u2i)
        torch.testing.assert_close(out, (deg_in > 0).float()[:, None].expand(n, f), rtol=0, atol=2e-7)   # sum * (1/count)
        # GCN with W = I on ones: out_i = sum_j norm_ij  (self loop included) -- checked against torch's own scatter
        gcn = GCNConv(f, f).cuda()
        gcn.weight.copy_(torch.eye(f)); gcn.bias.zero_()
        out = gcn(ones, u2i)
        keep = u2i[0] != u2i[1]
        row = torch.cat([u2i[0][keep], torch.arange(n, device='cuda')])
        colv = torch.cat([u2i[1][keep], torch.arange(n, device='cuda')])
        deg = torch.zeros(n, device='cuda', dtype=torch.float64).index_add_(0, row, torch.ones(row.numel(), device='cuda', dtype=torch.float64))
        dinv = deg.pow(-0.5)
        dinv[torch.isinf(dinv)] = 0
        want = torch.zeros(n, device='cuda', dtype=torch.float64).index_add_(0, colv, dinv[row] * dinv[colv])
        torch.testing.assert_close(out[:, 0].double(), want, rtol=2e-5, atol=1e-6)
        assert torch.equal(out, out[:, :1].expand(n, f))        # every column saw the same arithmetic
        # GAT: attention weights are a convex combination -- identical source rows come back unchanged (+ bias), whatever
        # the graph and the attention vectors
        gat = GATConv(f, f, heads=1).cuda()
        gat.bias.uniform_(-0.1, 0.1)
        const = torch.randn(1, f, device='cuda').expand(n, f).contiguous()
        out = gat(const, u2i)
        want = (const[:1] @ gat.lin.weight.t() + gat.bias).expand(n, f)
        torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-6)


def test_linearity_order_invariance_and_determinism(hin):
    from graph_recsys_benchmark_amd.nn import GCNConv, SAGEConv
    ds, u2i = hin
    n, f = ds.num_nodes, 64
    g = torch.Generator(device='cuda').manual_seed(5)
    x, y = torch.randn(n, f, generator=g, device='cuda'), torch.randn(n, f, generator=g, device='cuda')
    perm = torch.randperm(u2i.shape[1], generator=g, device='cuda')
    shuffled = u2i[:, perm].contiguous()
    with torch.no_grad():
        for conv in (GCNConv(f, 16).cuda(), SAGEConv(f, 16).cuda()):
            for p in conv.parameters():
                if p.dim() == 1:
                    p.zero_()                                    # no bias: the map x -> conv(x) is linear
            a = conv(x, u2i)
            assert torch.equal(a, conv(x, u2i))                  # no atomics: bitwise reproducible
            lin = conv(2.0 * x - 0.5 * y, u2i)
            ref = 2.0 * a.double() - 0.5 * conv(y, u2i).double()
            scale = float(ref.abs().max())
            assert float((lin.double() - ref).abs().max()) <= 2e-5 * scale
            # the same multiset of edges in another COO order: same sums up to fp32 summation order
            b = conv(x, shuffled)
            assert float((a.double() - b.double()).abs().max()) <= 2e-5 * float(a.abs().max())


def _sharded_worker(rank, world, port):
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import bench
        from graph_recsys_benchmark_amd.utils.synthetic import SyntheticHIN
        torch.cuda.set_device(0)
        ds = SyntheticHIN('ml25m_shaped', seed=2019)
        model = bench.build_model(ds, 'gat', torch.device('cuda', 0))
        batch = torch.from_numpy(ds.bpr_batch()).cuda()
        model.train()
        with torch.no_grad():
            ref_loss = model.loss(batch)
            ref = model.cached_repr.clone()
            model.shard(rank, world)
            loss = model.loss(batch)                       # batch rows only are exchanged
            model.eval()                                   # full table: second-layer sources + fused rows all-gathered
            assert torch.equal(model.cached_repr, ref), 'rank %d: fused table differs' % rank
        assert torch.equal(loss, ref_loss), 'rank %d: %r vs %r' % (rank, loss, ref_loss)
    finally:
        dist.destroy_process_group()


def test_sharded_step_equals_the_single_gpu_step():
    """Two ranks (sharing the box's one GPU, exchanges over gloo) at full size: every fused row and the BPR loss are
    bit-identical to the single-GPU ones -- each row is produced by exactly one rank with the same reduction order."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port), nprocs=2, join=True)


@pytest.mark.parametrize('kind', ['gat', 'sage'])
def test_full_size_training_gradients_vs_float64_on_the_2hop_neighbourhood(kind):
    """loss.backward() at BASELINE's full size (reference solvers.py:213-215): every parameter gradient of the loss of a
    sub-batch against float64 autograd on the sub-batch's complete 2-hop in-neighbourhood (oracle/grad64.py: exact -- the
    loss reads nothing else).  Bound: 2e-4 of the tensor's largest gradient, as in tests/test_gpu_backward.py; rows of x
    outside the neighbourhood carry exactly zero."""
    import bench
    from graph_recsys_benchmark_amd.utils.synthetic import SyntheticHIN
    ds = SyntheticHIN('ml25m_shaped', seed=2019)
    model = bench.build_model(ds, kind, torch.device('cuda', 0))
    model.train()
    batch = torch.from_numpy(ds.bpr_batch()).cuda()[:24]
    res = bench.gradient_check(ds, model, batch, kind)
    assert res['tensors'] == len(list(model.named_parameters()))
    assert res['loss_rel_err'] <= 2e-5, res
    assert res['worst_rel_err'] <= 2e-4, res
    assert res['x_grad_rel_err'] <= 2e-4, res
    assert res['x_grad_nonzero_outside_neighbourhood'] == 0, res
    assert 0 < res['x_rows_in_neighbourhood'] < ds.num_nodes
