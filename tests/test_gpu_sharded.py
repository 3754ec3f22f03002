"""GPU: the row-sharded multi-rank forward (world 2 and 3 ranks sharing the one GPU of the test box, exchanges over
gloo) equals the single-rank forward BIT FOR BIT: every output row is produced by exactly one rank with the same
reduction order (config 3).  On the 8-GPU node the same code runs with backend 'nccl' (RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, kind, heads, hidden=32, repr_dim=16):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from helpers import build_model, random_hin, random_state_dict
        torch.cuda.set_device(0)
        n, blocks, rel = random_hin(21, n_user=3000, n_item=900, n_attr=50, e_u2i=40000, e_attr=2500)
        u2i, a2i = rel['u2i'], rel['a2i']
        flip = lambda e: np.ascontiguousarray(e[::-1])
        edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)], [flip(a2i), a2i, flip(u2i)]]
        steps = [2, 2, 2, 3]
        if heads == 1:                      # a 1-step channel only stacks with num_heads = 1 (models/base.py:196)
            edges.append([a2i])
            steps.append(1)
        model = build_model(kind, n, edges, steps, 32, hidden, repr_dim, heads=heads)
        model.load_state_dict(random_state_dict(model, 9))
        model.eval()
        with torch.no_grad():
            ref, ref_stack = model.forward(return_stack=True)
            model.shard(rank, world, tile=64)
            got, got_stack = model.forward(return_stack=True)
            masked = model.forward(metapath_idx=2)
            model.shard(0, 1)
            ref_masked = model.forward(metapath_idx=2)
        assert torch.equal(got, ref), 'rank %d: fused rows differ (max %g)' % (rank, (got - ref).abs().max())
        # training-mode loss of a sharded model: batch rows only are exchanged; same value as the single-rank loss
        rng = np.random.default_rng(4)
        (u0, u1), (i0, i1) = blocks['u'], blocks['i']
        batch = torch.from_numpy(np.stack([rng.integers(u0, u1, 256), rng.integers(i0, i1, 256),
                                           rng.integers(i0, i1, 256)], axis=1).astype(np.int64)).cuda()
        model.train()
        with torch.no_grad():
            ref_loss = model.loss(batch)
            model.shard(rank, world, tile=64)
            got_loss = model.loss(batch)
            assert model._repr_partial
            pred = model.predict(batch[:, 0], batch[:, 1])            # completes the cached table lazily
            assert torch.equal(model.cached_repr, ref) and not model._repr_partial and pred.shape == (256, 1)
            model.shard(0, 1)
        model.eval()
        assert torch.equal(got_loss, ref_loss), 'rank %d: loss %r vs %r' % (rank, got_loss, ref_loss)
        assert torch.equal(got_stack, ref_stack)
        assert torch.equal(masked, ref_masked)
        info = model._engine.plan.relation_info(0)
        assert info['rows_owned'] == n       # back to a single-rank plan
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,heads,world', [('gat', 1, 2), ('gat', 2, 3), ('gcn', 1, 2), ('sage', 1, 3)])
def test_sharded_forward_is_bit_exact(kind, heads, world):
    mp.spawn(_worker, args=(world, _free_port(), kind, heads), nprocs=world, join=True)


@pytest.mark.parametrize('kind,hidden,repr_dim', [('gcn', 8, 8), ('gat', 8, 16), ('sage', 4, 12)])
def test_sharded_narrow_layers_are_bit_exact(kind, hidden, repr_dim):
    """Layers of <= 16 output columns: the per-relation first-layer jobs of a sharded plan and the single merged job of
    the one-GPU plan must run the same kernel family (the narrow-output kernel sums k in another order) -- found by
    profiles/tools/fuzz_sharded.py."""
    mp.spawn(_worker, args=(2, _free_port(), kind, 1, hidden, repr_dim), nprocs=2, join=True)


def _twostep_worker(rank, world, port, kind, emb, hidden):
    """Configurations the two-step inference schedule takes (every channel 2 steps, emb / hidden 64 or 128: csrc/mlp2.hip
    on the rank's own rows, first-layer aggregation of the replicated x): sharded == single GPU, bit for bit."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from helpers import build_model, random_hin, random_state_dict
        torch.cuda.set_device(0)
        n, blocks, rel = random_hin(23, n_user=2500, n_item=800, n_attr=40, e_u2i=30000, e_attr=2000)
        u2i, a2i = rel['u2i'], rel['a2i']
        flip = lambda e: np.ascontiguousarray(e[::-1])
        edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)], [a2i, flip(u2i)], [flip(a2i), a2i]]
        model = build_model(kind, n, edges, [2] * len(edges), emb, hidden, 16)
        model.load_state_dict(random_state_dict(model, 11, scale=0.2))
        model.eval()
        from test_gpu_edge_cases import _kernel_names_of_one_forward
        assert 'mlp2_fused' in _kernel_names_of_one_forward(model)      # this configuration does take the two-step schedule
        rng = np.random.default_rng(6)
        (u0, u1), (i0, i1) = blocks['u'], blocks['i']
        batch = torch.from_numpy(np.stack([rng.integers(u0, u1, 200), rng.integers(i0, i1, 200),
                                           rng.integers(i0, i1, 200)], axis=1).astype(np.int64)).cuda()
        with torch.no_grad():
            ref, ref_stack = model.forward(return_stack=True)
            model.train()
            ref_loss = model.loss(batch)
            model.eval()
            model.shard(rank, world, tile=64)
            got, got_stack = model.forward(return_stack=True)
            model.train()
            got_loss = model.loss(batch)
            model.eval()
        assert 'mlp2_fused' in _kernel_names_of_one_forward(model)      # ... and still does on a rank of a sharded plan
        assert torch.equal(got, ref), 'rank %d: fused rows differ (max %g)' % (rank, (got - ref).abs().max())
        assert torch.equal(got_stack, ref_stack)
        assert torch.equal(got_loss, ref_loss)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,emb,hidden,world', [('gat', 64, 64, 2), ('gcn', 64, 128, 3), ('sage', 64, 64, 2),
                                                   ('sage', 128, 64, 3), ('gat', 128, 128, 5)])
def test_sharded_two_step_schedule_is_bit_exact(kind, emb, hidden, world):
    mp.spawn(_twostep_worker, args=(world, _free_port(), kind, emb, hidden), nprocs=world, join=True)


def _train_worker(rank, world, port, kind, heads):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from helpers import build_model, random_hin, random_state_dict
        torch.cuda.set_device(0)
        n, blocks, rel = random_hin(23, n_user=2500, n_item=700, n_attr=40, e_u2i=30000, e_attr=2000)
        u2i, a2i = rel['u2i'], rel['a2i']
        flip = lambda e: np.ascontiguousarray(e[::-1])
        edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)], [flip(a2i), a2i, flip(u2i)]]
        steps = [2, 2, 2, 3]
        model = build_model(kind, n, edges, steps, 32, 32, 16, heads=heads)
        model.load_state_dict(random_state_dict(model, 10, scale=0.25))
        rng = np.random.default_rng(6)
        (u0, u1), (i0, i1) = blocks['u'], blocks['i']
        batch = torch.from_numpy(np.stack([rng.integers(u0, u1, 384), rng.integers(i0, i1, 384),
                                           rng.integers(i0, i1, 384)], axis=1).astype(np.int64)).cuda()
        model.train()
        model.zero_grad()
        ref_loss = model.loss(batch)                     # single-rank training step: the oracle of the sharded one
        ref_loss.backward()
        ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        model.shard(rank, world, tile=64)
        model.zero_grad()
        loss = model.loss(batch)
        assert model._repr_partial
        loss.backward()
        # same forward rows, same replicated loss arithmetic: the loss value is bit-identical
        assert torch.equal(loss.detach(), ref_loss.detach()), 'rank %d: loss %r vs %r' % (rank, loss, ref_loss)
        gscale = max(float(w.abs().max()) for w in ref.values())
        for k, p in model.named_parameters():
            assert p.grad is not None, k
            w = ref[k]
            scale = max(float(w.abs().max()), 1e-12)
            err = float((p.grad - w).abs().max())
            # the ranks' shares are summed in another order than the single-rank reduction: fp32 bound, not bit equality
            # (a gradient that is zero up to rounding -- d att_i of a layer whose logits sit on one leaky-relu branch --
            # is compared against the size of the model's gradients, not against itself)
            assert err <= 2e-5 * scale + 1e-7 * gscale, 'rank %d %s: max err %.3e vs scale %.3e' % (rank, k, err, scale)
        # every rank holds the SAME gradients bit for bit (all-reduced / all-gathered), so the replicas stay in lockstep
        sums = [None] * world
        dist.all_gather_object(sums, [float(p.grad.double().sum()) for p in model.parameters()])
        assert all(v == sums[0] for v in sums)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
        opt.step()
        with torch.no_grad():
            assert np.isfinite(float(model.loss(batch)))
        model.eval()                                     # full table after the step: exchanges + all-gather
        assert not model._repr_partial and bool(torch.isfinite(model.cached_repr).all())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,heads,world', [('gat', 1, 2), ('gat', 2, 3), ('gcn', 1, 2), ('sage', 1, 2)])
def test_sharded_training_step_matches_single_rank(kind, heads, world):
    """model.shard() + loss.backward() (reference step solvers.py:213-216): loss bit-identical, every parameter gradient
    within the fp32 bound of the single-rank gradients, identical on all ranks."""
    mp.spawn(_train_worker, args=(world, _free_port(), kind, heads), nprocs=world, join=True)


def _rccl_worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    try:
        from graph_recsys_benchmark_amd.sharding import ShardLayout
        shard = ShardLayout(1000, 0, 1)
        buf = torch.arange(64 * 8, dtype=torch.float32, device='cuda').view(64, 8)
        want = buf.clone()
        shard.world = 1
        # drive the RCCL branch directly (world 1: the only multi-rank-free way to touch it on a 1-GPU box)
        mine = buf[0:64]
        dist.all_gather_into_tensor(buf.view(-1), mine.reshape(-1))    # IN PLACE (send block = own slice), as sharding.py does
        t = torch.ones(4, device='cuda')
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert torch.equal(buf, want) and float(t.sum()) == 4.0
        assert dist.get_backend() == 'nccl'
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank_smoke():
    """backend 'nccl' is RCCL on ROCm: the collective calls the sharded forward issues work on this box."""
    mp.spawn(_rccl_worker, args=(1, _free_port()), nprocs=1, join=True)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment: the script starts its own two ranks before
    anything touches a GPU and rank 0 prints the one JSON line (here over gloo, both ranks on the box's one GPU)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--preset', 'ml_small',
                        '--steps', '3', '--warmup', '1', '--train-steps', '2'], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['parallelism'] == 'rows2'
    assert out['exchange_ms_per_step'] is not None and np.isfinite(out['training_step']['loss'])


def _replay_worker(rank, world, port, mode):
    """engine.PEAEngine.sharded_loss replays its launch sequences (tape / hipGraph / eager): several steps with DIFFERENT
    batches and weights updated IN PLACE in between must each equal the single-GPU loss bit for bit -- a replay that kept
    a stale pointer, batch or weight would not."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      PEA_SHARD_REPLAY=mode)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from helpers import build_model, random_hin, random_state_dict
        torch.cuda.set_device(0)
        n, blocks, rel = random_hin(29, n_user=2200, n_item=700, n_attr=40, e_u2i=26000, e_attr=1800)
        u2i, a2i = rel['u2i'], rel['a2i']
        flip = lambda e: np.ascontiguousarray(e[::-1])
        edges = [[u2i, flip(u2i)], [flip(u2i), u2i], [a2i, flip(u2i)], [flip(a2i), a2i]]
        sd = None
        losses = {}
        for sharded in (False, True):
            model = build_model('gat', n, edges, [2] * 4, 64, 64, 16)
            if sd is None:
                sd = random_state_dict(model, 13, scale=0.2)
            model.load_state_dict(sd)
            model.train()
            if sharded:
                model.shard(rank, world, tile=64)
            rng = np.random.default_rng(8)
            (u0, u1), (i0, i1) = blocks['u'], blocks['i']
            out = []
            with torch.no_grad():
                for step in range(4):
                    b = 160 if step < 3 else 96          # the last step changes the batch size (static buffers are rebuilt)
                    batch = torch.from_numpy(np.stack([rng.integers(u0, u1, b), rng.integers(i0, i1, b),
                                                       rng.integers(i0, i1, b)], axis=1).astype(np.int64)).cuda()
                    out.append(model.loss(batch).clone())
                    for p in model.parameters():         # an optimizer's in-place update
                        p.mul_(1.0 + 0.01 * (step + 1))
                if sharded:
                    eng = model._get_engine()
                    recs = eng._sl['graphs']
                    assert (recs is None) == (mode == 'eager')
                    if mode == 'tape':       # (a part may be empty: here every owned row is read by some other rank)
                        assert sum(len(t) for t in recs.values()) >= 5
            losses[sharded] = torch.stack(out).cpu()
        assert torch.equal(losses[True], losses[False]), 'rank %d (%s): %r vs %r' % (rank, mode, losses[True], losses[False])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['tape', 'graph', 'eager'])
def test_sharded_loss_replay_modes_track_batches_and_weights(mode):
    mp.spawn(_replay_worker, args=(2, _free_port(), mode), nprocs=2, join=True)
