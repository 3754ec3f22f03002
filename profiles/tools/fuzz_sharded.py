"""Randomised check of the row-sharded forward (GPU box, WORLD ranks sharing the one GPU, exchanges over gloo): on random
graphs / widths / heads / step counts the sharded fused table, stack and training-mode loss must equal the single-GPU
ones BIT FOR BIT.  python profiles/tools/fuzz_sharded.py [WORLD] [N] [seed]"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def worker(rank, world, port, count, seed, strict=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import build_model, random_state_dict
    torch.cuda.set_device(0)
    rng = np.random.default_rng(seed)                 # same stream on every rank
    bad = 0
    for i in range(count):
        kind = ['gat', 'gcn', 'sage'][rng.integers(0, 3)]
        heads = int(rng.choice([1, 1, 2])) if kind == 'gat' else 1
        n = int(rng.integers(40, 4000))
        emb, hidden, repr_dim = 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 5))
        twostep = os.environ.get('FUZZ_TWOSTEP') == '1'      # only what the two-step inference schedule takes (csrc/mlp2.hip)
        if twostep:
            heads, emb, hidden = 1, int(rng.choice([64, 128])), int(rng.choice([64, 128]))
        tile = int(rng.choice([16, 64, 256]))
        rels = []
        for _ in range(int(rng.integers(1, 4))):
            e = int(rng.choice([0, 30, 800, 20000]))
            dst = rng.integers(0, n, e)
            if rng.random() < 0.6 and e:
                dst = np.where(rng.random(e) < 0.8, rng.integers(0, max(2, n // 20), e), dst)
            rels.append(np.stack([rng.integers(0, n, e), dst]).astype(np.int64))
        steps, edges = [], []
        for _ in range(int(rng.integers(1, 5))):
            s = 2 if twostep else int(rng.integers(1, 4))
            if kind == 'gat' and heads > 1 and s == 1:
                s = 2
            steps.append(s)
            edges.append([rels[rng.integers(0, len(rels))] if rng.random() < 0.7 else
                          np.ascontiguousarray(rels[rng.integers(0, len(rels))][::-1]) for _ in range(s)])
        aggr = 'att' if rng.random() < 0.7 else 'mean'
        st = int(rng.integers(0, 1000))
        b = int(rng.integers(1, 300))
        batch = torch.from_numpy(np.stack([rng.integers(0, n, b), rng.integers(0, n, b), rng.integers(0, n, b)],
                                          axis=1).astype(np.int64)).cuda()
        desc = '%d: %s heads %d n %d emb %d hid %d repr %d steps %s aggr %s tile %d edges %s' % (
            i, kind, heads, n, emb, hidden, repr_dim, steps, aggr, tile, [r.shape[1] for r in rels])
        model = build_model(kind, n, edges, steps, emb, hidden, repr_dim, heads=heads, channel_aggr=aggr)
        model.load_state_dict(random_state_dict(model, st))
        with torch.no_grad():
            model.eval()
            ref, ref_stack = model.forward(return_stack=True)
            model.train()
            ref_loss = model.loss(batch)
            model.shard(rank, world, tile=tile)
            loss = model.loss(batch)
            model.eval()
            got, got_stack = model.forward(return_stack=True)
        ok = torch.equal(got, ref) and torch.equal(got_stack, ref_stack) and torch.equal(loss, ref_loss)
        flag = torch.tensor([0 if ok else 1])
        dist.all_reduce(flag)
        if not ok:
            d_f, d_s = (got - ref).abs(), (got_stack - ref_stack).abs()
            rows_bad = (d_s.amax(dim=(1, 2)) > 0).nonzero().flatten()
            chans_bad = (d_s.amax(dim=(0, 2)) > 0).nonzero().flatten().tolist()
            print('   rank %d: fused max diff %.3e (scale %.3e), stack max diff %.3e, %d rows differ (first %s), channels %s, loss %r vs %r'
                  % (rank, float(d_f.max()), float(ref.abs().max()), float(d_s.max()), rows_bad.numel(), rows_bad[:5].tolist(),
                     chans_bad, float(loss), float(ref_loss)), flush=True)
        if int(flag) and rank == 0:
            bad += 1
            print('FAIL', desc, flush=True)
        elif rank == 0 and i % 5 == 0:
            print('ok  ', desc, flush=True)
        del model
    if rank == 0:
        print('%d / %d failed (world %d)' % (bad, count, world), flush=True)
    total = torch.tensor([bad])
    dist.all_reduce(total)
    dist.destroy_process_group()
    if strict and int(total):
        raise AssertionError('%d sharded case(s) differ from the single-GPU result' % int(total))


if __name__ == '__main__':
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port, count, seed), nprocs=world, join=True)
