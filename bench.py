#!/usr/bin/env python3
"""Headline benchmark: BPR-scored edges/sec of the PEAGAT forward on a synthetic MovieLens-25m-shaped HIN
(emb_dim 64, 9 metapaths; BASELINE.json) on N MI355X GPUs of one node.

One "step" = what the reference executes per training batch up to the loss scalar (solvers.py:211-214 ->
models/base.py:43-48): the FULL-GRAPH forward (9 metapaths x 2 GAT layers over every node and edge, attentive
fusion) followed by BPR scoring of B = 4096 (u, i+, i-) triples.  Inputs (x, weights, CSR plan, batch) are
resident in HBM when the timed region starts.  value = messages reduced per step (sum over metapaths/steps of
E + N self loops) / step time, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--preset ml25m_shaped] [--kind gat]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (N > 1, one rank per GPU)

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch stream over the timed
region (pea_profile_*); `cpu_baseline` times the CPU oracle (oracle/pea_oracle.c, a port of the reference's
PyG op sequence) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from graph_recsys_benchmark_amd import _lib, models  # noqa: E402
from graph_recsys_benchmark_amd.utils import SyntheticHIN, update_pea_graph_input  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); 6290 GB/s measured float4 copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--preset', default='ml25m_shaped')
    ap.add_argument('--kind', default='gat', choices=['gat', 'gcn', 'sage'])
    ap.add_argument('--scale', type=float, default=1.0, help='edge/node count multiplier (tests only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip the HIP-event roofline leg')
    ap.add_argument('--graph', action='store_true', help='N=1 only: replay the forward from a captured hipGraph\n'
                    '(PEAEngine.forward_graphed; for launch-bound presets such as ml_small)')
    ap.add_argument('--train-steps', type=int, default=0, help='also time this many full training steps '
                    '(zero_grad, loss, backward, Adam step: reference solvers.py:213-216); extra field, N=1 only')
    ap.add_argument('--emulate-world', type=int, default=0, help='single process: time ONE rank (rank 0) of a sharded run of\n'
                    'this many ranks with the collectives skipped (results are wrong, timings are per-rank compute + host work)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' "
                    'only to rehearse several ranks on one GPU)')
    return ap.parse_args()


def build_model(dataset, kind, device):
    base = {'gat': models.PEAGATRecsysModel, 'gcn': models.PEAGCNRecsysModel, 'sage': models.PEASageRecsysModel}[kind]
    dataset_args = dataset.dataset_args()
    train_args = {'device': device, 'num_metapaths': dataset.spec['num_metapaths']}

    class PEARecsysModel(base):
        def update_graph_input(self, ds):
            return update_pea_graph_input(dataset_args, train_args, ds_obj)

    ds_obj = dataset
    torch.manual_seed(2020)                      # = 2019 + run 1 (solvers.py:123)
    model = PEARecsysModel(**dataset.model_args(kind=kind))
    with torch.no_grad():                        # biases start at 0 in the reference; give them values so the
        for name, p in model.named_parameters():  # bias path is exercised with realistic magnitudes
            if name.endswith('bias'):
                p.uniform_(-0.05, 0.05)
    return model.to(device)


def read_profile():
    lib = _lib.load()
    cap = 1 << 16
    names = C.create_string_buffer(cap * 32)
    ms = (C.c_float * cap)()
    units = (C.c_double * cap)()
    cnt = C.c_int()
    lib.pea_profile_read(cap, names, ms, units, C.byref(cnt))
    out = {}
    for i in range(cnt.value):
        nm = names.raw[i * 32:(i + 1) * 32].split(b'\0')[0].decode()
        rec = out.setdefault(nm, [0, 0.0, 0.0])
        rec[0] += 1
        rec[1] += ms[i]
        rec[2] += units[i]
    return out


def cpu_baseline(dataset, model, kind):
    """The CPU oracle (oracle/pea_oracle.c: the reference's PyG op sequence in C, OpenMP on the loops torch runs in
    parallel) on the SAME workload: every metapath channel of the same graph, then the fusion."""
    from oracle import oracle as orc
    from graph_recsys_benchmark_amd.utils import metapath_table
    table = metapath_table(dataset.dataset_args())[:dataset.spec['num_metapaths']]
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    n = dataset.num_nodes
    cores = min(len(os.sched_getaffinity(0)), 64)
    orc.set_num_threads(cores)
    cache, outs, msgs = {}, [], 0
    t0 = time.perf_counter()
    for p, steps in enumerate(table):
        edges = []
        for rel, flipped in steps:
            if (rel, flipped) not in cache:
                e = dataset.edge_index_nps[rel].astype(np.int64)
                cache[(rel, flipped)] = np.ascontiguousarray(e[::-1]) if flipped else e
            edges.append(cache[(rel, flipped)])
        lps = [{k.split('gnn_layers.%d.' % s)[1]: v for k, v in sd.items()
                if k.startswith('pea_channels.%d.gnn_layers.%d.' % (p, s))} for s in range(len(edges))]
        outs.append(orc.channel_forward(kind, sd['x'], edges, lps, [1] * len(edges)))
        msgs += sum(e.shape[1] + (n if kind != 'sage' else 0) for e in edges)
    fused = orc.fuse(np.stack(outs, axis=1), sd.get('att'))
    dt = time.perf_counter() - t0
    return dict(value=msgs / dt, unit='edges/s', cores=cores, kind='port',
                sample='one full step on the CPU: all %d metapath channels x 2 layers + fusion, %d messages, %.1f s '
                       '(edge-list int64 copies included)' % (len(table), msgs, dt)), fused


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)
    _lib.require_device()

    dataset = SyntheticHIN(args.preset, seed=2019, scale=args.scale)
    model = build_model(dataset, args.kind, device)
    model.train()
    batch = torch.from_numpy(dataset.bpr_batch()).to(device)
    if world > 1:
        model.shard(rank, world)     # destination rows tile-interleaved over the ranks; exchanges over RCCL
    elif args.emulate_world > 1:
        model.shard(0, args.emulate_world)
        model._get_engine().plan.layout.dry = True

    def step():
        with torch.no_grad():
            if args.graph and world == 1 and args.emulate_world <= 1:
                eng = model._get_engine()
                model.cached_repr = eng.forward_graphed(model._layer_params(), model.x.detach(), att=getattr(model, 'att', None))
                from graph_recsys_benchmark_amd.engine import bpr_score
                return bpr_score(model.cached_repr, batch, model.fc1.weight, model.fc1.bias, model.fc2.weight, model.fc2.bias)
            return model.loss(batch)

    for _ in range(args.warmup):
        loss = step()
    # clock settle: the part ramps its clocks over the first few hundred ms of load, so keep stepping (untimed) for
    # about 0.3 s; the timed region below then sees the steady state the steps after it would see.  Every rank must
    # run the SAME number of steps (each step holds collectives): the count comes from a max-reduced probe.
    torch.cuda.synchronize()
    t_probe = time.perf_counter()
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    per_step = torch.tensor([(time.perf_counter() - t_probe) / 3], dtype=torch.float64,
                            device=device if (world > 1 and args.backend == 'nccl') else 'cpu')
    if world > 1:
        dist.all_reduce(per_step, op=dist.ReduceOp.MAX)
    n_settle = int(min(2000, max(1, 0.3 / max(float(per_step.item()), 1e-6))))
    for _ in range(n_settle):
        loss = step()
    torch.cuda.synchronize()
    lib = _lib.load()
    profile = not args.no_profile

    def timed_region(with_events):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        lib.pea_profile_enable(1 if with_events else 0)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        lib.pea_profile_enable(0)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    # the timed region proper (K steps, nothing but the work), then the same K steps again with a pair of HIP events
    # around every kernel launch (on the launch stream) for the per-kernel roofline
    dt, loss = timed_region(False)
    prof, dt_prof = {}, None
    comm_ms = None
    if profile:
        from graph_recsys_benchmark_amd.sharding import CommTimer
        CommTimer.enabled, CommTimer.events = world > 1, []
        dt_prof, loss = timed_region(True)
        prof = read_profile()
        if world > 1:                      # device time between the start and the end of every collective, this rank
            comm_ms = CommTimer.total_ms() / args.steps
        CommTimer.enabled = False

    eng = model._engine
    messages = eng.messages
    ms_per_step = dt / args.steps * 1e3
    value = messages / (dt / args.steps)
    b = batch.shape[0]
    alg_bytes = eng.algorithmic_bytes + b * (12 + 12 * dataset.spec['repr_dim'] + 4)

    out = {
        'metric': 'BPR-scored edges/sec, PEAGAT MovieLens-25m, emb_dim=64, 9 metapaths',
        'value': value, 'unit': 'edges/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s PEA%s forward+BPR: N=%d nodes, %d metapaths x 2 steps, %d messages/step, '
                               'emb %d / hidden %d / repr %d, batch %d triples'
                               % (args.preset, args.kind.upper(), dataset.num_nodes, dataset.spec['num_metapaths'],
                                  messages, dataset.spec['emb_dim'], dataset.spec['hidden_size'],
                                  dataset.spec['repr_dim'], b),
                   'parallelism': 'rows%d' % world if world > 1 else 'single', 'loss': float(loss)},
        'bpr_triples_per_s': b / (dt / args.steps),
        'ms_per_step_with_kernel_events': None if dt_prof is None else dt_prof / args.steps * 1e3,
        'exchange_ms_per_step': comm_ms,   # N > 1: stream time inside the collectives (rank 0), from the profiled pass
        'forward_roofline': {'algorithmic_bytes_per_step': alg_bytes,
                             'achieved_GBs': alg_bytes / (dt / args.steps) / 1e9,
                             'frac_of_8TBs': alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS / world},
    }
    if prof:
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        name, (launches, ms, units) = dom
        achieved = units / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic = load_traffic(name)
        out['roofline'] = {'bound': 'hbm', 'kernel': name, 'launches': launches,
                           'avg_launch_ms': ms / launches, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                           'algorithmic_bytes_per_launch': units / launches,
                           # the committed PMC traffic of this kernel over its live duration: what the fabric really moved
                           # (algorithmic bytes count the reference's per-channel re-reads, so `frac` can exceed 1)
                           'traffic_GBs': None if not traffic or world > 1 else traffic / (ms / launches * 1e-3) / 1e9,
                           'traffic_frac': None if not traffic or world > 1 else traffic / (ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out['kernels_ms_per_step'] = {k: round(v[1] / args.steps, 4) for k, v in
                                      sorted(prof.items(), key=lambda kv: -kv[1][1])}
    if world == 1 and args.train_steps > 0:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)

        def train_step():
            opt.zero_grad()
            l = model.loss(batch)
            l.backward()
            opt.step()
            return l

        for _ in range(2):
            train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            l = train_step()
        torch.cuda.synchronize()
        tt = (time.perf_counter() - t0) / args.train_steps
        out['training_step'] = {'ms_per_step': tt * 1e3, 'steps': args.train_steps, 'loss': float(l),
                                'what': 'zero_grad + full-graph forward + BPR loss + backward + Adam step'}
        if profile:
            lib.pea_profile_enable(1)
            train_step()
            torch.cuda.synchronize()
            lib.pea_profile_enable(0)
            out['training_step']['hip_kernels_ms'] = {k: round(v[1], 3) for k, v in
                                                      sorted(read_profile().items(), key=lambda kv: -kv[1][1])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, want = cpu_baseline(dataset, model, args.kind)
        out['cpu_baseline'] = base
        with torch.no_grad():                                 # same weights as the oracle run (Adam may have stepped above)
            got = model.forward().cpu().numpy()               # full-size parity of the fused table against the oracle
        out['parity_vs_cpu_oracle'] = {'max_abs_err': float(np.abs(got - want).max()),
                                       'max_abs_value': float(np.abs(want).max()), 'rows': int(want.shape[0])}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def load_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), or None."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get('hbm_bytes_per_launch')
    except Exception:
        return None


if __name__ == '__main__':
    main()
