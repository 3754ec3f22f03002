#!/usr/bin/env python3
"""Headline benchmark: BPR-scored edges/sec of the PEAGAT forward on a synthetic MovieLens-25m-shaped HIN
(emb_dim 64, 9 metapaths; BASELINE.json) on N MI355X GPUs of one node.

One "step" = what the reference executes per training batch up to the loss scalar (solvers.py:211-214 ->
models/base.py:43-48): the FULL-GRAPH forward (9 metapaths x 2 GAT layers over every node and edge, attentive
fusion) followed by BPR scoring of B = 4096 (u, i+, i-) triples.  Inputs (x, weights, CSR plan, batch) are
resident in HBM when the timed region starts.  value = messages reduced per step (sum over metapaths/steps of
E + N self loops) / step time, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--preset ml25m_shaped] [--kind gat]

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (one process per GPU,
127.0.0.1 rendezvous) BEFORE anything touches a GPU; under an external launcher (torch.distributed.run) it is one
of the ranks.  Rank 0 prints ONE JSON line.

`roofline` (dominant kernel, measured live with HIP events on the launch stream, pea_profile_*):
    achieved = bytes the launch pulls through the memory system (row chunk + source index of every message, once)
               / its average duration
    peak     = the gather ceiling of the cache tier those rows are served from (MI355X_MICROARCH.md, "Indexed rows":
               L2 18.8 TB/s, Infinity Cache 8.6 TB/s, HBM 6.1 TB/s), tier named from the measured L2 hit rate of THIS
               preset / kind (profiles/traffic.json) or, without one, from the table's footprint
    traffic  = fabric-side bytes per launch from the rocprofv3 PMC passes of this preset / kind (or null)
    algorithmic = the SURVEY.md 8(d) yardstick (what the reference's separate conv calls would move), for reference
`hbm_floor`: compulsory HBM bytes of the step / 8 TB/s against the measured step time.
`cpu_baseline`: the CPU oracle (oracle/pea_oracle.c, a port of the reference's PyG op sequence) on the same step.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); 6290 GB/s measured float4 copy
# measured gather ceilings by where the rows are served from (MI355X_MICROARCH.md, section "Indexed rows: gather into
# LDS": 16.8-18.8 TB/s L2-resident, 8.6 TB/s from a 38 MB table in the Infinity Cache, 6.0-6.1 TB/s from HBM)
GATHER_CEILING_GBS = {'l2': 18800.0, 'infinity_cache': 8600.0, 'hbm': 6100.0}
L2_BYTES, MALL_BYTES = 32 << 20, 256 << 20   # aggregate L2 (8 x 4 MiB), Infinity Cache
FP32_MFMA_PEAK_TF = 157.3                    # fp32-input MFMA = the fp32 vector rate (MI355X_MICROARCH.md, Matrix cores)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--preset', default='ml25m_shaped')
    ap.add_argument('--kind', default='gat', choices=['gat', 'gcn', 'sage'])
    ap.add_argument('--scale', type=float, default=1.0, help='edge/node count multiplier (tests only)')
    ap.add_argument('--hbm-scale', type=float, default=1.0,
                    help='scale of the stress_10m (BASELINE configs[4]) leg of the default run: 1.0 = the whole configuration')
    ap.add_argument('--metapaths', type=int, default=0, help='use the first n metapaths of the dataset table (0 = the preset\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-samples', type=int, default=3, help='timed runs of the CPU oracle (median reported)')
    ap.add_argument('--no-profile', action='store_true', help='skip the HIP-event roofline leg')
    ap.add_argument('--no-extras', action='store_true', help='skip the eval-variant and 13-metapath legs')
    ap.add_argument('--graph', action='store_true', help='N=1 only: replay the forward from a captured hipGraph\n'
                    '(PEAEngine.forward_graphed; for launch-bound presets such as ml_small)')
    ap.add_argument('--train-steps', type=int, default=-1, help='also time this many full training steps '
                    '(zero_grad, loss, backward, Adam step: reference solvers.py:213-216) -> `training_step`; default: 6 on '
                    'one GPU, 0 on a sharded run (give a count to time the sharded training step too)')
    ap.add_argument('--grad-check-triples', type=int, default=48, help='training leg: full-size gradient check of a sub-batch '
                    'of this many triples against float64 autograd on its 2-hop neighbourhood (oracle/grad64.py); 0 = skip')
    ap.add_argument('--emulate-world', type=int, default=0, help='single process: time ONE rank (rank 0) of a sharded run of\n'
                    'this many ranks with the collectives skipped (results are wrong, timings are per-rank compute + host work)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' "
                    'only to rehearse several ranks on one GPU)')
    return ap.parse_args(argv)


def launch_ranks(cmds_envs, poll_s=0.05, grace_s=5.0):
    """Start one child per (argv, env) pair and supervise them TOGETHER: every child is polled (not waited on in rank
    order), and as soon as one exits non-zero the others are terminated (then killed after `grace_s`) and its code is
    returned -- a rank that dies early must not leave its peers sitting in a collective until the NCCL watchdog or the
    driver's own limit fires.  Returns 0 only if every child exited 0."""
    procs = [subprocess.Popen(cmd, env=env) for cmd, env in cmds_envs]
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = abs(bad[0]) or 1
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:            # exact PIDs of the children started above
            p.terminate()
        t_end = time.time() + grace_s
        for p in live:
            try:
                p.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def self_launch(args):
    """--gpus N > 1 without an external launcher: start N fresh ranks of this script (nothing in this process has
    touched a GPU: only argparse and imports ran) and relay rank 0's JSON line.  Returns the exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    jobs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        jobs.append(([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env))
    return launch_ranks(jobs)


HEADLINE_METRIC = 'BPR-scored edges/sec, PEAGAT MovieLens-25m, emb_dim=64, 9 metapaths'   # BASELINE.json


def is_headline(args):
    return args.preset == 'ml25m_shaped' and args.kind == 'gat' and args.scale == 1.0 and not args.metapaths


def metric_name(args):
    """BASELINE.json's metric string for the configuration it is quoted on; any other --preset / --kind / --scale /
    --metapaths run names its own workload, so a profile file of another preset cannot be mistaken for the headline."""
    if is_headline(args):
        return HEADLINE_METRIC
    return 'BPR-scored edges/sec, PEA%s %s%s%s (not the BASELINE.json headline configuration)' % (
        args.kind.upper(), args.preset, '' if args.scale == 1.0 else ' x %g' % args.scale,
        ', first %d metapaths' % args.metapaths if args.metapaths else '')


def build_model(dataset, kind, device):
    from graph_recsys_benchmark_amd import models
    from graph_recsys_benchmark_amd.utils import update_pea_graph_input
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
    """{kernel name: [launches, ms, algorithmic bytes, gathered bytes, largest gather table bytes]} since the last read."""
    import ctypes as C
    from graph_recsys_benchmark_amd import _lib
    lib = _lib.load()
    cap = 1 << 16
    names = C.create_string_buffer(cap * 32)
    ms = (C.c_float * cap)()
    units, pulled, table = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
    cnt = C.c_int()
    lib.pea_profile_read_ex(cap, names, ms, units, pulled, table, C.byref(cnt))
    out = {}
    for i in range(cnt.value):
        nm = names.raw[i * 32:(i + 1) * 32].split(b'\0')[0].decode()
        rec = out.setdefault(nm, [0, 0.0, 0.0, 0.0, 0.0])
        rec[0] += 1
        rec[1] += ms[i]
        rec[2] += units[i]
        rec[3] += pulled[i]
        rec[4] = max(rec[4], table[i])
    return out


def oracle_channels(dataset, model, kind, channels):
    """Per-channel [N, R] outputs of the CPU oracle for the listed channel indices (+ messages reduced)."""
    from oracle import oracle as orc
    from graph_recsys_benchmark_amd.utils import metapath_table
    table = metapath_table(dataset.dataset_args())
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    n = dataset.num_nodes
    cache, outs, msgs = {}, [], 0
    for p in channels:
        edges = []
        for rel, flipped in table[p]:
            if (rel, flipped) not in cache:
                e = dataset.edge_index_nps[rel].astype(np.int64)
                cache[(rel, flipped)] = np.ascontiguousarray(e[::-1]) if flipped else e
            edges.append(cache[(rel, flipped)])
        lps = [{k.split('gnn_layers.%d.' % s)[1]: v for k, v in sd.items()
                if k.startswith('pea_channels.%d.gnn_layers.%d.' % (p, s))} for s in range(len(edges))]
        outs.append(orc.channel_forward(kind, sd['x'], edges, lps, [1] * len(edges)))
        msgs += sum(e.shape[1] + (n if kind != 'sage' else 0) for e in edges)
    return outs, msgs, sd


def cpu_baseline(dataset, model, kind, samples):
    """The CPU oracle (oracle/pea_oracle.c: the reference's PyG op sequence in C, OpenMP on the loops torch runs in
    parallel) on the SAME workload: every metapath channel of the same graph, then the fusion.  Median of `samples`
    timed runs after none (the first run pages the inputs in and is part of the sample: the CPU path has no warm-up
    to speak of at 12 s per step)."""
    from oracle import oracle as orc
    cores = min(len(os.sched_getaffinity(0)), 64)
    orc.set_num_threads(cores)
    p_all = list(range(dataset.spec['num_metapaths']))
    times, fused, msgs = [], None, 0
    for _ in range(max(1, samples)):
        t0 = time.perf_counter()
        outs, msgs, sd = oracle_channels(dataset, model, kind, p_all)
        fused = orc.fuse(np.stack(outs, axis=1), sd.get('att'))
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return dict(value=msgs / dt, unit='edges/s', cores=cores, kind='port', samples=len(times),
                seconds=[round(t, 2) for t in times],
                sample='one full step on the CPU per sample: all %d metapath channels x 2 layers + fusion, %d messages, '
                       'median of %d runs = %.1f s (edge-list int64 copies included; BPR scoring of the batch, '
                       'microseconds, excluded)' % (len(p_all), msgs, len(times), dt)), fused


def load_traffic(preset, kind, scale, world, leg=None):
    """{kernel: {hbm_bytes_per_launch, l2_hit_rate, ...}} measured with rocprofv3 PMC passes for THIS preset / kind / scale
    on one GPU (profiles/traffic.json, written by profiles/summarize.py; key '<preset>/<kind>' at full scale,
    '<preset>@<scale>/<kind>' otherwise), or {}.  A rank of a sharded run gathers from the SAME source table through the
    same source slices (rows are sharded, sources are not), so it takes the L2 hit rate of the one-GPU profile to name
    the cache tier -- marked as such -- and no byte counts (those are per whole-graph launch)."""
    key = '%s/%s' % (preset, kind) if scale == 1.0 else '%s@%g/%s' % (preset, scale, kind)
    if leg:
        key += '/' + leg            # e.g. 'ml25m_shaped/gat/train': the PMC passes of the training step
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            tab = json.load(f).get(key, {})
    except Exception:
        return {}
    if world == 1:
        return tab
    return {k: {'l2_hit_rate': v.get('l2_hit_rate'), 'hit_from': 'the one-GPU profile of this preset'}
            for k, v in tab.items() if isinstance(v, dict) and v.get('l2_hit_rate') is not None}


def gather_tier(table_bytes, l2_hit_rate, achieved=None, hit_from=None):
    """(tier name, ceiling in GB/s, how it was decided) of a row gather.  With a measured L2 hit rate h of the kernel:
    h >= 0.5 -> the XCD L2s serve most rows: the guide's L2-resident ceiling; h < 0.5 -> mixed: the harmonic blend of the
    L2 ceiling (share h) and the ceiling of where the misses go (Infinity Cache while the table fits its 256 MiB, else
    HBM).  Without a PMC profile for this preset / kind: from the table's footprint alone -- unless the measured rate is
    above that tier's ceiling, which proves the rows come from a faster tier (the plans slice the sources so that a
    slice's rows stay in L2): then the next tier up is the bound."""
    below = 'infinity_cache' if table_bytes <= MALL_BYTES else 'hbm'
    if l2_hit_rate is not None:
        src = 'measured L2 hit rate %.2f' % l2_hit_rate + (' (%s)' % hit_from if hit_from else '')
        if l2_hit_rate >= 0.5:
            return 'l2', GATHER_CEILING_GBS['l2'], src
        blend = 1.0 / (l2_hit_rate / GATHER_CEILING_GBS['l2'] + (1.0 - l2_hit_rate) / GATHER_CEILING_GBS[below])
        why = '%s, table %.0f MB: harmonic blend of the L2 and %s gather ceilings' % (src, table_bytes / 1e6, below)
        if achieved is not None and achieved > blend:
            # the guide's miss-tier figure (measured with its own row width) is beaten by these rows: price the misses at
            # the hardware rate of that tier instead (HBM: the 8 TB/s pin rate; Infinity Cache: the L2 ceiling)
            miss = HBM_PEAK_GBS if below == 'hbm' else GATHER_CEILING_GBS['l2']
            blend = 1.0 / (l2_hit_rate / GATHER_CEILING_GBS['l2'] + (1.0 - l2_hit_rate) / miss)
            why += ('; measured %.0f GB/s is above that blend, so the misses are priced at %s instead'
                    % (achieved, 'the 8 TB/s HBM peak' if below == 'hbm' else 'the L2 ceiling'))
        return 'l2+' + below, blend, why
    if table_bytes <= L2_BYTES // 8:
        return 'l2', GATHER_CEILING_GBS['l2'], 'table %.1f MB fits one XCD L2 (no PMC profile for this preset / kind)' % (table_bytes / 1e6)
    order = ['hbm', 'infinity_cache', 'l2']
    tier, why = below, 'table %.0f MB vs the 256 MiB Infinity Cache (no PMC profile for this preset / kind)' % (table_bytes / 1e6)
    while achieved is not None and achieved > GATHER_CEILING_GBS[tier] and tier != 'l2':
        nxt = order[order.index(tier) + 1]
        why += '; measured %.0f GB/s is above the %s gather ceiling, so the sliced sources are served from %s' % (
            achieved, tier, nxt)
        tier = nxt
    return tier, GATHER_CEILING_GBS[tier], why


def kernel_roofline(name, rec, traffic_tab, flops):
    """roofline object of one kernel from its live HIP-event record (+ the committed PMC figures of this workload)."""
    launches, ms, units, pulled, table = rec
    avg_s = ms / launches * 1e-3
    meas = traffic_tab.get(name, {})
    hit, traffic = meas.get('l2_hit_rate'), meas.get('hbm_bytes_per_launch')
    out = {'kernel': name, 'launches': launches, 'avg_launch_ms': ms / launches}
    if name in flops:                                        # dense transform: fp32 MFMA (exact fp32 products)
        tf = flops[name] / (ms * 1e-3) / 1e12
        out.update(bound='mfma', achieved=tf, peak=FP32_MFMA_PEAK_TF, unit='TFLOP/s', frac=tf / FP32_MFMA_PEAK_TF,
                   definition='2*rows*K*n_out flops of the launches / their time; peak = fp32-input MFMA '
                              '(v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s, MI355X_MICROARCH.md)')
    elif pulled > 0:                                         # neighbour aggregation: a row gather
        achieved = pulled / launches / avg_s / 1e9
        tier, peak, why = gather_tier(table, hit, achieved, meas.get('hit_from'))
        out.update(bound='hbm', achieved=achieved, peak=peak, unit='GB/s', frac=achieved / peak, tier=tier, tier_from=why,
                   gather_table_bytes=table, gathered_bytes_per_launch=pulled / launches,
                   definition='achieved = (4*W row bytes + 4 index bytes) x messages of the launch / avg launch time; peak '
                              '= measured row-gather ceiling of the named cache tier (MI355X_MICROARCH.md, Indexed rows)')
    else:                                                    # streaming kernels (fusion, scoring, packing)
        achieved = units / launches / avg_s / 1e9
        out.update(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                   definition='algorithmic bytes of the launch / avg launch time against the 8 TB/s HBM peak')
    out.update(traffic=traffic,                              # fabric-side bytes per launch (rocprofv3 PMC) or null
               traffic_GBs=None if not traffic else traffic / avg_s / 1e9,
               traffic_frac_of_8TBs=None if not traffic else traffic / avg_s / 1e9 / HBM_PEAK_GBS,
               l2_hit_rate=hit, algorithmic_bytes_per_launch=units / launches,
               algorithmic_GBs=units / launches / avg_s / 1e9)
    return out


def transform_flops(sp, n, steps, kind='gat'):
    """Useful flops of the dense transforms of a 2-step model over `steps` steps, by kernel name: the level-wise
    schedule's two launches (PEA_FUSED2=0) or the two-step schedule's one (csrc/mlp2.hip: both layers chained; the
    padding of the second product to 32 output rows is not counted).  SAGE: every layer is lin_rel + lin_root (twice the
    products), all of them inside the one launch of the two-step schedule."""
    l0 = 2.0 * n * sp['emb_dim'] * sp['hidden_size'] * sp['num_metapaths'] * steps
    l1 = 2.0 * n * sp['hidden_size'] * sp['repr_dim'] * sp['num_metapaths'] * steps
    if kind == 'sage':
        return {'mlp2_fused': 2.0 * (l0 + l1)}
    return {'gemm_mfma_shared': l0, 'gemm_mfma_narrow': l1, 'mlp2_fused': l0 + l1}


def single_gpu_schedule(world, args):
    return world == 1 and args.emulate_world <= 1


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    from graph_recsys_benchmark_amd import _lib
    from graph_recsys_benchmark_amd.utils import SyntheticHIN
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        import datetime
        # a rank that never arrives (or dies in a collective) ends the job after two minutes, not after the default 10-30
        limit = datetime.timedelta(seconds=int(os.environ.get('PEA_DIST_TIMEOUT_S', '120')))
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device, timeout=limit)
        else:
            dist.init_process_group(args.backend, timeout=limit)
    _lib.require_device()
    if world > 1:
        from graph_recsys_benchmark_amd.sharding import ShardLayout
        ShardLayout.verify_inplace_all_gather(device if args.backend == 'nccl' else 'cpu')

    dataset = SyntheticHIN(args.preset, seed=2019, scale=args.scale)
    if args.metapaths:
        dataset.spec = dict(dataset.spec, num_metapaths=args.metapaths)
    model = build_model(dataset, args.kind, device)
    model.train()
    batch_host = dataset.bpr_batch()
    batch = torch.from_numpy(batch_host).to(device)
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

    from graph_recsys_benchmark_amd import engine as _eng

    def timed_region(fn, steps, with_events=False):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        _eng.GRAPHS_ENABLED = not with_events     # HIP events around every launch need eager launches (not a graph replay)
        lib.pea_profile_enable(1 if with_events else 0)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        lib.pea_profile_enable(0)
        _eng.GRAPHS_ENABLED = True
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    # the timed region proper (K steps, nothing but the work), then the same K steps again with a pair of HIP events
    # around every kernel launch (on the launch stream) for the per-kernel roofline
    dt, loss = timed_region(step, args.steps)
    prof, dt_prof = {}, None
    comm_ms = None
    if profile:
        from graph_recsys_benchmark_amd.sharding import CommTimer
        CommTimer.enabled, CommTimer.events = world > 1, []
        dt_prof, loss = timed_region(step, args.steps, True)
        prof = read_profile()
        if world > 1:                      # device time between the start and the end of every collective, this rank
            comm_ms = CommTimer.total_ms() / args.steps
        CommTimer.enabled = False

    eng = model._engine
    messages = eng.messages
    step_s = dt / args.steps
    ms_per_step = step_s * 1e3
    value = messages / step_s
    b = batch.shape[0]
    alg_bytes = eng.algorithmic_bytes + b * (12 + 12 * dataset.spec['repr_dim'] + 4)
    floor_bytes = eng.compulsory_bytes
    floor_ms = floor_bytes / (HBM_PEAK_GBS * 1e9) * 1e3 / world

    out = {
        'metric': metric_name(args),
        'value': value, 'unit': 'edges/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s PEA%s forward+BPR: N=%d nodes, %d metapaths x 2 steps, %d messages/step, '
                               'emb %d / hidden %d / repr %d, batch %d triples'
                               % (args.preset, args.kind.upper(), dataset.num_nodes, dataset.spec['num_metapaths'],
                                  messages, dataset.spec['emb_dim'], dataset.spec['hidden_size'],
                                  dataset.spec['repr_dim'], b),
                   'headline': is_headline(args),
                   'parallelism': 'rows%d' % world if world > 1 else 'single', 'loss': float(loss),
                   'batch_h2d': 'excluded: the %d-byte batch is resident in HBM when the timed region starts (bench '
                                'contract); over PCIe it is a ~10 us copy' % batch_host.nbytes},
        'bpr_triples_per_s': b / step_s,
        'ms_per_step_with_kernel_events': None if dt_prof is None else dt_prof / args.steps * 1e3,
        'exchange_ms_per_step': comm_ms,   # N > 1: stream time inside the collectives (rank 0), from the profiled pass
        # whole step against DRAM: every buffer of the schedule written once + read once, indices once, at 8 TB/s
        'hbm_floor': {'compulsory_bytes_per_step': floor_bytes, 'floor_ms_at_8TBs': floor_ms,
                      'frac': floor_ms / ms_per_step,
                      'note': 'step time is dominated by gathers served from L2 / Infinity Cache (roofline below), not by DRAM'},
        # the SURVEY.md 8(d) yardstick: bytes the REFERENCE's op sequence would move (per conv call its own index read,
        # [N, F] writes ...).  The schedule reads indices once per relation and keeps tables cache resident, so this
        # rate is NOT DRAM traffic and may exceed 8 TB/s: reported for comparison with BASELINE.md only.
        'reference_yardstick': {'algorithmic_bytes_per_step': alg_bytes, 'GBs': alg_bytes / step_s / 1e9},
    }
    if prof:
        traffic_tab = load_traffic(args.preset, args.kind, args.scale, max(world, args.emulate_world or 1))
        flops = {}
        if single_gpu_schedule(world, args):     # the transform launch(es) of a 2-step model
            sp, n = dataset.spec, dataset.num_nodes
            flops = transform_flops(sp, n, args.steps, args.kind)
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        out['roofline'] = kernel_roofline(dom[0], dom[1], traffic_tab, flops)
        gathers = {k: v for k, v in prof.items() if v[3] > 0}
        if gathers:                                            # the dominant neighbour-aggregation kernel, when it is not
            g = max(gathers.items(), key=lambda kv: kv[1][1])   # the dominant kernel overall (stress preset: the transform)
            if g[0] != dom[0]:
                out['roofline_gather'] = kernel_roofline(g[0], g[1], traffic_tab, flops)
        out['kernels_ms_per_step'] = {k: round(v[1] / args.steps, 4) for k, v in
                                      sorted(prof.items(), key=lambda kv: -kv[1][1])}
    eff_world = max(world, args.emulate_world or 1)
    if eff_world > 1:
        out['exchange_model'] = exchange_model(model, eff_world, out.get('kernels_ms_per_step', {}), b, dataset.spec['repr_dim'])
        if args.emulate_world > 1:
            out['emulated'] = ('ONE rank (rank 0) of %d on one GPU, collectives skipped: ms_per_step = that rank\'s kernels + host '
                               'work; value is NOT a job throughput' % args.emulate_world)
    single = world == 1 and args.emulate_world <= 1
    train_steps = args.train_steps if args.train_steps >= 0 else (6 if single else 0)
    if train_steps > 0:          # sharded too: every rank steps its replica with the all-reduced gradients
        snapshot = {k: v.detach().clone() for k, v in model.state_dict().items()}
        try:
            out['training_step'] = training_leg(dataset, model, batch, args, timed_region, train_steps, world, profile,
                                                rank == 0 and single and not args.no_cpu_baseline)
        except torch.cuda.OutOfMemoryError as e:
            # the training buffers (per level: T, O and their gradients for every channel) of the largest presets do not fit
            # beside the inference schedule's (stress_10m at full size: 4 x 82 GB): reported, not fatal -- unless sharded,
            # where the other ranks are inside the step's collectives
            if world > 1:
                raise
            if getattr(model, '_train_engine', None) is not None:
                model._train_engine = None
            torch.cuda.empty_cache()
            out['training_step'] = {'skipped': 'training buffers do not fit this GPU beside the inference schedule: %s'
                                               % str(e).split('.')[0]}
        with torch.no_grad():    # the legs below (eval variant, CPU baseline + full-size parity) see the weights the timed
            for k, v in model.state_dict().items():   # forward steps above ran on, not the ones Adam has stepped
                v.copy_(snapshot[k])
        del snapshot
    if single and not args.no_extras:
        out['eval_variant'] = eval_variant(dataset, model, args, timed_region)
    if rank == 0 and single and not args.no_cpu_baseline:
        base, want = cpu_baseline(dataset, model, args.kind, args.cpu_samples)
        out['cpu_baseline'] = base
        with torch.no_grad():                                 # same weights as the oracle run (Adam may have stepped above)
            got = model.forward().cpu().numpy()               # full-size parity of the fused table against the oracle
        out['parity_vs_cpu_oracle'] = {'max_abs_err': float(np.abs(got - want).max()),
                                       'max_abs_value': float(np.abs(want).max()), 'rows': int(want.shape[0])}
    if rank == 0 and single and args.preset == 'stress_10m' and args.kind in ('gat', 'sage'):
        # the CPU oracle is out of reach at this size (pass --no-cpu-baseline): sampled rows in float64 instead
        out['parity_vs_float64_sampled_rows'] = sampled_rows_check(dataset, model, args.kind, device)
    if single and not args.no_extras and args.preset == 'ml25m_shaped' and not args.metapaths and args.scale == 1.0:
        del model
        torch.cuda.empty_cache()
        out['metapaths_13'] = thirteen_metapaths(dataset, args, device, batch, timed_region, not args.no_cpu_baseline)
    if single and not args.no_extras and args.preset == 'ml25m_shaped' and not args.metapaths and args.scale == 1.0:
        # BASELINE configs[4] at FULL size (10 M nodes, 200 M edges, 16 metapaths, emb 128: 240 of the 288 GB of HBM, ~100 s of
        # the run, most of it generating the graph on the host); should the card not have that much free, at 30 %
        try:
            out['hbm_resident'] = hbm_resident_leg(args, device, timed_region, not args.no_cpu_baseline, scale=args.hbm_scale)
        except (torch.cuda.OutOfMemoryError, _lib.PeaError) as e:     # (the plan's CSR arrays come from hipMalloc: PeaError)
            if not isinstance(e, torch.cuda.OutOfMemoryError) and 'memory' not in str(e).lower():
                raise
            torch.cuda.empty_cache()
            out['hbm_resident'] = hbm_resident_leg(args, device, timed_region, not args.no_cpu_baseline, scale=0.3)
            out['hbm_resident']['fallback'] = 'scale %g did not fit the free HBM' % args.hbm_scale
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


# xGMI model of the exchanges (NOT measured: no multi-GPU lease was available to this build).  MI355X: 7 links per GPU, 153.6 GB/s
# per link both directions together = 76.8 GB/s per direction; in an 8-GPU all-gather every rank receives one block from each
# peer over that peer's own link, so the receive side peaks at (world - 1) x 76.8 GB/s.  RCCL is assumed to reach HALF of that
# at these message sizes (3.5 MB per rank and peer) and to cost 15 us per collective on top.
XGMI_GBS_PER_DIRECTION, RCCL_EFFICIENCY, COLLECTIVE_LATENCY_MS = 76.8, 0.5, 0.015


def exchange_model(model, world, kernels_ms, batch_rows, repr_dim):
    """Bytes a rank receives per step and the modelled time of the collectives of the sharded loss step: the all-gathers of
    the next level's gather sources (issued asynchronously after the rows other ranks read; the rest of the stage runs
    behind them: its measured kernel time is the overlap window) and the all-reduce of the batch's fused rows."""
    eng = model._get_engine()
    recv = 0.0
    n_coll = 0
    for row in eng._exchanges:
        for d, _src, _dst, lay in row:
            recv += (world - 1) * lay.slots_per_rank * d.width * 4.0
            n_coll += 1
    bw = (world - 1) * XGMI_GBS_PER_DIRECTION * RCCL_EFFICIENCY * 1e9
    gather_ms = recv / bw * 1e3 + COLLECTIVE_LATENCY_MS
    # the rest-of-stage transform launch is the second (larger) of the two `mlp2_fused` launches of a step
    window_ms = None
    reduce_bytes = 3 * batch_rows * repr_dim * 4.0
    reduce_ms = 2.0 * reduce_bytes * (world - 1) / world / bw * 1e3 + COLLECTIVE_LATENCY_MS
    return {'source_allgather_bytes_received_per_rank': recv, 'source_allgathers_per_step': n_coll,
            'source_allgather_ms': gather_ms, 'loss_allreduce_bytes': reduce_bytes, 'loss_allreduce_ms': reduce_ms,
            'exchange_model_ms': gather_ms + reduce_ms,
            'overlap': 'the all-gathers run behind the transform of the rows no other rank reads (PEA_PART_REST: about 3/4 of '
                       'a rank\'s rows on this graph; its share of mlp2_fused is the window, see kernels_ms_per_step)',
            'assumptions': 'NOT measured. xGMI 76.8 GB/s per link and direction, %d peers, RCCL at %.0f %% of the link rate, '
                           '%.0f us per collective' % (world - 1, RCCL_EFFICIENCY * 100, COLLECTIVE_LATENCY_MS * 1e3)}


def training_leg(dataset, model, batch, args, timed_region, train_steps, world, profile, with_check):
    """The reference's training step (solvers.py:213-216): zero_grad, loss = model.loss(batch) on the full-graph forward,
    loss.backward() through the HIP conv stack, optimizer.step() -- timed like the headline step, then one more step with
    HIP events around every launch.  `roofline` is the dominant kernel of the step, built like the forward's (for a
    gradient gather: row chunk + index (+ side record) of every message once / its time, against the gather ceiling of the
    cache tier the rows come from; PMC figures from profiles/traffic.json key '<preset>/<kind>/train').  `gradient_check`:
    the same model's gradients of a sub-batch at FULL graph size against float64 autograd on the sub-batch's complete 2-hop
    in-neighbourhood (oracle/grad64.py: exact, the loss reads nothing else)."""
    lib = __import__('graph_recsys_benchmark_amd._lib', fromlist=['_lib']).load()
    # torch.optim.Adam as the reference builds it (solvers.py:141-146), in torch's single-kernel form (fused=True: the
    # default foreach form runs ~9 passes over the 70 MB embedding table and its moments, 0.36 ms of the step)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3, fused=True)
    res = {}

    def train_step():
        opt.zero_grad()
        l = model.loss(batch)
        l.backward()
        opt.step()
        return l

    for _ in range(2):
        train_step()
    tt, l = timed_region(train_step, train_steps)
    tt /= train_steps
    res.update({'ms_per_step': tt * 1e3, 'steps': train_steps, 'loss': float(l),
                'what': 'zero_grad + full-graph forward + BPR loss + backward + Adam step (torch.optim.Adam, fused=True)'
                        + (' (row-sharded over %d ranks: gradient-row fill-ins, gradient all-reduce, dx all-gather)' % world
                           if world > 1 else '')})
    eng = getattr(model, '_train_engine', None)
    layout = getattr(getattr(eng, 'plan', None), 'layout', None) if eng is not None else None
    if layout is not None and getattr(layout, 'dry', False):
        # emulated rank: the collectives were skipped; what each would have moved was noted (ShardLayout.dry_log) -> the same
        # xGMI model as the forward's exchange_model, nothing overlapped (the training step's exchanges are synchronous)
        layout.dry_log.clear()
        train_step()
        torch.cuda.synchronize()
        w = layout.world
        bw = (w - 1) * XGMI_GBS_PER_DIRECTION * RCCL_EFFICIENCY * 1e9
        gathers = [b for k, b in layout.dry_log if k == 'all_gather']
        reduces = [b for k, b in layout.dry_log if k == 'all_reduce']
        ms = sum(b / bw * 1e3 + COLLECTIVE_LATENCY_MS for b in gathers) + \
            sum(2.0 * b * (w - 1) / w / bw * 1e3 + COLLECTIVE_LATENCY_MS for b in reduces)
        res['exchange_model'] = {
            'all_gathers': len(gathers), 'all_gather_bytes_received_per_rank': float(sum(gathers)),
            'all_reduces': len(reduces), 'all_reduce_bytes': float(sum(reduces)), 'exchange_model_ms': ms,
            'projected_ms_per_step': res['ms_per_step'] + ms,
            'assumptions': 'NOT measured. xGMI %.1f GB/s per link and direction, %d peers, RCCL at %.0f %% of the link rate, %.0f us '
                           'per collective, no overlap with compute' % (XGMI_GBS_PER_DIRECTION, w - 1, RCCL_EFFICIENCY * 100,
                                                                       COLLECTIVE_LATENCY_MS * 1e3)}
    live = getattr(model._train_engine, '_live_rows', None) if getattr(model, '_train_engine', None) is not None else None
    if live is not None:     # rows the loss's gradient reaches beyond the last layer (csrc/rows.hip): the dense backward walks these
        res['gradient_support_rows'] = int(live.count.item())
        res['gradient_support_share'] = res['gradient_support_rows'] / float(dataset.num_nodes)
    if profile:
        n_prof = 3
        lib.pea_profile_enable(1)
        for _ in range(n_prof):
            train_step()
        torch.cuda.synchronize()
        lib.pea_profile_enable(0)
        prof = read_profile()
        res['hip_kernels_ms'] = {k: round(v[1] / n_prof, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        res['hip_kernels_ms_sum'] = round(sum(v[1] for v in prof.values()) / n_prof, 3)
        tab = load_traffic(args.preset, args.kind, args.scale, max(world, args.emulate_world or 1), leg='train')
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        res['roofline'] = kernel_roofline(dom[0], dom[1], tab, {})
        bwd = {k: v for k, v in prof.items() if k.startswith('gat_bwd') and v[3] > 0}
        if bwd:
            g = max(bwd.items(), key=lambda kv: kv[1][1])
            if g[0] != dom[0]:
                res['roofline_backward_gather'] = kernel_roofline(g[0], g[1], tab, {})
    # after the timed steps: the float64 side is ~20 s of multi-threaded CPU work, whose worker threads would still be
    # spinning next to the (host-bound) enqueue loop of the steps timed above
    if with_check and args.grad_check_triples > 0 and args.kind in ('gat', 'sage'):
        res['gradient_check'] = gradient_check(dataset, model, batch[:args.grad_check_triples], args.kind)
    return res


def gradient_check(dataset, model, sub_batch, kind):
    """Gradients of loss(sub_batch) at full graph size: HIP (model.loss(...).backward()) against float64 torch autograd on
    the sub-batch's 2-hop in-neighbourhood.  Per tensor: max |hip - f64| / max |f64| (the bound tests/test_gpu_backward.py
    uses is 2e-4); x: over all N rows (rows outside the neighbourhood must be exactly zero)."""
    from graph_recsys_benchmark_amd.utils import metapath_table
    from oracle.grad64 import f64_subgraph_loss_and_grads
    t0 = time.perf_counter()
    model.zero_grad()
    loss = model.loss(sub_batch)
    loss.backward()
    got = {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in model.named_parameters()}
    model.zero_grad()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    table = metapath_table(dataset.dataset_args())[:dataset.spec['num_metapaths']]
    cache, edges = {}, []
    for steps in table:
        row = []
        for rel, flipped in steps:
            if (rel, flipped) not in cache:
                e = dataset.edge_index_nps[rel].astype(np.int64)
                cache[(rel, flipped)] = np.ascontiguousarray(e[::-1]) if flipped else e
            row.append(cache[(rel, flipped)])
        edges.append(row)
    want_loss, want, touched = f64_subgraph_loss_and_grads(kind, sd, edges, sub_batch.cpu().numpy())
    # a tensor's error is measured against ITS largest float64 gradient plus 5e-3 of the largest gradient of any tensor, so
    # the tests' bound of 2e-4 reads  err <= 2e-4 * max|tensor| + 1e-6 * max|any|: gradients that vanish analytically are
    # sums of O(global) terms that cancel (a last-layer att_i: a per-row shift of all logits only matters through the
    # leaky-relu kink; SAGE layer-2 lin_rel of the channels ending at user rows: with the relu masks of the positive and the
    # negative score equal, d(pos - neg) / d repr_u is EXACTLY zero), and their fp32 residue is relative to those terms
    g_max = max(float(np.abs(w).max()) for w in want.values())
    worst, worst_name, per = 0.0, None, []
    for k, w in want.items():
        err, scale = float(np.abs(got[k] - w).max()), float(np.abs(w).max())
        rel_err = err / (scale + 5e-3 * g_max)
        per.append((rel_err, k, err, scale))
        if rel_err > worst:
            worst, worst_name = rel_err, k
    per.sort(reverse=True)
    outside = np.ones(want['x'].shape[0], bool)
    outside[touched] = False
    x_rel = float(np.abs(got['x'] - want['x']).max() / max(np.abs(want['x']).max(), 1e-30))
    return {'triples': int(sub_batch.shape[0]), 'tensors': len(want), 'loss_rel_err': abs(float(loss) - want_loss) / abs(want_loss),
            'worst_rel_err': worst, 'worst_tensor': worst_name, 'largest_gradient': g_max,
            'worst_three': [{'tensor': k, 'rel_err': r, 'abs_err': e, 'max_abs_f64': sc} for r, k, e, sc in per[:3]],
            'x_grad_rel_err': x_rel,
            'x_rows_in_neighbourhood': int(touched.size), 'x_grad_nonzero_outside_neighbourhood': int(np.count_nonzero(got['x'][outside])),
            'bound_in_tests': 2e-4, 'seconds': round(time.perf_counter() - t0, 1),
            'what': 'every parameter gradient of loss(sub-batch) on the FULL graph vs float64 autograd on the sub-batch\'s '
                    'complete 2-hop in-neighbourhood (oracle/grad64.py); rel err = max |diff| / (max |f64| of the tensor + 5e-3 x the largest gradient of any tensor)'}


def eval_variant(dataset, model, args, timed_region):
    """SURVEY.md 8(d) "eval variant": one eval() forward (full graph) + the scoring of U x (1 + 99) candidates and the
    rank / AUC / eval-loss reductions of solvers.py:56-96 for ALL users in one launch (pea_rank_eval); the timed
    region ends with the rank vector on the device.  Candidate ids are drawn on the host beforehand (the reference draws
    them with np.random.choice inside its loop; here uniform over the item block, vectorised)."""
    from graph_recsys_benchmark_amd import engine
    u_nids, cand = dataset.eval_candidates()
    dev = model.x.device
    u_t, c_t = torch.from_numpy(u_nids).to(dev), torch.from_numpy(cand).to(dev)

    def run():
        model.eval()
        return engine.rank_eval(model.cached_repr, u_t, c_t, model.fc1.weight, model.fc1.bias, model.fc2.weight, model.fc2.bias)

    run()
    steps = max(3, args.steps // 2)
    dt, (scores, rank, auc, loss) = timed_region(run, steps)
    dt /= steps
    # parity of a sample of users against the CPU oracle's scorer on the SAME fused table (the table itself is checked
    # at full size below): integer ranks equal unless two scores are closer than fp32 noise
    from oracle import oracle as orc
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith('fc')}
    table = model.cached_repr.cpu().numpy()
    pick = np.linspace(0, len(u_nids) - 1, 256).astype(np.int64)
    want = np.stack([orc.predict(table, np.full(cand.shape[1], u_nids[k]), cand[k], sd['fc1.weight'], sd['fc1.bias'],
                                 sd['fc2.weight'], sd['fc2.bias']) for k in pick])
    got = scores[torch.from_numpy(pick).to(dev)].cpu().numpy()
    want_rank = (want[:, 1:] > want[:, :1]).sum(axis=1)
    got_rank = rank[torch.from_numpy(pick).to(dev)].cpu().numpy()
    model.train()
    return {'ms': dt * 1e3, 'users': int(len(u_nids)), 'candidates_per_user': int(cand.shape[1]),
            'edges_per_s': model._engine.messages / dt, 'scores_per_s': cand.size / dt,
            'what': 'eval() forward over the full graph + pea_rank_eval of every user (rank, AUC, eval loss on device)',
            'parity_vs_cpu_oracle': {'users_checked': int(pick.size), 'score_max_abs_err': float(np.abs(got - want).max()),
                                     'ranks_differing': int((want_rank != got_rank).sum())}}


def hbm_resident_leg(args, device, timed_region, with_check, scale=1.0):
    """The same step on a workload whose gather tables do NOT fit the 256 MiB Infinity Cache: BASELINE config 5 (stress_10m:
    10 M nodes, 200 M edges, 16 metapaths, emb / hidden 128; item table of the 9-channel layer-2 gather: 2 M x 576 B) at
    `scale` (1.0: the whole configuration on ONE GPU; 0.3 was the default until round 3).  The default workload's tables
    (26-42 MB) are cache resident, this one shows the kernels against HBM.  PMC profiles: traffic.json keys 'stress_10m/gat'
    (profiles/r03/summary_r03s.json) and 'stress_10m@0.3/gat' (profiles/r02/summary_r02s.json).  Parity: sampled destination
    rows of two channels recomputed in float64 on their 2-hop in-neighbourhood (oracle/rows64.py)."""
    from graph_recsys_benchmark_amd import _lib
    from graph_recsys_benchmark_amd.utils import SyntheticHIN, metapath_table
    ds = SyntheticHIN('stress_10m', seed=2019, scale=scale)
    model = build_model(ds, args.kind, device)
    model.train()
    batch = torch.from_numpy(ds.bpr_batch()).to(device)

    def step():
        with torch.no_grad():
            return model.loss(batch)

    for _ in range(2):
        step()
    steps = 8 if scale < 0.6 else 4
    dt, loss = timed_region(step, steps)
    timed_region(step, steps, True)
    prof = read_profile()
    dt /= steps
    eng = model._engine
    sp, n = ds.spec, ds.num_nodes
    flops = {}
    if args.kind in ('gat', 'gcn'):
        flops = transform_flops(sp, n, steps)
    tab = load_traffic('stress_10m', args.kind, scale, 1)
    dom = max(prof.items(), key=lambda kv: kv[1][1])
    gathers = {k: v for k, v in prof.items() if v[3] > 0}
    g = max(gathers.items(), key=lambda kv: kv[1][1])
    out = {'workload': 'stress_10m x %g PEA%s forward+BPR: N=%d nodes, %d metapaths x 2 steps, %d messages/step, emb %d / hidden %d'
                       % (scale, args.kind.upper(), n, sp['num_metapaths'], eng.messages, sp['emb_dim'], sp['hidden_size']),
           'ms_per_step': dt * 1e3, 'edges_per_s': eng.messages / dt, 'loss': float(loss),
           'hbm_floor_frac': eng.compulsory_bytes / (HBM_PEAK_GBS * 1e9) / dt,
           'roofline': kernel_roofline(dom[0], dom[1], tab, flops),
           'roofline_gather': kernel_roofline(g[0], g[1], tab, flops),
           'kernels_ms_per_step': {k: round(v[1] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]}}
    if with_check and args.kind in ('gat', 'sage'):
        out['parity_vs_float64_sampled_rows'] = sampled_rows_check(ds, model, args.kind, device)
    del model
    torch.cuda.empty_cache()
    _lib.load().pea_profile_enable(0)
    return out


def sampled_rows_check(ds, model, kind, device):
    """stress preset (BASELINE config 5; SURVEY.md 8: "parity spot-checked on sampled destination rows recomputed on CPU"):
    sampled destination rows of two channels recomputed in float64 on their complete 2-hop in-neighbourhood (oracle/rows64.py)."""
    from oracle.rows64 import f64_rows_two_step
    from graph_recsys_benchmark_amd.utils import metapath_table
    table = metapath_table(ds.dataset_args())
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    n = ds.num_nodes
    rng = np.random.default_rng(5)
    u0 = ds.type_accs['uid']
    rows = np.unique(np.concatenate([rng.integers(u0, u0 + ds.num_uids, 12), rng.integers(ds.type_accs['attr_0'], n, 8)]))
    rows_t = torch.from_numpy(rows).to(device)
    with torch.no_grad():
        _, stack = model.forward(return_stack=True)
        picked = stack[rows_t].cpu().numpy()
    del stack
    err = scale_v = 0.0
    for p in (0, 2):            # [u2i, u2i^T] and [attr_0 -> item, u2i^T]: destination rows = users / attribute nodes
        rels = [np.ascontiguousarray(ds.edge_index_nps[r].astype(np.int64)[::-1]) if f else ds.edge_index_nps[r].astype(np.int64)
                for r, f in table[p]]
        truth = f64_rows_two_step(kind, sd, p, rels[0], rels[1], rows)
        err = max(err, float(np.abs(picked[:, p] - truth).max()))
        scale_v = max(scale_v, float(np.abs(truth).max()))
    return {'rows': int(rows.size), 'channels': [1, 3], 'max_abs_err': err, 'max_abs_value': scale_v}


def thirteen_metapaths(dataset, args, device, batch, timed_region, with_oracle):
    """The reference's own 25m configuration runs all THIRTEEN metapaths of utils/general_utils.py:335-356
    (experiments/scripts/script_movielens_25m.ps1:41); BASELINE.json's metric is quoted on the first nine.  Same step,
    13 channels; the four extra channels (flip(user2item) -> user2item, tag2user -> user2item, and the two tag2item
    endings) are checked against the CPU oracle at full size."""
    dataset.spec = dict(dataset.spec, num_metapaths=13)
    model = build_model(dataset, args.kind, device)
    model.train()

    def step():
        with torch.no_grad():
            return model.loss(batch)

    for _ in range(3):
        step()
    steps = max(3, args.steps // 2)
    dt, loss = timed_region(step, steps)
    dt /= steps
    out = {'ms_per_step': dt * 1e3, 'messages_per_step': model._engine.messages, 'edges_per_s': model._engine.messages / dt,
           'loss': float(loss)}
    if with_oracle:
        extra = [9, 10, 11, 12]
        outs, _, sd = oracle_channels(dataset, model, args.kind, extra)
        with torch.no_grad():
            _, stack = model.forward(return_stack=True)
        got = stack[:, extra].cpu().numpy()
        want = np.stack(outs, axis=1)
        out['parity_vs_cpu_oracle'] = {'channels': [p + 1 for p in extra], 'max_abs_err': float(np.abs(got - want).max()),
                                       'max_abs_value': float(np.abs(want).max())}
        if args.kind in ('gat', 'sage'):
            # Metapath 10 ends in user2item: its hottest items reduce up to ~2 M messages each, which the reference's
            # fp32 scatter (and so the CPU oracle) sums SEQUENTIALLY, while the HIP path folds 512-edge chunks.  Which
            # side carries the difference is settled in float64 on that one layer with IDENTICAL fp32 inputs for all
            # three: the oracle's own first-layer output h1 -> HIP conv drop-in / fp32 oracle / float64 (oracle/rows64.py)
            # at sampled item rows, the hottest included.
            from oracle import oracle as orc
            from oracle.rows64 import f64_rows_one_step
            from graph_recsys_benchmark_amd.utils import metapath_table
            p = 9
            (r1, f1), (r2, f2) = metapath_table(dataset.dataset_args())[p]
            rel = lambda r, f: np.ascontiguousarray(dataset.edge_index_nps[r].astype(np.int64)[::-1]) if f else dataset.edge_index_nps[r].astype(np.int64)
            lps = [{k.split('gnn_layers.%d.' % s_)[1]: v for k, v in sd.items()
                    if k.startswith('pea_channels.%d.gnn_layers.%d.' % (p, s_))} for s_ in range(2)]
            h1 = orc.relu_(orc.conv(args.kind, sd['x'], rel(r1, f1), lps[0], 1))
            e2 = rel(r2, f2)
            want2 = orc.conv(args.kind, h1, e2, lps[1], 1)
            with torch.no_grad():
                got2 = model.pea_channels[p].gnn_layers[1](torch.from_numpy(h1).to(device), model.meta_path_edge_index_list[p][1]).cpu().numpy()
            deg = np.bincount(e2[1], minlength=dataset.num_nodes)
            order = np.argsort(-deg)
            rows = np.unique(np.concatenate([order[:8], order[100:104], np.random.default_rng(7).choice(order[:dataset.num_iids], 16)]))
            truth = f64_rows_one_step(args.kind, lps[1], h1, e2, rows)
            out['last_layer_of_metapath_10_vs_float64'] = {
                'rows': int(rows.size), 'max_messages_per_row': int(deg.max()),
                'hip_max_abs_err': float(np.abs(got2[rows] - truth).max()),
                'cpu_oracle_fp32_max_abs_err': float(np.abs(want2[rows] - truth).max()),
                'max_abs_value': float(np.abs(truth).max())}
    return out


if __name__ == '__main__':
    main()
