#!/usr/bin/env python3
"""Summarises a profiles/collect.sh run (gpurun_out/prof_<tag>/): per-kernel average duration from the
rocprofv3 kernel-trace stats and per-launch HBM-side counters.  FETCH_SIZE is doubled as
guides/MI355X_MICROARCH.md (section HBM) prescribes for gfx950 (it counts 64 B per 128-B request);
WRITE_SIZE is taken as is.  Units: rocprofv3 reports both in KiB."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r'(agg_\w+_kernel)<(\d+), *(\d+)(?:, *\d+)?>', name)
    if m:
        mode = {'0': 'gat', '1': 'gcn', '2': 'mean'}[m.group(3)]
        return '%s_g%s_%s' % (m.group(1).replace('_kernel', ''), m.group(2), mode)
    m = re.search(r'bwd_(rows|merge)_kernel<(\d+), *(\d+)', name)
    if m:                        # backward gather passes: MODE 3 = D (destination rows), 4 = S (source rows), 6 = the
        side = 'dst' if m.group(3) == '3' else 'src'          # weighted reverse sum of GCN / SAGE; bench.py names
        fam = 'sum_bwd' if m.group(3) == '6' else 'gat_bwd'
        return '%s_%s_g%s' % (fam, side, m.group(2)) if m.group(1) == 'rows' else '%s_%s_merge' % (fam, side)
    m = re.search(r'mlp2_kernel<\d+, *\d+, *(true|false|0|1)>', name)
    if m:                        # the training instance also stores the hidden tile: its own line
        return 'mlp2_kernel_train' if m.group(1) in ('true', '1') else 'mlp2_kernel'
    m = re.search(r'(gemm_mfma|gemm_persist|gemm_skinny)_kernel<(\d+)', name)
    if m:                        # bench.py names: gemm_persist = gemm_mfma_shared / _batch, gemm_skinny = gemm_mfma_narrow
        return '%s_k%s' % (m.group(1), m.group(2))
    m = re.search(r'pea::\(anonymous namespace\)::(\w+)', name)
    return m.group(1) if m else name[:48]


def main(root):
    out = {}
    f = glob.glob(os.path.join(root, 'stats', '*', '*_kernel_stats.csv'))
    if f:
        for r in csv.DictReader(open(f[0])):
            k = short(r['Name'])
            if 'pea' in r['Name']:
                e = out.setdefault(k, {})
                calls, tot = int(r['Calls']), float(r['AverageNs']) / 1e6 * int(r['Calls'])
                e['calls'] = e.get('calls', 0) + calls          # template instances that share a short name are pooled
                e['_ms'] = e.get('_ms', 0.0) + tot
                e['avg_ms'] = e['_ms'] / e['calls']
    for sub in ('fetch', 'write', 'l2'):
        f = glob.glob(os.path.join(root, sub, '*', '*_counter_collection.csv'))
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f[0])):
            if 'pea' not in r['Kernel_Name']:
                continue
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k].add(r['Dispatch_Id'])
        for k, v in agg.items():
            n = len(disp[k])
            for c, x in v.items():
                out.setdefault(k, {})[c + '_per_launch'] = x / n
    for k, v in out.items():
        v.pop('_ms', None)
        fetch = v.get('FETCH_SIZE_per_launch')
        write = v.get('WRITE_SIZE_per_launch')
        if fetch is not None and write is not None:
            v['hbm_bytes_per_launch'] = (2.0 * fetch + write) * 1024.0
        h, m = v.get('TCC_HIT_sum_per_launch'), v.get('TCC_MISS_sum_per_launch')
        if h is not None and m is not None and h + m > 0:
            v['l2_hit_rate'] = h / (h + m)
    return out


BENCH_NAME = {'gemm_persist_k32': 'gemm_mfma_shared', 'gemm_skinny_k4': 'gemm_mfma_narrow',   # names bench.py prints
              'gemm_persist_k64': 'gemm_mfma_shared', 'gemm_skinny_k8': 'gemm_mfma_narrow', 'mlp2_kernel': 'mlp2_fused', 'mlp2_sage_kernel': 'mlp2_fused', 'mlp2_kernel_train': 'mlp2_fused',
              'mlp2_bwd_kernel': 'mlp2_bwd_fused', 'block_sum_kernel': 'block_sum', 'gw_stage1_lds': 'grad_weight', 'gw_stage1': 'grad_weight'}


def merge_traffic(res, key, source):
    """profiles/traffic.json[key] = per-kernel fabric bytes per launch + L2 hit rate, key = '<preset>/<kind>': what
    bench.py reports as roofline.traffic for THAT workload (and nothing for any other)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'traffic.json')
    try:
        tab = json.load(open(path))
    except Exception:
        tab = {}
    if any(not isinstance(v, dict) or 'avg_ms' in v for k, v in tab.items() if not k.startswith('_')):
        tab = {}                                   # round-1 layout (keyed by kernel name only): start over
    tab['_doc'] = ('keys: <preset>/<kind> of a full-scale single-GPU bench.py run; per kernel (bench.py names): bytes = '
                   '(2*FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes (FETCH doubled on gfx950 per '
                   'guides/MI355X_MICROARCH.md); fabric-side bytes, Infinity-Cache hits included (upper bound on DRAM)')
    entry = {'_source': source}
    for k, v in res.items():
        if 'hbm_bytes_per_launch' not in v:
            continue
        entry[BENCH_NAME.get(k, k)] = {a: v[a] for a in ('avg_ms', 'hbm_bytes_per_launch', 'l2_hit_rate', 'calls') if a in v}
    tab[key] = entry
    json.dump(tab, open(path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    argv = list(sys.argv[1:])
    traffic_key = None
    if '--traffic' in argv:
        i = argv.index('--traffic')
        traffic_key = argv[i + 1]
        del argv[i:i + 2]
    res = main(argv[0])
    if len(argv) > 1:
        json.dump(res, open(argv[1], 'w'), indent=1, sort_keys=True)
    if traffic_key:
        merge_traffic(res, traffic_key, argv[1] if len(argv) > 1 else argv[0])
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('avg_ms', 0)):
        print('%-24s %s' % (k, {a: (round(b, 4) if isinstance(b, float) and b < 1e4 else (int(b) if isinstance(b, float) else b)) for a, b in v.items()}))
