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
                out.setdefault(k, {})['avg_ms'] = float(r['AverageNs']) / 1e6
                out[k]['calls'] = int(r['Calls'])
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
        fetch = v.get('FETCH_SIZE_per_launch')
        write = v.get('WRITE_SIZE_per_launch')
        if fetch is not None and write is not None:
            v['hbm_bytes_per_launch'] = (2.0 * fetch + write) * 1024.0
        h, m = v.get('TCC_HIT_sum_per_launch'), v.get('TCC_MISS_sum_per_launch')
        if h is not None and m is not None and h + m > 0:
            v['l2_hit_rate'] = h / (h + m)
    return out


if __name__ == '__main__':
    res = main(sys.argv[1])
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('avg_ms', 0)):
        print('%-24s %s' % (k, {a: (round(b, 4) if isinstance(b, float) and b < 1e4 else (int(b) if isinstance(b, float) else b)) for a, b in v.items()}))
