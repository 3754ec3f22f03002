#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a collect_sq.sh / collect_hot_ab.sh output directory
(gpurun_out/prof_<tag>/sq*/...): python profiles/summarize_counters.py <dir> [kernel-name substring] [out.json]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import short  # noqa: E402


def main(root, want=''):
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, '*', '*', '*_counter_collection.csv')):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if 'pea' not in r['Kernel_Name']:
                continue
            k = short(r['Kernel_Name'])
            if want and want not in k:
                continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k].add(r['Dispatch_Id'])
        for k, v in agg.items():
            for c, x in v.items():
                out[k][c] = x / len(disp[k])
    return out


if __name__ == '__main__':
    res = main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    for k, v in sorted(res.items()):
        print(k)
        for c, x in sorted(v.items()):
            print('   %-28s %.4g' % (c, x))
