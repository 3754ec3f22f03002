#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction / wait counters of the level launches of one model kind on the default graph.
# Usage: bash profiles/collect_sq_kind.sh gcn|sage|gat
set -o pipefail
KIND=${1:-gcn}
OUT=gpurun_out/prof_sqk_$KIND
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --kind $KIND --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extras --train-steps 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1 || echo "sq failed"
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('$OUT/sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(agg_rows_kernel<\d+, *\d+, *\d+>)', r['Kernel_Name'])
        if not m: continue
        acc[m.group(1)][r['Counter_Name']] += float(r['Counter_Value']); n[(m.group(1), r['Counter_Name'])] += 1
for k, v in sorted(acc.items()):
    print('$KIND', k, {c: round(x / max(n[(k, c)], 1)) for c, x in v.items()})
PY
rm -rf $OUT/sq
