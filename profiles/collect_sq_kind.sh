#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction / wait counters of the level launches of one model kind on the default graph.
# Usage: bash profiles/collect_sq_kind.sh gcn|sage|gat [train]     (train: the training step's kernels, backward included)
set -o pipefail
KIND=${1:-gcn}
OUT=gpurun_out/prof_sqk_$KIND
mkdir -p $OUT
export TMPDIR=/tmp
TRAIN=0; PAT='agg_rows_kernel<\d+, *\d+, *\d+>'
if [ "${2:-}" = "train" ]; then TRAIN=4; PAT='(agg_rows_kernel|bwd_rows_kernel|mlp2_kernel|mlp2_bwd_kernel|gw_stage1_lds|colsum_stage1|block_sum_kernel|bwd_merge_kernel)(<[^>]*>)?'; fi
BENCH="python3 bench.py --kind $KIND --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras --train-steps $TRAIN --grad-check-triples 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1 || echo "sq failed"
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('$OUT/sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'$PAT', r['Kernel_Name'])
        if not m: continue
        acc[m.group(0)][r['Counter_Name']] += float(r['Counter_Value']); n[(m.group(0), r['Counter_Name'])] += 1
for k, v in sorted(acc.items()):
    print('$KIND', k, {c: round(x / max(n[(k, c)], 1)) for c, x in v.items()})
PY
rm -rf $OUT/sq
