#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes of the weight-gradient micro-benchmark (profiles/tools/gw_bench.py): where do the waves
# of gw_stage1_lds spend their cycles?  Usage: bash profiles/collect_sq_gw.sh [tag]
set -o pipefail
TAG=${1:-sqgw}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RUN="python3 profiles/tools/gw_bench.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- $RUN > $OUT/sq1.log 2>&1 || echo "sq1 failed"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- $RUN > $OUT/sq2.log 2>&1 || echo "sq2 failed"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/sq3 -- $RUN > $OUT/sq3.log 2>&1 || echo "sq3 failed"
python3 - <<PY
import csv, glob, collections
for p in ('sq1', 'sq2', 'sq3'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            if 'gw_stage1' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        print(p, k, {c: round(x / max(n[(k, c)], 1)) for c, x in v.items()})
PY
