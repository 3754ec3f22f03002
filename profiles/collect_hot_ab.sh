#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes of the default bench workload with the plain long-row kernel (PEA_HOT=0) and with
# the LDS-staged hot sources (PEA_HOT=1): where the cycles of the dominant gather go in each.  Separate --pmc passes,
# kernel-trace only.  -> gpurun_out/prof_<tag>_hot{0,1}/{sq1,sq2,sq3}
set -o pipefail
TAG=${1:-hotab}
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extras"
for HOT in 0 1; do
  export PEA_HOT=$HOT
  OUT=gpurun_out/prof_${TAG}_hot$HOT
  mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1 || echo "sq1 failed"
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1 || echo "sq2 failed"
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq3 -- $BENCH > $OUT/sq3.log 2>&1 || echo "sq3 failed"
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- $BENCH > $OUT/l2.log 2>&1 || echo "l2 failed"
done
ls gpurun_out/prof_${TAG}_hot*/*/ | head -30
