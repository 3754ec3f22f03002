#!/bin/bash
# Runs ON THE GPU BOX: counter passes aimed at the dense-transform kernels (separate --pmc passes, kernel-trace only).
set -o pipefail
TAG=${1:-gemm}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1 || echo "sq1 failed"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1 || echo "sq2 failed"
find $OUT -name "*counter_collection.csv" | head
