#!/bin/bash
# Runs ON THE GPU BOX: SQ / TCP counter passes of the default bench workload (separate --pmc passes).
set -o pipefail
TAG=${1:-sq}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1 || echo "sq1 failed"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1 || echo "sq2 failed"
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum --output-format csv -d $OUT/tcp -- $BENCH > $OUT/tcp.log 2>&1 || echo "tcp failed"
ls $OUT/*/*/ | head
