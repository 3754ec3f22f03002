#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes of the default bench workload (separate --pmc passes).
set -o pipefail
TAG=${1:-sq}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1 || echo "sq1 failed"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1 || echo "sq2 failed"
# No TCP_* / TA_* pass: a pass holding several of them aborts rocprofv3 on gfx950 ("rocprofiler_create_counter_config ...
# error code 38: Request exceeds the capabilities of the hardware to collect", gpurun_out/prof_gemm1/tcp.log in round 1):
# the counter set exceeds the per-block slots, it is a profiler-configuration abort, not a hang and not a product fault.
ls $OUT/*/*/ | head
