#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace stats of a run that includes full training steps.
set -o pipefail
TAG=${1:-train}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 2 --train-steps 6 --no-cpu-baseline --no-profile --no-extras > $OUT/stats.log 2>&1 || echo "stats pass failed"
find $OUT -name "*kernel_stats.csv" | head
