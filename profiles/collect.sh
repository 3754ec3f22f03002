#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + HBM-traffic PMC passes of the default bench workload.
# Usage: bash profiles/collect.sh <tag> [extra bench.py flags, e.g. --preset stress_10m --scale 0.25]
#        -> gpurun_out/prof_<tag>/{stats,fetch,write,l2}/...
set -o pipefail
TAG=${1:-r02}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || echo "stats pass failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1 || echo "write pass failed"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- $BENCH > $OUT/l2.log 2>&1 || echo "l2 pass failed"
find $OUT -name "*.csv" | head -20
