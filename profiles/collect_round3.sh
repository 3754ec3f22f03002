#!/bin/bash
# Runs ON THE GPU BOX (via gpurun), round 3.
#   A: default workload (inference step) kernel stats + PMC passes -> r03a; the training step the same -> r03_train;
#      emulate-world lines (2, 4, 8); host cost per step
#   B: the default bench line (with training leg, gradient check, cpu baseline)
# Usage: bash profiles/collect_round3.sh A|B
export TMPDIR=/tmp
one() {  # tag, extra bench flags
  tag=$1; shift
  bash profiles/collect.sh $tag "$@" > gpurun_out/collect_$tag.log 2>&1
  python profiles/summarize.py gpurun_out/prof_$tag gpurun_out/summary_$tag.json > gpurun_out/summary_$tag.txt
  cp $(find gpurun_out/prof_$tag/stats -name "*kernel_stats.csv" | head -n 1) gpurun_out/kernel_stats_$tag.csv
  rm -rf gpurun_out/prof_$tag
}
if [ "${1:-A}" = "A" ]; then
  one r03a --train-steps 0
  one r03_train --train-steps 8 --steps 1 --warmup 1
  for w in 2 4 8; do python bench.py --emulate-world $w --steps 20 --no-extras --no-cpu-baseline > gpurun_out/r3_emu${w}.json 2>/dev/null; done
  python bench.py --emulate-world 8 --train-steps 6 --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r3_emu8_train.json 2>/dev/null
  python profiles/tools/host_profile.py 8 300 2>&1 | grep world > gpurun_out/r3_host8.txt
  python profiles/tools/host_profile.py 8 100 train 2>&1 | grep world > gpurun_out/r3_host8_train.txt
  head -n 12 gpurun_out/summary_r03a.txt; head -n 12 gpurun_out/summary_r03_train.txt
else
  python bench.py > gpurun_out/r3_default.json 2> gpurun_out/r3_default.err
  tail -c 600 gpurun_out/r3_default.json
fi
