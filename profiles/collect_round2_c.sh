#!/bin/bash
# Runs ON THE GPU BOX (via gpurun), end of round 2 after the launch fusion / SAGE two-step / HIP training head:
#   part A (default):  default workload r02c + stress x 0.3 r02u (stats + PMC), training-step kernel stats, emulate-world lines
#   part B (presets):  PMC summaries of the other presets / kinds
# Usage: bash profiles/collect_round2_c.sh A|B
make -C graph_recsys_benchmark_amd/csrc -j16 -s 2>&1 | grep -E "error"
export TMPDIR=/tmp
one() {  # tag, extra bench flags
  tag=$1; shift
  bash profiles/collect.sh $tag "$@" > gpurun_out/collect_$tag.log 2>&1
  python profiles/summarize.py gpurun_out/prof_$tag gpurun_out/summary_$tag.json > gpurun_out/summary_$tag.txt
  cp $(find gpurun_out/prof_$tag/stats -name "*kernel_stats.csv" | head -n 1) gpurun_out/kernel_stats_$tag.csv
  rm -rf gpurun_out/prof_$tag
}
if [ "${1:-A}" = "A" ]; then
  one r02c
  one r02u --preset stress_10m --scale 0.3
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 bench.py --train-steps 8 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-profile > gpurun_out/prof_train.log 2>&1
  cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -n 1) gpurun_out/kernel_stats_train_r02c.csv
  rm -rf gpurun_out/prof_train
  for w in 2 4 8; do python bench.py --emulate-world $w --steps 10 --no-extras > gpurun_out/r2c_emu${w}.json 2>/dev/null; done
  python bench.py --emulate-world 8 --train-steps 6 --steps 10 --no-extras > gpurun_out/r2c_emu8_train.json 2>/dev/null
  python bench.py --train-steps 8 --no-extras --cpu-samples 0 > gpurun_out/r2c_train.json 2>/dev/null
  head -n 8 gpurun_out/summary_r02c.txt
else
  for spec in "ml25m_shaped gcn" "ml25m_shaped sage" "yelp_shaped sage" "yelp_shaped gat" "yelp_shaped gcn" "ml_small gcn" "ml_small gat"; do
    set -- $spec
    one r02c_$1_$2 --preset $1 --kind $2
    python bench.py --preset $1 --kind $2 --no-extras --cpu-samples 1 > gpurun_out/r2c_$1_$2.json 2>/dev/null
  done
  ls gpurun_out | grep r02c_ | head -n 30
fi
