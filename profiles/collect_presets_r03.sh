#!/bin/bash
# Runs ON THE GPU BOX (via gpurun), round 3: the other BASELINE presets / kinds through the default bench (training leg included
# where the model trains on one GPU) and the randomised sweeps.  Usage: bash profiles/collect_presets_r03.sh presets|fuzz
export TMPDIR=/tmp
if [ "${1:-presets}" = "presets" ]; then
  for spec in "ml25m_shaped gcn" "ml25m_shaped sage" "yelp_shaped sage" "yelp_shaped gat" "yelp_shaped gcn" "ml_small gcn" "ml_small gat"; do
    set -- $spec
    python bench.py --preset $1 --kind $2 --no-extras --cpu-samples 1 > gpurun_out/r3_$1_$2.json 2> gpurun_out/r3_$1_$2.err || echo "FAILED $1 $2"
    python - <<PY
import json
d = json.loads(open('gpurun_out/r3_$1_$2.json').read().strip().splitlines()[-1])
t = d.get('training_step', {})
print('$1 $2: %.3f ms/step, %.3e edges/s, train %.3f ms, parity %.2e' % (d['ms_per_step'], d['value'], t.get('ms_per_step', float('nan')), d['parity_vs_cpu_oracle']['max_abs_err']))
PY
  done
else
  cd profiles/tools
  python fuzz_backward.py 150 31 > ../../gpurun_out/r3_fuzz_backward.log 2>&1; tail -n 1 ../../gpurun_out/r3_fuzz_backward.log
  python fuzz_convs.py 150 32 > ../../gpurun_out/r3_fuzz_convs.log 2>&1; tail -n 1 ../../gpurun_out/r3_fuzz_convs.log
  python fuzz_scoring.py 120 33 > ../../gpurun_out/r3_fuzz_scoring.log 2>&1; tail -n 1 ../../gpurun_out/r3_fuzz_scoring.log
  FUZZ_TWOSTEP=1 python fuzz_parity.py 200 34 > ../../gpurun_out/r3_fuzz_twostep.log 2>&1; tail -n 1 ../../gpurun_out/r3_fuzz_twostep.log
  python fuzz_sharded.py 2 40 35 > ../../gpurun_out/r3_fuzz_sharded2.log 2>&1; tail -n 1 ../../gpurun_out/r3_fuzz_sharded2.log
  FUZZ_TWOSTEP=1 python fuzz_sharded.py 3 30 36 > ../../gpurun_out/r3_fuzz_sharded3.log 2>&1; tail -n 1 ../../gpurun_out/r3_fuzz_sharded3.log
fi
