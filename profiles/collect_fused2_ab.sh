make -C graph_recsys_benchmark_amd/csrc -j16 -s 2>&1 | grep -E "error"
for cfg in "ml25m_shaped gat 1" "yelp_shaped gat 1" "yelp_shaped gcn 1" "stress_10m gat 0.3" "ml_small gat 1" "ml_small gcn 1"; do
  set -- $cfg
  for f in 0 1; do
    PEA_FUSED2=$f python bench.py --preset $1 --kind $2 --scale $3 --no-extras --no-cpu-baseline --steps 10 > gpurun_out/ab_$1_$2_f$f.json 2>/dev/null
  done
done
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t19.log 2>&1; echo rc=$? >> gpurun_out/r2_t19.log; tail -3 gpurun_out/r2_t19.log
