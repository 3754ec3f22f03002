make -C graph_recsys_benchmark_amd/csrc -j16 -s 2>&1 | grep -E "error"
for spec in "ml25m_shaped gcn" "ml25m_shaped sage" "yelp_shaped sage" "yelp_shaped gat" "ml_small gcn" "ml_small gat"; do
  set -- $spec
  tag=r02_$1_$2
  bash profiles/collect.sh $tag --preset $1 --kind $2 > gpurun_out/collect_$tag.log 2>&1
  python profiles/summarize.py gpurun_out/prof_$tag gpurun_out/summary_$tag.json > gpurun_out/summary_$tag.txt
  rm -rf gpurun_out/prof_$tag
done
python bench.py --emulate-world 8 --steps 10 > gpurun_out/r2_emu8_after.json 2> gpurun_out/r2_emu8_after.err
python bench.py --emulate-world 4 --steps 10 > gpurun_out/r2_emu4_after.json 2>/dev/null
python bench.py --emulate-world 2 --steps 10 > gpurun_out/r2_emu2_after.json 2>/dev/null
python bench.py --train-steps 8 --no-extras --no-cpu-baseline > gpurun_out/r2_b9_train.json 2> gpurun_out/r2_b9_train.err
python bench.py --emulate-world 8 --train-steps 6 --steps 10 > gpurun_out/r2_emu8_train.json 2> gpurun_out/r2_emu8_train.err
echo done
