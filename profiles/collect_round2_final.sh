make -C graph_recsys_benchmark_amd/csrc -j16 -s 2>&1 | grep -E "error"
bash profiles/collect.sh r02b > gpurun_out/collect_r02b.log 2>&1
python profiles/summarize.py gpurun_out/prof_r02b gpurun_out/summary_r02b.json > gpurun_out/summary_r02b.txt
cp $(find gpurun_out/prof_r02b/stats -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_r02b.csv
rm -rf gpurun_out/prof_r02b
bash profiles/collect.sh r02t --preset stress_10m --scale 0.3 > gpurun_out/collect_r02t.log 2>&1
python profiles/summarize.py gpurun_out/prof_r02t gpurun_out/summary_r02t.json > gpurun_out/summary_r02t.txt
cp $(find gpurun_out/prof_r02t/stats -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_r02t.csv
rm -rf gpurun_out/prof_r02t
for w in 2 4 8; do python bench.py --emulate-world $w --steps 10 > gpurun_out/r2_emu${w}_f2.json 2>/dev/null; done
python bench.py --emulate-world 8 --train-steps 6 --steps 10 > gpurun_out/r2_emu8_train_f2.json 2>/dev/null
python bench.py --train-steps 8 --no-extras --no-cpu-baseline > gpurun_out/r2_train_f2.json 2>/dev/null
for k in gcn sage; do python bench.py --kind $k --no-extras --cpu-samples 1 > gpurun_out/r2_ml25m_${k}_f2.json 2>/dev/null; done
python bench.py --preset yelp_shaped --kind sage --no-extras --cpu-samples 1 > gpurun_out/r2_yelp_sage_f2.json 2>/dev/null
head -6 gpurun_out/summary_r02b.txt
