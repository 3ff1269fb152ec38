cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_small gpurun_out/final_r09b
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/final_r09b/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_r09b/pytest_gpu.log; tail -4 gpurun_out/final_r09b/pytest_gpu.log
cd /tmp
for cfg in "1 new" "2 new" "4 new"; do
  set -- $cfg
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$1_$2 -o t -- python $R/tools/prof_small.py $1 200 $2 > $R/gpurun_out/prof_small/log_$1_$2.txt 2>&1
  f=$(find /tmp/ps_$1_$2 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/prof_small/kernel_stats_B$1_$2.csv
done
cd $R
timeout 200 python tools/bench_small.py latency > gpurun_out/small8.jsonl 2> gpurun_out/small8.err; echo "small rc=$?"
timeout 300 python bench.py > gpurun_out/final_r09b/bench_default.json 2> gpurun_out/final_r09b/bench_default.err; echo "bench rc=$?"
