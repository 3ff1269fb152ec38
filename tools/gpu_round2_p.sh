cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_small
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 100 python tools/gpu_stage_small.py all > gpurun_out/stage4.jsonl 2> gpurun_out/stage4.err; echo "stage rc=$?"
timeout 600 python -m pytest tests -m gpu -x -q -k "small_batch or ticketed or captured or uint8" > gpurun_out/gpu_tests8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests8.log
tail -6 gpurun_out/gpu_tests8.log
cd /tmp
for cfg in "1 new" "4 new"; do
  set -- $cfg
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$1_$2 -o t -- python $R/tools/prof_small.py $1 200 $2 > $R/gpurun_out/prof_small/log_$1_$2.txt 2>&1
  echo "prof $cfg rc=$?"
  f=$(find /tmp/ps_$1_$2 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/prof_small/kernel_stats_B$1_$2.csv
done
cd $R
timeout 200 python tools/bench_small.py latency > gpurun_out/small7.jsonl 2> gpurun_out/small7.err; echo "small rc=$?"
