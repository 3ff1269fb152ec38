cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_small
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "1 new" "4 new"; do
  set -- $cfg
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$1_$2 -o t -- python $R/tools/prof_small.py $1 200 $2 > $R/gpurun_out/prof_small/log_$1_$2.txt 2>&1
  echo "prof $cfg rc=$?"
  f=$(find /tmp/ps_$1_$2 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/prof_small/kernel_stats_B$1_$2.csv
done
