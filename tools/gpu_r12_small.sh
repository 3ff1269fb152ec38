#!/bin/bash
# Round 4, per-detection kernels: rocprofv3 --kernel-trace --stats of the fused B = 1 / 2 / 4 query, default plan and named
# encoder-option variants, then eager latencies of the same variants in one process.
# Usage: tools/gpu_r12_small.sh <tag> "<variant>=<opts> ..." [pytest]      e.g.  "base= tiny8=wavek_tiny_waves=8"
TAG=${1:-a}
VARIANTS=${2:-"base="}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r12_small_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$3" = "pytest" ]; then
  timeout -s KILL 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
fi
timeout -s KILL 300 python tools/latency_variants.py "$VARIANTS" > $OUT/latency_variants.jsonl 2> $OUT/latency_variants.err || { tail -5 $OUT/latency_variants.err; echo "latency run failed: no profiling"; exit 1; }
cat $OUT/latency_variants.jsonl
cd /tmp
for B in 1 2 4; do
  for VV in $VARIANTS; do
    V=${VV%%=*}; OPT=${VV#*=}
    timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_B${B}_$V -o q -- python $GRAFT_REPO_ROOT/tools/prof_small.py $B 300 new $OPT > $OUT/trace_B${B}_$V.log 2>&1 || { echo "B=$B $V failed"; tail -3 $OUT/trace_B${B}_$V.log; exit 1; }
    S=$(find $OUT/trace_B${B}_$V -name "*kernel_stats.csv" | head -1)
    [ -n "$S" ] && cp $S $OUT/kernel_stats_B${B}_$V.csv
    rm -rf $OUT/trace_B${B}_$V
  done
done
cd $GRAFT_REPO_ROOT
python tools/kernel_stats_table.py $OUT/kernel_stats_B*.csv
