#!/bin/bash
# Round 4, per-detection kernels: GPU tests, then rocprofv3 --kernel-trace --stats of the fused B = 1 / 2 / 4 query with the
# default plan and with the 32 x 32 wave tiles on eight waves (two per SIMD).   Usage: tools/gpu_r12_small.sh <tag> [pytest]
TAG=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r12_small_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$2" = "pytest" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
fi
cd /tmp
for B in 1 2 4; do
  for V in base tiny8; do
    OPT=""
    [ "$V" = "tiny8" ] && OPT="wavek_tiny_waves=8"
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_B${B}_$V -o q -- python $GRAFT_REPO_ROOT/tools/prof_small.py $B 300 new $OPT > $OUT/trace_B${B}_$V.log 2>&1
    S=$(find $OUT/trace_B${B}_$V -name "*kernel_stats.csv" | head -1)
    [ -n "$S" ] && cp $S $OUT/kernel_stats_B${B}_$V.csv
    find $OUT/trace_B${B}_$V -name "*kernel_trace.csv" -size +4M -delete
  done
done
cd $GRAFT_REPO_ROOT
for f in $OUT/kernel_stats_B*.csv; do echo "== $f"; cut -d, -f1-4 $f | head -8 | cut -c1-150; done
timeout 400 python tools/bench_small.py latency > $OUT/latency.jsonl 2> $OUT/latency.err
python - <<PY
import json
for l in open('$OUT/latency.jsonl'):
    r = json.loads(l)
    print(r['B'], r['new']['encode+nn_us'], r['new'].get('graph_replay_us'), r['new']['nn_us'], r['new']['kernels_us'])
PY
