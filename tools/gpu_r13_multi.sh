#!/bin/bash
# round 5: the grouped multi-object query -- GPU tests of the new path, bench line, objects x detections sweep, rocprofv3 stats of 8 x 1 / 8 x 4 frames
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r13
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest_multi.log 2>&1; echo "pytest multi rc=$?"; tail -5 $OUT/pytest_multi.log
timeout -s KILL 600 python tools/bench_multi.py > $OUT/multi.jsonl 2> $OUT/multi.err; echo "bench_multi rc=$?"; tail -3 $OUT/multi.err
prof() {   # name, args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- python $GRAFT_REPO_ROOT/tools/bench_multi.py "$@" > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof_$name.err); echo "rocprof $name rc=$?"
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_multi_$name.csv
}
prof 8x1_group_plan --profile 1 --sequential
prof 8x1_per_object_plans --profile 1 --per-object-plans
prof 8x4_group_plan --profile 4 --sequential
ls -la $OUT
