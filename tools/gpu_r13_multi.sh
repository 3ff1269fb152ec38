#!/bin/bash
# round 5, first look at the grouped multi-object query: GPU tests of the new path, the bench line, rocprofv3 stats of a multi-object frame loop
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r13
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest_multi.log 2>&1; echo "pytest multi rc=$?"; tail -5 $OUT/pytest_multi.log
timeout -s KILL 600 python bench.py > $OUT/bench_multi.json 2> $OUT/bench_multi.err; echo "bench rc=$?"
timeout -s KILL 600 python tools/bench_multi.py > $OUT/multi.jsonl 2> $OUT/multi.err; echo "bench_multi rc=$?"; tail -3 $OUT/multi.err
(cd /tmp && timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_multi -o multi -- python $GRAFT_REPO_ROOT/tools/bench_multi.py --profile 1 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof_multi.err); echo "rocprof rc=$?"
find /tmp/prof_multi -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_multi.csv
ls -la $OUT
