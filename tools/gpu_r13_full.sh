#!/bin/bash
# round 5: the whole GPU suite, the default bench line, the estimator with eight classes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r13
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
timeout -s KILL 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
timeout -s KILL 600 python tools/bench_estimator_multi.py > $OUT/estimator_multi.jsonl 2> $OUT/estimator_multi.err; echo "estimator rc=$?"; tail -3 $OUT/estimator_multi.err
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
ls -la $OUT | tail -8
