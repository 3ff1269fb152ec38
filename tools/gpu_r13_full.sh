#!/bin/bash
# round 5: the whole GPU suite on the product library, again on the experiments library, the default bench line, the estimator with eight classes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r13
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
if [ -f augmentedautoencoder_amd/libaae_hip_experiments.so ]; then
  AAE_EXPERIMENTS=1 timeout -s KILL 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_experiments.log 2>&1; echo "pytest experiments rc=$?"; tail -4 $OUT/pytest_gpu_experiments.log
fi
timeout -s KILL 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
timeout -s KILL 600 python tools/bench_estimator_multi.py > $OUT/estimator_multi.jsonl 2> $OUT/estimator_multi.err; echo "estimator rc=$?"; tail -3 $OUT/estimator_multi.err
ls -la $OUT | tail -8
