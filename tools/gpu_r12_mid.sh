#!/bin/bash
# Round 4 checkpoint run: GPU tests, the default bench line, the estimator benchmark + host profile.   Usage: tools/gpu_r12_mid.sh <tag>
TAG=${1:-m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r12_mid_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-300
timeout -s KILL 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'frac', d['roofline']['frac'], 'lat', {k: v['encode+nn_us'] for k, v in d['latency'].items() if k != 'note'})
    print('scan', {k: d['scan'][k] for k in ('kernel_period_us', 'kernel_period_us_min', 'kernel_period_frac', 'B256_kernel_period_us', 'B1_whole_call_warm_us', 'B1_whole_call_cold_us', 'single_call_between_events_us', 'B256_whole_call_us')})
    print('config5', {k: d['config5'][k] for k in ('B256_argmax_us', 'B256_top5_us', 'B1_argmax_us')})
except Exception as e:
    print('bench parse failed', e)
PY
timeout -s KILL 300 python tools/bench_extra.py estimator > $OUT/bench_extra_estimator.jsonl 2> $OUT/bench_extra_estimator.err; cat $OUT/bench_extra_estimator.jsonl | cut -c1-400
timeout -s KILL 120 python tools/prof_estimator.py 1 300 > $OUT/prof_estimator_D1.txt 2>&1; head -3 $OUT/prof_estimator_D1.txt | tail -2
timeout -s KILL 120 python tools/prof_estimator.py 64 30 > $OUT/prof_estimator_D64.txt 2>&1; head -3 $OUT/prof_estimator_D64.txt | tail -2
