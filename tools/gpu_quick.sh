#!/bin/bash
# quick GPU iteration: parity tests + bench + extra measurements; logs under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-quick}
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
tail -3 gpurun_out/bench_$TAG.log
shift
if [ -n "$1" ]; then timeout 600 python tools/bench_extra.py "$@" > gpurun_out/extra_$TAG.log 2>&1; cat gpurun_out/extra_$TAG.log | tail -40; fi
