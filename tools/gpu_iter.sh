#!/bin/bash
# iteration run: parity tests, bench, chosen extra measurements, scan kernel trace
TAG=${1:-it}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1800
if [ -n "$1" ]; then timeout 600 python tools/bench_extra.py "$@" > gpurun_out/extra_$TAG.log 2>&1; tail -40 gpurun_out/extra_$TAG.log; fi
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/scanprof_$TAG -o scan -- python $GRAFT_REPO_ROOT/tools/bench_extra.py scanprof > $GRAFT_REPO_ROOT/gpurun_out/scanprof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT; cat gpurun_out/scanprof_$TAG/scan_kernel_stats.csv 2>/dev/null | cut -c1-160
