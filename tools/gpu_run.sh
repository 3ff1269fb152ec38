#!/bin/bash
# One parameterised GPU-box script (replaces the per-round tools/gpu_r1*_*.sh): tools/gpu_run.sh <tag> <step> [<step> ...]
# Steps write under gpurun_out/<tag>/.  Run as: gpurun --timeout N -- 'bash tools/gpu_run.sh r15 wino_time wino_stamps'
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for step in "$@"; do
  echo "== $step"
  case $step in
    wino_time)    timeout 300 ./build/wino_layer_time 256 10 > $OUT/wino_layer_time.jsonl 2>&1; tail -20 $OUT/wino_layer_time.jsonl ;;
    wino_stamps)  timeout 300 ./build/wino_layer_stamps 256 3 > $OUT/wino_layer_stamps.jsonl 2>&1; cut -c1-1500 $OUT/wino_layer_stamps.jsonl ;;
    tests)        timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu_tail.log ;;
    bench)        timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-1200 $OUT/bench_default.json ;;
    bench_quick)  timeout 600 python bench.py --no-extras > $OUT/bench_quick.json 2> $OUT/bench_quick.err; cut -c1-1500 $OUT/bench_quick.json ;;
    smoke)        timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5 | tee $OUT/smoke.log ;;
    run:*)        # run:<binary under build/>:<args separated by commas>  -> $OUT/<binary>.jsonl
                  bin=$(echo $step | cut -d: -f2); args=$(echo $step | cut -d: -f3 | tr ',' ' ')
                  timeout 300 ./build/$bin $args > $OUT/$bin.jsonl 2>&1; echo "$bin rc=$?" ;;
    *)            echo "unknown step $step" ;;
  esac
done
