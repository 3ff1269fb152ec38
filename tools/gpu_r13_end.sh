#!/bin/bash
# Round-5 closing evidence on ONE box: tools/gpu_final.sh (smoke, GPU tests, default bench, the same under rocprofv3 --stats, PMC passes), then the
# tables DESIGN.md quotes for the multi-object regime.   Usage: tools/gpu_r13_end.sh <tag>   -> gpurun_out/final_<tag>/ + gpurun_out/end_<tag>/
TAG=${1:-r13}
bash $GRAFT_REPO_ROOT/tools/gpu_final.sh $TAG
OUT=$GRAFT_REPO_ROOT/gpurun_out/end_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout -s KILL 400 python tools/bench_multi.py > $OUT/multi_objects_sweep.jsonl 2> $OUT/multi.err; echo "bench_multi rc=$?"
timeout -s KILL 300 python tools/bench_estimator_multi.py > $OUT/estimator_multi.jsonl 2> $OUT/estimator_multi.err; echo "estimator_multi rc=$?"
timeout -s KILL 300 python tools/bench_extra.py estimator > $OUT/estimator.jsonl 2> $OUT/estimator.err; echo "estimator rc=$?"
timeout -s KILL 300 python bench.py --gpus 1 --force-dist --no-extras --config4 > $OUT/bench_force_dist_world1.json 2> $OUT/force_dist.err; echo "force-dist rc=$?"
if [ -f augmentedautoencoder_amd/libaae_hip_experiments.so ]; then
  timeout -s KILL 400 python tools/bench_small.py latency > $OUT/latency_eager_and_graph.jsonl 2> $OUT/latency.err; echo "latency rc=$?"
  AAE_EXPERIMENTS=1 timeout -s KILL 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_experiments.log 2>&1; echo "pytest experiments rc=$?"; tail -3 $OUT/pytest_gpu_experiments.log
fi
ls -la $OUT
