#!/bin/bash
# Round-5 closing evidence (second: with the Winograd conv layers) on ONE box: tools/gpu_final.sh (smoke, GPU tests, default bench, the same under rocprofv3 --stats, PMC passes), then the
# tables DESIGN.md quotes for the multi-object regime.   Usage: tools/gpu_r13_end.sh <tag>   -> gpurun_out/final_<tag>/ + gpurun_out/end_<tag>/
TAG=${1:-r14}
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
# the Winograd path: per-layer timing harness, the A/B against the direct kernels over batch sizes, the fused 3 x 3-phase micro-benchmark
timeout -s KILL 100 tools/ubench/wino_layer_time.bin > $OUT/wino_layer_time.jsonl 2>&1; echo "wino_layer_time rc=$?"
WINO_MIN=1 WINO_MIN_BLOCKS=1 timeout -s KILL 300 python tools/wino_ab.py 8 12 16 24 32 48 64 93 96 128 192 256 > $OUT/winograd_vs_direct_every_layer_forced.jsonl 2> $OUT/wino_ab.err; echo "wino_ab forced rc=$?"
timeout -s KILL 300 python tools/wino_ab.py 8 12 16 24 32 48 64 93 96 128 192 256 > $OUT/winograd_vs_direct_by_batch.jsonl 2>> $OUT/wino_ab.err; echo "wino_ab rule rc=$?"
WF_ABLATE=1 timeout -s KILL 100 tools/ubench/polyphase_winograd.bin > $OUT/polyphase_winograd_ubench.jsonl 2>&1; echo "ubench rc=$?"
ls -la $OUT
