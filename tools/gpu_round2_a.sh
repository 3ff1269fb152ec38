set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -15 gpurun_out/gpu_tests.log
timeout 600 python tools/bench_small.py latency variants thresholds > gpurun_out/small.jsonl 2> gpurun_out/small.err; echo "small rc=$?"
timeout 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/bench.json
