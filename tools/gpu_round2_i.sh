cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests3.log
tail -12 gpurun_out/gpu_tests3.log
timeout 200 python tools/bench_small.py thresholds > gpurun_out/small6.jsonl 2> gpurun_out/small6.err; echo "small rc=$?"
timeout 300 python bench.py > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench rc=$?"
