set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_small.py latency variants > gpurun_out/small2.jsonl 2> gpurun_out/small2.err; echo "small rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q -k "small_batch or ticketed or captured or reentrant or scan_kernels" > gpurun_out/gpu_tests2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests2.log
tail -15 gpurun_out/gpu_tests2.log
