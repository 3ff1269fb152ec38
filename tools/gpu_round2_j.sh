cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests4.log
tail -30 gpurun_out/gpu_tests4.log
