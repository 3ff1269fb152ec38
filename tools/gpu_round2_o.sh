cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "adversarial or captured or f32x3h or split_precision" > gpurun_out/gpu_tests7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests7.log
tail -25 gpurun_out/gpu_tests7.log
