cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q -k "rccl" > gpurun_out/gpu_tests6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests6.log
tail -15 gpurun_out/gpu_tests6.log
# the multi-process launch line of the driver, on the one GPU (world_size 1): exercises the N > 1 code of bench.py as far as one GPU can
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-split-precision --no-extras > gpurun_out/bench_tr1.json 2> gpurun_out/bench_tr1.err; echo "torchrun bench rc=$?"
tail -c 300 gpurun_out/bench_tr1.json
