cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python bench.py --force-dist --steps 10 --warmup 3 --no-split-precision > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo "dist bench rc=$?"
tail -3 gpurun_out/bench_dist1.err
