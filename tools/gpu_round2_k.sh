cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 400 python bench.py ) > gpurun_out/bench3.json 2> gpurun_out/bench3.err; echo "bench rc=$?"
tail -5 gpurun_out/bench3.err
