cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for seed in 31 32 33 34; do
  timeout 500 python tools/gpu_fuzz.py 16 $seed > gpurun_out/fuzz_$seed.jsonl 2> gpurun_out/fuzz_$seed.err; echo "fuzz $seed rc=$?"; tail -1 gpurun_out/fuzz_$seed.jsonl | cut -c1-300; tail -3 gpurun_out/fuzz_$seed.err | cut -c1-600
done
