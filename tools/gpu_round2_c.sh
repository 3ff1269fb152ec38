cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/stage.jsonl
for st in base gemv scan dense wavek all; do
  timeout 100 python tools/gpu_stage_small.py $st >> gpurun_out/stage.jsonl 2>> gpurun_out/stage.err; echo "{\"stage\": \"$st\", \"rc\": $?}" >> gpurun_out/stage.jsonl
done
timeout 240 python tools/bench_small.py latency > gpurun_out/small3.jsonl 2> gpurun_out/small3.err; echo "small rc=$?"
