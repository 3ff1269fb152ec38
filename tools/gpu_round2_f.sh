cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 150 python tools/ablate_wavek.py > gpurun_out/ablate2.jsonl 2> gpurun_out/ablate2.err; echo "ablate rc=$?"
