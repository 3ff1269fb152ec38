cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 tools/ubench/mfma_rate > gpurun_out/mfma_rate.jsonl 2>&1; echo "mfma rc=$?"
timeout 120 python tools/ablate_wavek.py > gpurun_out/ablate.jsonl 2> gpurun_out/ablate.err; echo "ablate rc=$?"
