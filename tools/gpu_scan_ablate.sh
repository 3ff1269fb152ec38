#!/bin/bash
# Builds (if missing) and runs the compile-time ablations of the query-resident bf16 scan (tools/ubench/scan_resident_ablate.hip)
# on the GPU box; one JSON line per (ablation, kernel form) -> gpurun_out/scan_resident_ablate.jsonl
cd "$(dirname "$0")/.."
mkdir -p build/ubench gpurun_out
out=gpurun_out/scan_resident_ablate.jsonl
: > $out
for a in 0 1 4 8 5 16 32 64 128; do
  bin=build/ubench/scan_resident_ablate_$a
  [ -x $bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I augmentedautoencoder_amd/csrc -DAAE_SCAN_RESIDENT_ABLATE=$a -o $bin tools/ubench/scan_resident_ablate.hip
  timeout 120 $bin >> $out 2>> gpurun_out/scan_resident_ablate.err
done
cat $out
