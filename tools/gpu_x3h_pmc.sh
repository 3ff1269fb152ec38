#!/bin/bash
# f32x3h (opt-in split precision) evidence: the bench in --precision f32x3h under rocprofv3 --kernel-trace --stats,
# then the PMC passes (each its own run, --kernel-trace only).   Usage: tools/gpu_x3h_pmc.sh <tag> -> gpurun_out/x3h_<tag>/
TAG=${1:-x3h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/x3h_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --precision f32x3h --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-split-precision --profile-steps 2"
timeout 200 $BENCH > $OUT/bench_x3h.json 2> $OUT/bench_x3h.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o bench -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
find $OUT -name "*kernel_trace.csv" -size +8M -delete
cd $GRAFT_REPO_ROOT
cut -c1-170 $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | head -10
cat $OUT/pmc_status.txt; tail -c 600 $OUT/bench_x3h.json
