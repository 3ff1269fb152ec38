#!/bin/bash
# round-end style run: smoke, parity tests, default bench (as the driver runs it), x3h profiles
TAG=${1:-full}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -3 gpurun_out/smoke_$TAG.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log; tail -3 gpurun_out/pytest_$TAG.log
timeout 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log; tail -2 gpurun_out/bench_$TAG.log | cut -c1-3000
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_x3h; mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 --precision f32x3h"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o bench -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT; cat $OUT/trace/bench_kernel_stats.csv | cut -c1-150 | head -8
