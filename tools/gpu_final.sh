#!/bin/bash
# Round-end evidence run: smoke, parity tests, the DEFAULT bench command exactly as the driver runs it, the same
# command under rocprofv3 --kernel-trace --stats, then the PMC passes (each in its own run, --kernel-trace only).
# Usage: tools/gpu_final.sh <tag>     -> gpurun_out/final_<tag>/
TAG=${1:-final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout -s KILL 600 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?" >> $OUT/bench_default.log; tail -2 $OUT/bench_default.log | cut -c1-600
cd /tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_rocprof.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -size +8M -delete        # (the extras launch tens of thousands of kernels; the stats summary is what is kept)
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --profile-steps 1"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o bench -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT
cut -c1-170 $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | head -14
cat $OUT/pmc_status.txt
