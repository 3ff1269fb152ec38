#!/bin/bash
# round 5: rocprofv3 kernel stats + PMC passes (each in its own run, --kernel-trace only) of a loop of grouped 8 x 1 frames
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r13_multi
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_multi.py --profile ${1:-1}"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o multi -- $CMD > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -size +8M -delete
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o multi -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT --json $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +4M -delete
cat $OUT/pmc_status.txt; grep -E "multi" $OUT/pmc_summary.txt | cut -c1-330
