#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats (csv) + PMC passes (each in its own run,
# --kernel-trace only, as the pool requires).  Usage: tools/gpu_pmc.sh <tag>
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
grep -i -E "mfma|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_BUSY_CYC|LDS_BANK_CONFLICT|TCC_HIT|TCC_MISS|TCC_EA0_RDREQ|SQ_WAIT|SQ_ACTIVE_INST" $OUT/counters_available.txt | head -60 > $OUT/counters_grep.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o bench -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*.csv" | head -30
cat $OUT/pmc_status.txt
du -sh $OUT
