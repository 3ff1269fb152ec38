#!/bin/bash
# rocprofv3 evidence for the HBM-bound / per-detection launches (tools/prof_mix.py): one kernel-trace + stats run, then
# one run per PMC set (counter collection alone with --kernel-trace, as the pool requires).
# Usage: tools/gpu_pmc_small.sh <tag> [reps] [encoder options name=value,...]
TAG=${1:-small}
REPS=${2:-100}
OPTS=${3:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/prof_mix.py $REPS $OPTS"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o mix -- $CMD > $OUT/trace.log 2>&1
echo "trace rc=$?" > $OUT/pmc_status.txt
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o mix -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/prof_$TAG --json gpurun_out/prof_$TAG/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
cp $(find gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
# keep the merge-back small: the per-dispatch CSVs of the PMC passes are summarised above
find gpurun_out/prof_$TAG -name "*counter_collection.csv" -size +8M -delete
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +8M -delete
cat $OUT/pmc_status.txt
du -sh $OUT
