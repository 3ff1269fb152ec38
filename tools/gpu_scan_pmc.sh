#!/bin/bash
# rocprofv3 counter passes over the query-resident scan micro-benchmark (tools/ubench/scan_resident_ablate.hip, unablated build):
# one kernel-trace + stats run, then one run per counter set (collected alone with --kernel-trace, as the pool requires).
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_scan_resident
BIN=$GRAFT_REPO_ROOT/build/ubench/scan_resident_ablate_0
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o scan -- $BIN > $OUT/trace.log 2>&1
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o scan -- $BIN > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/prof_scan_resident --json gpurun_out/prof_scan_resident/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/pmc_status.txt
