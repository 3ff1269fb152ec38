#!/bin/bash
# round 5: PMC passes (each in its own run, --kernel-trace only) of the B = 256 forward with the Winograd conv layers (tools/wino_ab.py)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/wino_layer_pmc
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/wino_ab.py 256"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o w -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT --json $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +4M -delete
cat $OUT/pmc_status.txt; grep -E "wino" $OUT/pmc_summary.txt | cut -c1-400
