#!/bin/bash
# round 5: PMC passes (each in its own run, --kernel-trace only) of the fused Winograd micro-benchmark incl. its ablation variants
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/wino_pmc
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
export WF_ABLATE=1
if [ -n "$1" ]; then CMDX="$GRAFT_REPO_ROOT/$1"; fi
CMD="${CMDX:-$GRAFT_REPO_ROOT/tools/ubench/polyphase_winograd.bin}"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o w -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT --json $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +4M -delete
cat $OUT/pmc_status.txt
