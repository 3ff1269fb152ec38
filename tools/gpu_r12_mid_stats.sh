#!/bin/bash
# rocprofv3 --kernel-trace --stats of 300 fused queries at mid batch sizes, defaults against the planner without the tail cut
# and with K splits for one block per CU (the two planner changes of the second half of round 4).  Usage: tools/gpu_r12_mid_stats.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r12_mid_stats
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for B in 9 12 20; do
  for V in default:none=0 whole_tiles_one_block_per_cu:wavek_tail_split=0,wavek_g_boost=1; do
    NAME=${V%%:*}; OPT=${V#*:}
    [ "$NAME" = default ] && OPT=wavek_tail_split=1
    timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_${B}_$NAME -o q -- python $GRAFT_REPO_ROOT/tools/prof_small.py $B 300 new $OPT > $OUT/t_${B}_$NAME.log 2>&1 || { echo "B=$B $NAME failed"; tail -3 $OUT/t_${B}_$NAME.log; exit 1; }
    f=$(find $OUT/t_${B}_$NAME -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats_B${B}_$NAME.csv; rm -rf $OUT/t_${B}_$NAME
    echo "B=$B $NAME"; head -6 $OUT/kernel_stats_B${B}_$NAME.csv | cut -c1-150
  done
done
