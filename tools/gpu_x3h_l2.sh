#!/bin/bash
# L2 / fabric request mix of the f32x3h conv kernels, 256 x 256 kernel on and off.  Usage: tools/gpu_x3h_l2.sh <tag>
TAG=${1:-l2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/x3h_l2_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for WIDE in 1 0; do
  BENCH="python $GRAFT_REPO_ROOT/bench.py --precision f32x3h --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-split-precision --profile-steps 1 --enc-opt x3h_wide256=$WIDE"
  i=0
  for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/w${WIDE}_pmc$i -o bench -- $BENCH > $OUT/w${WIDE}_pmc$i.log 2>&1
    echo "wide=$WIDE pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
  done
done
find $OUT -name "*kernel_trace.csv" -size +8M -delete
cat $OUT/pmc_status.txt
