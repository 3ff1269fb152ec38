#!/bin/bash
# Round 4: (a) the query-resident scan normalising its own queries against the l2norm_pack launch in front; (b) the dense layer as a
# GEMV up to B = 8 against the wave-split-K tile.  A B A B on one box + the kernel names of one config-5 query.   Usage: tools/gpu_r12_norm.sh <tag>
TAG=${1:-n}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r12_norm_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 420 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "resident or config5 or topk or mid or batch or dense or gemv or label" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log; tail -4 $OUT/pytest_subset.log | cut -c1-300
grep -q "rc=0" $OUT/pytest_subset.log || exit 1
timeout -s KILL 300 python tools/config5_ab.py > $OUT/scan_fused_norm_ab.jsonl 2> $OUT/scan_fused_norm_ab.err; echo "config5_ab rc=$?"; cut -c1-260 $OUT/scan_fused_norm_ab.jsonl
timeout -s KILL 400 python tools/latency_variants.py "base= packed=scan_mode=7 gemv4=dense_gemv_max_batch=4" 5,6,8,12,16,64 > $OUT/latency_mid_variants.jsonl 2> $OUT/latency_mid_variants.err; echo "latency rc=$?"; cut -c1-260 $OUT/latency_mid_variants.jsonl
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/tools/config5_ab.py > $OUT/prof_c5.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof_c5 -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_config5_ab.csv && head -12 $OUT/kernel_stats_config5_ab.csv | cut -c1-200
rm -rf $OUT/prof_c5
