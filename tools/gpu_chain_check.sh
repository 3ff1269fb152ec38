#!/bin/bash
# First contact of the persistent per-detection launch with the GPU: barrier micro-benchmark, the parity / race tests that touch
# it, per-kernel durations (rocprofv3 --stats) of the fused query at B = 1, 2, 4 with and without it, eager latencies.
OUT=$GRAFT_REPO_ROOT/gpurun_out/chain_$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 120 tools/ubench/grid_barrier > $OUT/grid_barrier_ubench.jsonl 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_batch or race_free or stale or captured or realistic or saturation" 2>&1 | tail -15 > $OUT/pytest_chain.log
export TMPDIR=/tmp
cd /tmp
for B in 1 2 4; do
  for CH in 1 0; do
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_B${B}_chain${CH} -o t -- python $GRAFT_REPO_ROOT/tools/prof_small.py $B 200 new detect_chain=$CH > $OUT/trace_B${B}_chain${CH}.log 2>&1
    cp $(find $OUT/trace_B${B}_chain${CH} -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_B${B}_chain${CH}.csv 2>/dev/null
    rm -rf $OUT/trace_B${B}_chain${CH}
  done
done
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_chain.py 300 > $OUT/bench_chain.jsonl 2>&1
cat $OUT/grid_barrier_ubench.jsonl | tail -9
tail -5 $OUT/pytest_chain.log
for f in $OUT/kernel_stats_B*_chain*.csv; do echo $f; head -8 $f | cut -d, -f1-4 | cut -c1-150; done
tail -6 $OUT/bench_chain.jsonl
