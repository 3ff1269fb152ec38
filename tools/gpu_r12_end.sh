#!/bin/bash
# Round-4 closing evidence on ONE box: tools/gpu_final.sh (smoke, GPU tests, default bench, the same under rocprofv3 --stats, PMC passes), then
# the secondary tables DESIGN.md quotes.   Usage: tools/gpu_r12_end.sh <tag>   -> gpurun_out/final_<tag>/ + gpurun_out/end_<tag>/
TAG=${1:-end}
bash $GRAFT_REPO_ROOT/tools/gpu_final.sh $TAG
OUT=$GRAFT_REPO_ROOT/gpurun_out/end_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout -s KILL 400 python tools/bench_small.py latency > $OUT/latency_eager_and_graph.jsonl 2> $OUT/latency.err; echo "latency rc=$?"
timeout -s KILL 400 python tools/bench_extra.py estimator decoder config5 embed stream graph > $OUT/bench_extra.jsonl 2> $OUT/bench_extra.err; echo "bench_extra rc=$?"
timeout -s KILL 300 python tools/mid_batch_split.py > $OUT/mid_batch_kernel_split.jsonl 2> $OUT/mid_batch.err; echo "mid_batch rc=$?"
timeout -s KILL 300 python bench.py --gpus 1 --force-dist --no-extras > $OUT/bench_force_dist_world1.json 2> $OUT/force_dist.err; echo "force-dist rc=$?"
python - <<PY
import json
for l in open('$OUT/latency_eager_and_graph.jsonl'):
    d = json.loads(l)
    if d.get('what') == 'latency':
        print('B', d['B'], d['new']['encode+nn_us'], [k[0].split(':')[1].replace('conv_wavek_f32_', '')[:22] for k in d['new']['kernels_us']])
PY
