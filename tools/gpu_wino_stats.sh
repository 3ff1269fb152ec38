#!/bin/bash
# rocprofv3 kernel stats of the Winograd A/B (tools/wino_ab.py)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/wino_stats
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python $GRAFT_REPO_ROOT/tools/wino_ab.py ${1:-256} > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -size +8M -delete
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for path in glob.glob('gpurun_out/wino_stats/trace/**/*kernel_trace.csv', recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        acc[(row['Kernel_Name'][:90], row['Grid_Size'])].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if 'wino' in k or 'igemm' in k:
            print('%-92s grid=%-9s n=%-4d avg_us=%.1f' % (k, g, len(v), sum(v) / len(v) / 1e3))
PY
