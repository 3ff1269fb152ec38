#!/bin/bash
# Per-layer option sweeps of the per-detection path under rocprofv3 (kernel durations without the event-timing launch gaps):
# every entry of CASES is "<B> <opt=value,opt=value>" for tools/prof_small.py.  This is how the wavek / conv1 defaults were set
# (profiles/r09_small/variants_tile_shape_and_depth.txt).   gpurun -- 'bash tools/gpu_sweep_small.sh'
CASES=${CASES:-"1 wavek_tiny_max_tiles=64|1 wavek_tiny_max_tiles=0|1 wavek_depth=3|4 wavek_narrow_max_tiles=128|1 first_group_split_max_tiles=0"}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_var
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
IFS='|'
for cfg in $CASES; do
  IFS=' ' read -r B OPTS <<< "$cfg"
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$i -o t -- python $R/tools/prof_small.py $B 200 new $OPTS > $R/gpurun_out/prof_var/log_$i.txt 2>&1
  f=$(find /tmp/pv_$i -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/prof_var/B${B}_${OPTS}.csv"
  echo "== B=$B $OPTS"; python - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls'])>=200:
        tot+=float(r['AverageNs'])/1e3*int(r['Calls'])/200
        print('   %-60s %7.2f'%(r['Name'].replace('aae::','').replace('void ','')[:60], float(r['AverageNs'])/1e3))
print('   sum %.1f'%tot)
PY
done
