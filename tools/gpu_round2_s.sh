cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_var
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for cfg in "8 first_group_split_max_tiles=4096" "8 first_group_split_max_tiles=0" "16 first_group_split_max_tiles=4096" "16 first_group_split_max_tiles=0" "32 first_group_split_max_tiles=4096" "32 first_group_split_max_tiles=0"; do
  set -- $cfg
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$i -o t -- python $R/tools/prof_small.py $1 200 new $2 > $R/gpurun_out/prof_var/log_$i.txt 2>&1
  f=$(find /tmp/pv_$i -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/prof_var/B$1_$2.csv"
  echo "== B=$1 $2"; python - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls'])>=200:
        tot+=float(r['AverageNs'])/1e3*int(r['Calls'])/200
        print('   %-60s %7.2f'%(r['Name'].replace('aae::','').replace('void ','')[:60], float(r['AverageNs'])/1e3))
print('   sum %.1f'%tot)
PY
done
