#!/bin/bash
# The ONE GPU-box script (round 6: replaces the per-round tools/gpu_r1*_*.sh, gpu_final.sh, gpu_pmc*.sh, gpu_wino_*.sh, gpu_scan_*.sh,
# gpu_x3h_*.sh -- they differed by a tag and a command; git history keeps them).
#   /usr/local/graft/bin/gpurun --timeout N -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'      -> gpurun_out/<tag>/
# Steps (in the order given):
#   smoke | tests | tests_experiments | bench | bench_quick | bench_config4
#   bench_rocprof              the default bench command under rocprofv3 --kernel-trace --stats (kernel_stats CSV kept, big traces dropped)
#   pmc_bench                  the six PMC passes (each its own run, --kernel-trace only, as the pool requires) of the headline step + pmc_summary
#   prof:<name>:<cmd>          rocprofv3 stats + the six PMC passes of an arbitrary command (commas stand for spaces) -> <name>_trace/, <name>_pmc*/
#   stats:<name>:<cmd>         rocprofv3 --kernel-trace --stats of an arbitrary command + per-symbol durations / gaps (tools/trace_gaps.py)
#   run:<binary>:<args>        build/<binary> with comma-separated args                                   -> <binary>.jsonl
#   py:<out>:<script>:<args>   python tools/<script> with comma-separated args                            -> <out>
#   env:<NAME=VALUE>           export for the following steps
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PMC_SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS")

prof() {    # prof <name> <command...>
  local name=$1; shift
  ( cd /tmp
    timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -o $name -- "$@" > $OUT/${name}_trace.log 2>&1
    find $OUT/${name}_trace -name "*kernel_trace.csv" -size +8M -delete
    local i=0
    for SET in "${PMC_SETS[@]}"; do
      i=$((i+1))
      timeout -s KILL 400 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/${name}_pmc$i -o $name -- "$@" > $OUT/${name}_pmc$i.log 2>&1
      echo "${name} pmc$i [$SET] rc=$?" >> $OUT/pmc_status.txt
    done )
  python $ROOT/tools/pmc_summary.py $OUT --json $OUT/${name}_pmc_summary.json > $OUT/${name}_pmc_summary.txt 2>&1
  find $OUT -name "*counter_collection.csv" -size +4M -delete
  cat $OUT/pmc_status.txt | tail -6
}

cd $ROOT
for step in "$@"; do
  echo "== $step"
  case $step in
    env:*)        export "${step#env:}" ;;
    smoke)        timeout -s KILL 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -4 $OUT/smoke.log ;;
    tests)        timeout -s KILL 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log | tee $OUT/pytest_gpu_tail.log ;;
    tests_experiments) AAE_EXPERIMENTS=1 timeout -s KILL 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest_gpu_experiments.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_experiments.log; tail -4 $OUT/pytest_gpu_experiments.log | tee $OUT/pytest_gpu_experiments_tail.log ;;
    bench)        timeout -s KILL 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cut -c1-700 $OUT/bench_default.json ;;
    bench_quick)  timeout -s KILL 600 python bench.py --no-extras > $OUT/bench_quick.json 2> $OUT/bench_quick.err; cut -c1-700 $OUT/bench_quick.json ;;
    bench_config4) timeout -s KILL 600 python bench.py --no-extras --config4 > $OUT/bench_config4.json 2> $OUT/bench_config4.err; python -c "import json; r=json.load(open('$OUT/bench_config4.json')); print(r['value'], r['config4']['value'], r['config4']['ms_per_batch'])" ;;
    bench_rocprof) ( cd /tmp; timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --no-config3-b64 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err )
                  find $OUT/trace -name "*kernel_trace.csv" -size +8M -delete
                  cut -c1-170 $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | head -14 ;;
    pmc_bench)    prof bench python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --profile-steps 1 ;;
    prof:*)       name=$(echo $step | cut -d: -f2); cmd=$(echo $step | cut -d: -f3- | tr ',' ' '); prof $name $cmd ;;
    stats:*)      name=$(echo $step | cut -d: -f2); cmd=$(echo $step | cut -d: -f3- | tr ',' ' ')
                  ( cd /tmp; timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -o $name -- $cmd > $OUT/${name}_trace.log 2>&1 )
                  python $ROOT/tools/trace_gaps.py $OUT/${name}_trace > $OUT/${name}_kernel_durations.txt 2>&1
                  find $OUT/${name}_trace -name "*kernel_trace.csv" -size +8M -delete
                  cut -c1-170 $(find $OUT/${name}_trace -name "*kernel_stats.csv" | head -1) | head -14 ;;
    run:*)        bin=$(echo $step | cut -d: -f2); args=$(echo $step | cut -d: -f3 | tr ',' ' ')
                  timeout -s KILL 300 ./build/$bin $args > $OUT/$bin.jsonl 2>&1; echo "$bin rc=$?" ;;
    py:*)         out=$(echo $step | cut -d: -f2); script=$(echo $step | cut -d: -f3); args=$(echo $step | cut -d: -f4- | tr ',' ' ')
                  timeout -s KILL 900 python tools/$script $args > $OUT/$out 2> $OUT/$out.err; echo "$script rc=$?"; tail -3 $OUT/$out | cut -c1-400 ;;
    *)            echo "unknown step $step" ;;
  esac
done
