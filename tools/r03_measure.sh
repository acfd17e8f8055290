#!/bin/bash
# usage (on the GPU box): bash tools/r03_measure.sh <step>...   -- the round's measurement record; every step under its own
# time limit, the script stops at the first step that fails.  Output under gpurun_out/m3/; the summaries are copied to
# profiles/r03_* by hand.
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/m3; mkdir -p $O
prof() {  # prof <tag> <seconds> <program args...>: rocprofv3 kernel trace + stats of one command
  local tag=$1 lim=$2; shift 2
  timeout -k 10 $lim rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 "$@" > $O/prof_$tag.log 2>&1 || return 1
  cp $(ls $O/prof_$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
}
for step in "$@"; do
  echo "== $step"
  case $step in
    bench)    timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1 ;;
    pbench)   prof bench 300 bench.py --from-files-images 0 --no-cpu-baseline || exit 1 ;;
    pmc)      for c in FETCH_SIZE WRITE_SIZE; do
                timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/prof_solve.py --images 1024 --steps 300 --reps 1 --grid > $O/pmc_$c.log 2>&1 || exit 1
              done ;;
    pmc_tables) for c in FETCH_SIZE WRITE_SIZE; do
                timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $O/pmct_$c -- python3 tools/prof_solve.py --images 1024 --steps 300 --reps 1 > $O/pmct_$c.log 2>&1 || exit 1
              done
              python3 tools/pmc_tables_summary.py $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE $O/pmct_FETCH_SIZE.log $O/pmc_solve_tables.json 1024 || exit 1 ;;
    sq_tables) bash tools/pmc_sq.sh "" r03tab > $O/sq_solve_tables.txt 2>&1 || exit 1 ;;
    bench_whole) NEMO_SOLVE_LAZY=0 timeout -k 10 400 python3 bench.py --from-files-images 0 --no-cpu-baseline > $O/bench_whole_tables.json 2> $O/bench_whole.err || exit 1 ;;
    pmc_dense12) for c in FETCH_SIZE WRITE_SIZE; do
                timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmcd12_$c -- python3 tools/bench_configs.py dense12 > $O/pmcd12_$c.log 2>&1 || exit 1
              done
              python3 tools/pmc_kernels.py $O/pmcd12_FETCH_SIZE $O/pmcd12_WRITE_SIZE $O/pmc_dense12.json "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/bench_configs.py dense12" || exit 1 ;;
    clutter)  prof clutter 200 tools/time_clutter.py || exit 1 ;;
    head)     prof head 200 tools/prof_head.py || exit 1 ;;
    multi|dense|dense12|big)
              prof cfg_$step 400 tools/bench_configs.py $step || exit 1
              grep "^{" $O/prof_cfg_$step.log > $O/cfg_$step.json ;;
    all4)     prof cfg_all4 600 tools/bench_configs.py multi dense dense12 big || exit 1
              grep "^{" $O/prof_cfg_all4.log > $O/cfg_all4.json
              cp $(ls $O/prof_cfg_all4/*/*kernel_trace.csv | head -1) $O/cfg_all4_kernel_trace.csv ;;
    bigtrace) cp $(ls $O/prof_cfg_big/*/*kernel_trace.csv | head -1) $O/cfg_big_kernel_trace.csv ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
echo done
