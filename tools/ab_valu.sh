#!/bin/bash
# usage: tools/ab_valu.sh <lib suffix>...  -- vector instructions and busy cycles of the solve kernel per diagnostic build
# ("-" = the shipped library): one rocprofv3 --pmc pass over tools/prof_solve.py (1024 jobs x 100 free-running steps).
# Instruction counts do not drift with the box's clocks the way wall times do.
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for sfx in "$@"; do
  [ "$sfx" = "-" ] && sfx=""
  if [ -n "$sfx" ]; then export NEMO_HIP_LIB=$GRAFT_REPO_ROOT/nemo_amd/lib/libnemo_hip$sfx.so; else unset NEMO_HIP_LIB; fi
  d=gpurun_out/abv_${sfx:-shipped}; rm -rf $d
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $d -- python3 tools/prof_solve.py --images 1024 --steps 100 --reps 1 > $d.log 2>&1 || { echo "pass failed"; tail -3 $d.log; exit 1; }
  echo "== ${sfx:-shipped}"; python3 tools/pmc_read.py $d | grep -A6 "solve_kernel"
done
