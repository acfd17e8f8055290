#!/bin/bash
# usage: tools/ab_free.sh <rounds> <lib suffix>...  -- alternating free-running solves (1024 jobs x 300 steps) of diagnostic
# builds nemo_amd/lib/libnemo_hip<suffix>.so ("-" = the shipped library); prints the solve kernel's device time
cd "$GRAFT_REPO_ROOT"
rounds=$1; shift
for i in $(seq $rounds); do for sfx in "$@"; do
  [ "$sfx" = "-" ] && sfx=""
  echo "== ${sfx:-shipped}"
  if [ -n "$sfx" ]; then export NEMO_HIP_LIB=$GRAFT_REPO_ROOT/nemo_amd/lib/libnemo_hip$sfx.so; else unset NEMO_HIP_LIB; fi
  timeout -k 10 150 python tools/prof_solve.py --images 1024 --steps 300 --reps 3 2>&1 | grep "tables form" | tail -2 || exit 1
done; done
