#!/bin/bash
# usage: tools/ab.sh <lib suffix>...   -- runs tools/prof_solve.py (fixed-pose solve) once per diagnostic build
# nemo_amd/lib/libnemo_hip<suffix>.so ("-" = the shipped library) and prints the timing lines.
cd "$GRAFT_REPO_ROOT"
for sfx in "$@"; do
  [ "$sfx" = "-" ] && sfx=""
  echo "== ${sfx:-shipped}"
  if [ -n "$sfx" ]; then export NEMO_HIP_LIB=$GRAFT_REPO_ROOT/nemo_amd/lib/libnemo_hip$sfx.so; else unset NEMO_HIP_LIB; fi
  timeout -k 10 150 python tools/prof_solve.py --images ${IMAGES:-1024} --steps ${STEPS:-100} --lr0 2>&1 | grep -E "solve:|job-step|split|waves|tables form|Error|error" || exit 1
done
