#!/bin/bash
# usage: tools/ab_stamps.sh <lib suffix>...  -- phase-stamp builds (-DNEMO_STAMPS): per-phase microseconds per job-step
cd "$GRAFT_REPO_ROOT"
for sfx in "$@"; do
  echo "== $sfx"
  export NEMO_HIP_LIB=$GRAFT_REPO_ROOT/nemo_amd/lib/libnemo_hip$sfx.so
  timeout -k 10 150 python tools/prof_solve.py --images ${IMAGES:-1024} --steps ${STEPS:-100} --lr0 2>&1 | grep -E "solve:|job-step|split|tables form|Error|error" || exit 1
done
