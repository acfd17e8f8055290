#!/bin/bash
# usage: tools/ab_dense.sh <lib suffix>...  -- times tools/bench_configs.py dense with diagnostic builds of the library
cd "$GRAFT_REPO_ROOT"
for sfx in "$@"; do
  [ "$sfx" = "-" ] && sfx=""
  echo "== ${sfx:-shipped}"
  if [ -n "$sfx" ]; then export NEMO_HIP_LIB=$GRAFT_REPO_ROOT/nemo_amd/lib/libnemo_hip$sfx.so; else unset NEMO_HIP_LIB; fi
  timeout -k 10 200 python tools/bench_configs.py ${MODE:-dense} 2>&1 | grep -o '"seconds": [0-9.]*' | head -3 || exit 1
done
