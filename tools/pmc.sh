#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [driver args...]   (one PMC pass, CSV output under gpurun_out/pmc_<tag>)
set -u
tag=$1; shift; ctrs=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/prof_solve.py "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
