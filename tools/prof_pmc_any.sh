#!/bin/bash
# usage: tools/prof_pmc_any.sh <name> <script.py> "<counters>"
set -u
NAME=$1; SCRIPT=$2; CNT=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o pmc -- python3 $GRAFT_REPO_ROOT/$SCRIPT > $OUT/pmc.log 2>&1
tail -1 $OUT/pmc.log
