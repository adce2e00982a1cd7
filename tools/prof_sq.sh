#!/bin/bash
# usage: tools/prof_sq.sh <name> "<counters>"  -> gpurun_out/<name>/sq_counter_collection.csv   (probe_roofline.py kernels)
set -u
NAME=$1; CNT=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o sq -- python3 $GRAFT_REPO_ROOT/tools/probe_roofline.py > $OUT/sq.log 2>&1
tail -2 $OUT/sq.log
