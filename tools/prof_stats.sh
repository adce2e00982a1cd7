#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_stats.sh <name> <bench args...>
# rocprofv3 kernel-trace + stats of bench.py; results under gpurun_out/<name>/
set -u
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o prof -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench.log 2>&1
grep '"metric"' $OUT/bench.log > $OUT/bench.json
ls -la $OUT
rm -f $OUT/prof_kernel_trace.csv  # large; the stats file is what we keep
