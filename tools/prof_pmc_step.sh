#!/bin/bash
# usage (GPU box): tools/prof_pmc_step.sh <name> <bench args...>  - HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) of EVERY kernel of a
# short bench run; tools/pmc_by_kernel.py prints per-kernel averages (FETCH doubled per MI355X_MICROARCH.md).
set -u
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pmc_$C.log 2>&1
  tail -1 $OUT/pmc_$C.log | cut -c1-200
  rm -f $OUT/pmc_${C}_kernel_trace.csv
done
