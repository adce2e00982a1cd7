#!/bin/bash
# usage (GPU box): tools/prof_pmc.sh <name> [mode]  - HBM traffic counters of the dominant kernel (separate passes per the guide)
set -u
NAME=$1; MODE=${2:-bf16a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc_$C -- python3 $GRAFT_REPO_ROOT/tools/probe_roofline.py $MODE > $OUT/pmc_$C.log 2>&1
  tail -1 $OUT/pmc_$C.log
done
ls $OUT
