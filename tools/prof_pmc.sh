#!/bin/bash
# usage (on the GPU box): tools/prof_pmc.sh <name>   -> gpurun_out/<name>/{fetch,write}_counter_collection.csv
# Separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), kernel-trace only.
set -u
NAME=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc_$C -- python3 /root/repo/tools/probe_roofline.py > $OUT/pmc_$C.log 2>&1
done
ls -la $OUT
