#!/bin/bash
# usage (GPU box): tools/prof_pmc_dgmr.sh <name>  - HBM traffic counters of the DGMR line's roofline launch, then profiles/<round>_dgmr_bf16_pmc_conv.json
set -u
NAME=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc_$C -- python3 $GRAFT_REPO_ROOT/tools/probe_dgmr_conv.py > $OUT/pmc_$C.log 2>&1
  tail -1 $OUT/pmc_$C.log
done
cd $GRAFT_REPO_ROOT
python tools/parse_pmc.py gpurun_out/$NAME profiles/${ROUND:-r05}_dgmr_bf16_pmc_conv.json "conv3x3_bf16_kernel<8, 4, 0, false, true, false, false, false, false>" | tail -8
cp profiles/${ROUND:-r05}_dgmr_bf16_pmc_conv.json gpurun_out/$NAME/
