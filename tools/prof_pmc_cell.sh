#!/bin/bash
# usage (GPU box): tools/prof_pmc_cell.sh <name>  - HBM traffic counters of the fused ConvLSTM cell at the bench shape (separate passes per the guide),
# then the per-launch record profiles/<round>_convlstm_bf16a_pmc_cell.json that bench.py copies into the ConvLSTM line's roofline.traffic
set -u
NAME=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc_$C -- python3 $GRAFT_REPO_ROOT/tools/probe_lstm_cell.py pmc > $OUT/pmc_$C.log 2>&1
  tail -1 $OUT/pmc_$C.log
done
cd $GRAFT_REPO_ROOT
python tools/parse_pmc.py gpurun_out/$NAME profiles/${ROUND:-r05}_convlstm_bf16a_pmc_cell.json "conv3x3_bf16_kernel<4, 4, 2, false, true, false, true" | tail -8
cp profiles/${ROUND:-r05}_convlstm_bf16a_pmc_cell.json gpurun_out/$NAME/
