#!/bin/bash
# A/B of library variants on the dominant convolution (tools/probe_conv.py) inside ONE gpurun call: tools/ab_conv.sh <tag> ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in "$@"; do
  if [ "$t" = base ]; then L=$GRAFT_REPO_ROOT/satflow_amd/lib/libsatflow_hip.so; else L=$GRAFT_REPO_ROOT/satflow_amd/lib/libsatflow_hip_$t.so; fi
  SATFLOW_HIP_LIB=$L SF_ACT=bf16 python tools/probe_conv16.py 2>&1 | grep -i "conv" | sed "s/^/$t /"
done
done
