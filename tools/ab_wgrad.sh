#!/bin/bash
# A/B of library variants inside ONE gpurun call: tools/ab_wgrad.sh <tag> <tag> ...   ("base" = the product library)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in "$@"; do
  if [ "$t" = base ]; then L=$GRAFT_REPO_ROOT/satflow_amd/lib/libsatflow_hip.so; else L=$GRAFT_REPO_ROOT/satflow_amd/lib/libsatflow_hip_$t.so; fi
  SATFLOW_HIP_LIB=$L SF_ACT=bf16 SF_ONLY=${SF_ONLY:-1} python tools/probe_wgrad.py 2>&1 | grep wgrad
done
done
