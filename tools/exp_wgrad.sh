#!/bin/bash
# tools/exp_wgrad.sh <tag> <extra hipcc flags...>: library variant satflow_amd/lib/libsatflow_hip_<tag>.so that differs from the
# product build only in conv3x3_wgrad_bf16_dma.hip (compiled with the extra flags); links the other prebuilt objects.
TAG=$1; shift
cd $(dirname $0)/..
SRC=${SF_EXP_SRC:-conv3x3_wgrad_bf16_dma}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c satflow_amd/csrc/$SRC.hip -o /tmp/exp_$TAG.o || exit 1
OBJS=$(ls satflow_amd/lib/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/exp_$TAG.o -o satflow_amd/lib/libsatflow_hip_$TAG.so
