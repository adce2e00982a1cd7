#!/bin/bash
# tools/exp_build.sh <tag> <extra hipcc flags...>: experimental library variant satflow_amd/lib/libsatflow_hip_<tag>.so
TAG=$1; shift
cd $(dirname $0)/..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" satflow_amd/csrc/*.hip -o satflow_amd/lib/libsatflow_hip_$TAG.so
