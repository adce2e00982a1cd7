#!/bin/bash
# tools/ablate_w4.sh: variant libraries of the one-wave-per-SIMD convolution (conv3x3_bf16_persist4.hip built with -DSF_EXP_W4_*, linked against
# the shipped objects), then - on the GPU box - tools/probe_conv_w4.py once per variant (SATFLOW_HIP_LIB selects the library).
#   build (here):    tools/ablate_w4.sh build
#   run (gpurun):    tools/ablate_w4.sh run
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-CLK CLK_NODMA CLK_NOREAD CLK_NODMA_NOREAD CLK_NOEPI CLK_NODMA_NOREAD_NOEPI}"
OBJ=satflow_amd/lib/obj
mkdir -p tools/ablate
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    defs=""
    for d in ${v//_/ }; do [ "$d" != base ] && defs="$defs -DSF_EXP_W4_$d"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $defs -c satflow_amd/csrc/conv3x3_bf16_persist4.hip -o tools/ablate/p4_$v.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v conv3x3_bf16_persist4.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ablate/p4_$v.o -o tools/ablate/libsatflow_w4_$v.so || exit 1
    echo "built $v ($defs)"
  done
else
  for v in $VARIANTS; do
    echo "== $v"
    SATFLOW_HIP_LIB=$PWD/tools/ablate/libsatflow_w4_$v.so python tools/probe_conv_w4.py 2>&1 | grep 'folded\|shader\|input gradient' | sort | uniq -c | sort -rn | head -8
  done
fi
