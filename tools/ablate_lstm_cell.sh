#!/bin/bash
# Ablation of the fused ConvLSTM cell's epilogue.  Run HERE (hipcc cross-compiles) to build variant libraries into tools/ablate/
# (git-ignored, shipped to the GPU box by gpurun), then on the GPU box:  bash tools/ablate_lstm_cell.sh run
#   base     the shipped kernel
#   notrans  -DSF_EXP_LSTM_NOTRANS: sigmoid / tanh replaced by one multiply (what the gate arithmetic costs)
#   noepi    -DSF_EXP_NOEPI: no epilogue at all (no state loads, no gate arithmetic, no stores)
#   nostore  -DSF_EXP_LSTM_NOSTORE: the epilogue's stores predicated off;  nocprev  -DSF_EXP_LSTM_NOCPREV: no previous-cell-state loads
#   nostage  -DSF_EXP_NOSTAGE: no LDS-DMA / loads of the next chunk (MFMAs on stale LDS contents)
# and, without a rebuild (runtime arguments of sf_convlstm_cell_fwd): nogates = no saved gates (inference), fp32 h = fp32-stored hidden state.
set -u
cd "$(dirname "$0")/.."
A=tools/ablate; mkdir -p $A
if [ "${1:-build}" = run ]; then
  for v in base notrans nostore nocprev noepi nostage; do
    lib=$PWD/satflow_amd/lib/libsatflow_hip.so; [ $v != base ] && lib=$PWD/$A/libsatflow_$v.so
    SATFLOW_HIP_LIB=$lib python tools/probe_lstm_cell.py $v
  done
  exit 0
fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
OBJ=satflow_amd/lib/obj
others=$(ls $OBJ/*.o | grep -v "/conv3x3_bf16.o")
for v in notrans:-DSF_EXP_LSTM_NOTRANS nostore:-DSF_EXP_LSTM_NOSTORE nocprev:-DSF_EXP_LSTM_NOCPREV noepi:-DSF_EXP_NOEPI nostage:-DSF_EXP_NOSTAGE; do
  ( n=${v%%:*}; d=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS $d -c satflow_amd/csrc/conv3x3_bf16.hip -o $A/conv3x3_bf16_$n.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $A/conv3x3_bf16_$n.o -o $A/libsatflow_$n.so ) &
done
wait
ls -la $A
