#!/bin/bash
# tools/ablate_gru.sh: timing-only variants of the ConvGRU sequence kernels (convgru_seq.hip built with -DSF_EXP_GRU_*, results are WRONG by
# construction), one library per variant, tools/probe_gru_seq.py once per variant on the GPU box.
#   build (here): tools/ablate_gru.sh build        run (gpurun): tools/ablate_gru.sh run
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-base NOPOLL NOSEND_NOPOLL NOSTORE NOTRANS NOMFMA NOGX NOSTORE_NOGX NOSEND_NOPOLL_NOSTORE_NOGX NOSEND_NOPOLL_NOSTORE_NOGX_NOTRANS NOSEND_NOPOLL_NOSTORE_NOGX_NOTRANS_NOMFMA}"
OBJ=satflow_amd/lib/obj
mkdir -p tools/ablate
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    defs=""
    for d in ${v//_/ }; do [ "$d" != base ] && defs="$defs -DSF_EXP_GRU_$d"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $defs -c satflow_amd/csrc/convgru_seq.hip -o tools/ablate/gru_$v.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v convgru_seq.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ablate/gru_$v.o -o tools/ablate/libsatflow_gru_$v.so || exit 1
    echo "built $v ($defs)"
  done
else
  for v in $VARIANTS; do
    echo "== $v"
    for n in ${MAPS:-96}; do
    SATFLOW_HIP_LIB=$PWD/tools/ablate/libsatflow_gru_$v.so python tools/probe_gru_seq.py $n 2>/dev/null | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read()); print('$n maps: fwd %.1f us (%.2f us/step)  bwd %.1f us (%.2f us/step)' % (d['fwd_us'], d['fwd_us'] / 24, d['bwd_us'], d['bwd_us'] / 24))"
    done
  done
fi
