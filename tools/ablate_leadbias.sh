#!/bin/bash
# tools/ablate_leadbias.sh build | run: variant libraries of metnet_pointwise.hip (-DSF_EXP_LB_*: one stream of leadbias_pool_bwd2_kernel squeezed into a
# cache-resident window - timing only, wrong results) and tools/prof_leadbias_ab.sh-style kernel timings per variant.
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-base NODOUT NOBASE NOSTORE NODOUT_NOBASE_NOSTORE}"
OBJ=satflow_amd/lib/obj
mkdir -p tools/ablate
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    defs=""
    for d in ${v//_/ }; do [ "$d" != base ] && defs="$defs -DSF_EXP_LB_$d"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $defs -c satflow_amd/csrc/metnet_pointwise.hip -o tools/ablate/mp_$v.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v metnet_pointwise.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ablate/mp_$v.o -o tools/ablate/libsatflow_lb_$v.so || exit 1
    echo "built $v ($defs)"
  done
else
  cd /tmp && export TMPDIR=/tmp
  for v in $VARIANTS; do
    export SATFLOW_HIP_LIB=$GRAFT_REPO_ROOT/tools/ablate/libsatflow_lb_$v.so
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_lb_abl_$v -o st -- python3 $GRAFT_REPO_ROOT/tools/probe_leadbias.py > /dev/null 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r04_lb_abl_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "bwd2" in r["Name"]: print(f'$v: {float(r["AverageNs"])/1e3:9.1f} us  bwd2')
PY
  done
fi
