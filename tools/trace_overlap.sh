#!/bin/bash
# Do kernels of two streams really run at the same time?  rocprofv3 kernel trace of a few steps of a workload, then: time with >= 2 kernels in flight,
# which kernel pairs overlap, and a slice of the timeline.   tools/trace_overlap.sh <name> <env assignments...> -- <bench args...>
#   e.g. tools/trace_overlap.sh lstm_diag SF_LSTM_DIAG=1 -- --workload convlstm --steps 3 --warmup 2 --no-cpu-baseline --no-extra
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the snapshot root)}"
[ $# -ge 1 ] || { echo "usage: tools/trace_overlap.sh <name> <env assignments...> -- <bench args...>" >&2; exit 2; }
NAME=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ $# -gt 0 ] || { echo "trace_overlap.sh: missing '--' before the bench arguments" >&2; exit 2; }
shift
for e in ${ENVS[@]+"${ENVS[@]}"}; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py $(find $OUT -name 'tr_kernel_trace.csv' | head -1) | tee $OUT/overlap.txt
find $OUT -name 'tr_kernel_trace.csv' -delete
