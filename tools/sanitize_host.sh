#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer run of the C-ABI's argument validation (no GPU needed).
# Builds every csrc/*.hip with host-only sanitizers (-fno-gpu-sanitize: GPU ASan is unavailable on this pool), links
# tests/sanitize/harness.c against it with the same clang runtime, runs it.  Log: $1 (default /tmp/sf_host_sanitize.log).
set -u
cd "$(dirname "$0")/.."
LOG=${1:-/tmp/sf_host_sanitize.log}
OUT=/tmp/sf_asan; mkdir -p $OUT
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
pids=()
for f in satflow_amd/csrc/*.hip; do
  o=$OUT/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find satflow_amd/csrc include -name '*.h' -newer $o 2>/dev/null)" ] || [ -n "$(find satflow_amd/csrc -name '*.hip' -newer $o 2>/dev/null)" ]; then $HIPCC $FLAGS -c $f -o $o & pids+=($!); fi
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize $OUT/*.o -o $OUT/libsatflow_hip_asan.so || exit 1
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer tests/sanitize/harness.c -o $OUT/harness \
  -L$OUT -lsatflow_hip_asan -Wl,-rpath,$OUT -Wl,-rpath,/opt/rocm/lib || exit 1
{ echo "# tools/sanitize_host.sh: $(date -u +%F) hipcc host-side -fsanitize=address,undefined (-fno-gpu-sanitize), harness tests/sanitize/harness.c";
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $OUT/harness; echo "exit code $?"; } 2>&1 | tee $LOG
