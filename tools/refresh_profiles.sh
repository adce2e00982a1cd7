set -u
cd $GRAFT_REPO_ROOT
for m in bf16a bf16; do
  bash tools/prof_stats.sh r01_metnet_$m --dtype $m > /dev/null 2>&1
  bash tools/prof_stats.sh r01_convlstm_$m --workload convlstm --dtype $m > /dev/null 2>&1
done
bash tools/prof_stats.sh r01_metnet_f32 --dtype f32 > /dev/null 2>&1
bash tools/prof_stats.sh r01_convlstm_f32 --workload convlstm --dtype f32 > /dev/null 2>&1
for m in bf16a bf16; do bash tools/prof_pmc.sh pmc_metnet_$m $m > /dev/null 2>&1; done
for m in bf16a bf16 f32; do for w in metnet convlstm; do cut -c1-260 gpurun_out/r01_${w}_$m/bench.json; done; done
ls gpurun_out/pmc_metnet_bf16a
