#!/bin/bash
# ONE gpurun call: kernel-trace stats of every (workload, mode), PMC traffic passes of the dominant kernel, one full bench line.
# Results under gpurun_out/${ROUND}_*; tools/collect_profiles.sh copies the summaries into profiles/.
set -u
cd $GRAFT_REPO_ROOT
R=${ROUND:-r06}
for m in bf16a bf16 f32 f32e; do
  bash tools/prof_stats.sh ${R}_metnet_$m --dtype $m --no-cpu-baseline --no-extra > /dev/null 2>&1
  bash tools/prof_stats.sh ${R}_convlstm_$m --workload convlstm --dtype $m --no-cpu-baseline > /dev/null 2>&1
done
bash tools/prof_pmc.sh ${R}_pmc_metnet_bf16 bf16 > /dev/null 2>&1
# bf16a: HBM traffic of EVERY kernel of a short profiled run (the bench line's kernel table reads it per kernel name)
bash tools/prof_pmc_step.sh ${R}_pmc_step_bf16a --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
python tools/pmc_by_kernel.py gpurun_out/${R}_pmc_step_bf16a 30 --json profiles/${R}_metnet_bf16a_pmc_step.json > gpurun_out/${R}_pmc_step_bf16a/by_kernel.txt 2>&1
cp profiles/${R}_metnet_bf16a_pmc_step.json gpurun_out/${R}_pmc_step_bf16a/
# the full bench lines below read the sha-stamped traffic record from profiles/: refresh it on this box first (collect_profiles.sh repeats this at home)
KFB="conv3x3_bf16_kernel<8, 4, 0, false, true, false, false, false, false>"
python tools/parse_pmc.py gpurun_out/${R}_pmc_metnet_bf16 profiles/${R}_metnet_bf16_pmc_conv256.json "$KFB" > /dev/null 2>&1
ROUND=$R bash tools/prof_pmc_dgmr.sh ${R}_pmc_dgmr_conv > /dev/null 2>&1           # DGMR line's roofline launch
ROUND=$R bash tools/prof_pmc_cell.sh ${R}_pmc_convlstm_cell > /dev/null 2>&1   # fused ConvLSTM cell: traffic record for the ConvLSTM line
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_metnet_bf16a_bench_full.json 2> gpurun_out/${R}_metnet_bf16a_bench_full.err
python bench.py --workload convlstm --steps 20 --warmup 5 > gpurun_out/${R}_convlstm_bf16a_bench_full.json 2>> gpurun_out/${R}_metnet_bf16a_bench_full.err
python bench.py --workload cloudgan --steps 40 --warmup 20 --no-extra > gpurun_out/${R}_cloudgan_bench.json 2>> gpurun_out/${R}_metnet_bf16a_bench_full.err
python bench.py --workload stlstm --steps 10 --warmup 3 --no-extra > gpurun_out/${R}_stlstm_bf16a_bench.json 2>> gpurun_out/${R}_metnet_bf16a_bench_full.err
python bench.py --workload stlstm --dtype f32 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/${R}_stlstm_f32_bench.json 2>> gpurun_out/${R}_metnet_bf16a_bench_full.err
SF_NO_GRAPH=1 bash tools/prof_stats.sh ${R}_dgmr_bf16 --workload dgmr --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 900 python bench.py --workload dgmr --dtype bf16 --steps 5 --warmup 2 > gpurun_out/${R}_dgmr_bench_full.json 2>> gpurun_out/${R}_metnet_bf16a_bench_full.err
timeout 900 python bench.py --workload dgmr --dtype f16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${R}_dgmr_f16_bench.json 2>> gpurun_out/${R}_metnet_bf16a_bench_full.err
for m in bf16a bf16 f32 f32e; do for w in metnet convlstm; do cut -c1-200 gpurun_out/${R}_${w}_$m/bench.json; done; done
cut -c1-300 gpurun_out/${R}_metnet_bf16a_bench_full.json
