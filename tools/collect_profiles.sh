#!/bin/bash
# After tools/refresh_profiles.sh (gpurun): copy the judged summaries from gpurun_out/ into profiles/ (tracked).
set -u
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
for m in bf16a bf16 f32 f32e; do for w in metnet convlstm; do
  d=gpurun_out/${R}_${w}_$m
  [ -f $d/prof_kernel_stats.csv ] && cp $d/prof_kernel_stats.csv profiles/${R}_${w}_${m}_kernel_stats.csv
  [ -s $d/bench.json ] && cp $d/bench.json profiles/${R}_${w}_${m}_bench.json
done; done
[ -s gpurun_out/${R}_pmc_step_bf16a/${R}_metnet_bf16a_pmc_step.json ] && cp gpurun_out/${R}_pmc_step_bf16a/${R}_metnet_bf16a_pmc_step.json profiles/
for m in bf16; do
  d=gpurun_out/${R}_pmc_metnet_$m
  for c in FETCH_SIZE WRITE_SIZE; do [ -f $d/pmc_${c}_counter_collection.csv ] && python - $d/pmc_${c}_counter_collection.csv profiles/${R}_metnet_${m}_pmc_$c.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv3x3_bf16_" in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size"])
w.writeheader()
for r in rows: w.writerow({k: r[k] for k in w.fieldnames})
PY
  done
  # the probe's launch: bf16-stored activations take the persistent kernel, fp32-stored ones the one-item kernel
  if [ $m = bf16a ]; then KF="conv3x3_bf16_persist_kernel<4, 0>"; else KF="conv3x3_bf16_kernel<8, 4, 0, false, true, false, false, false, false>"; fi
  [ -d $d ] && python tools/parse_pmc.py $d profiles/${R}_metnet_${m}_pmc_conv256.json "$KF" > /dev/null
done
[ -s gpurun_out/${R}_pmc_dgmr_conv/${R}_dgmr_bf16_pmc_conv.json ] && cp gpurun_out/${R}_pmc_dgmr_conv/${R}_dgmr_bf16_pmc_conv.json profiles/
[ -s gpurun_out/${R}_pmc_convlstm_cell/${R}_convlstm_bf16a_pmc_cell.json ] && cp gpurun_out/${R}_pmc_convlstm_cell/${R}_convlstm_bf16a_pmc_cell.json profiles/
[ -s gpurun_out/${R}_cloudgan_bench.json ] && cp gpurun_out/${R}_cloudgan_bench.json profiles/${R}_cloudgan_bf16a_bench.json
for f in stlstm_bf16a_bench stlstm_f32_bench; do [ -s gpurun_out/${R}_$f.json ] && cp gpurun_out/${R}_$f.json profiles/${R}_$f.json; done
for f in metnet_bf16a_bench_full convlstm_bf16a_bench_full; do [ -s gpurun_out/${R}_$f.json ] && cp gpurun_out/${R}_$f.json profiles/${R}_$f.json; done
[ -s gpurun_out/${R}_parity_observed.jsonl ] && cp gpurun_out/${R}_parity_observed.jsonl profiles/${R}_parity_observed.jsonl
[ -s gpurun_out/${R}_dgmr_bf16/prof_kernel_stats.csv ] && cp gpurun_out/${R}_dgmr_bf16/prof_kernel_stats.csv profiles/${R}_dgmr_bf16_kernel_stats.csv
[ -s gpurun_out/${R}_dgmr_bench_full.json ] && cp gpurun_out/${R}_dgmr_bench_full.json profiles/${R}_dgmr_bf16_bench_full.json
[ -s gpurun_out/${R}_dgmr_f16_bench.json ] && cp gpurun_out/${R}_dgmr_f16_bench.json profiles/${R}_dgmr_f16_bench.json
[ -s gpurun_out/${R}_full_tests.log ] && cp gpurun_out/${R}_full_tests.log profiles/${R}_gpu_tests.log
ls profiles | grep $R
