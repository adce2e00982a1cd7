cd $GRAFT_REPO_ROOT
for t in base NODMA NOMFMA; do
  if [ "$t" = base ]; then L=$GRAFT_REPO_ROOT/satflow_amd/lib/libsatflow_hip.so; else L=$GRAFT_REPO_ROOT/satflow_amd/lib/libsatflow_hip_$t.so; fi
  export SATFLOW_HIP_LIB=$L
  bash tools/prof_pmc_any.sh clk_$t tools/probe_wgrad1.py "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > /dev/null 2>&1
  echo "== $t"; KFILTER=wgrad_bf16_dma python tools/pmc_sum.py gpurun_out/clk_$t
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/clk_$t/pmc_kernel_trace.csv")) if "wgrad_bf16_dma" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("kernel us:", [round(x,1) for x in d])
PY
done
