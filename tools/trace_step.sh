cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace1
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-exchange-probe > $OUT/log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_gaps.py $OUT/tr_kernel_trace.csv; python3 - <<'PY'
import csv,os
f=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/trace1/tr_kernel_trace.csv"
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: from last preprocess kernel
idx=[i for i,r in enumerate(rows) if "preprocess" in r["Kernel_Name"]]
s=idx[-1]
step=rows[s:]
t0=int(step[0]["Start_Timestamp"])
prev_end=t0
gaps=0
for i,r in enumerate(step):
    st,en=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=st-prev_end
    if gap>0: gaps+=gap
    name=r["Kernel_Name"]
    if "copyBuffer" in name or gap>20000:
        print(f'{(st-t0)/1e3:9.1f}us dur {(en-st)/1e3:7.1f} gap {gap/1e3:6.1f} grid {r["Grid_Size_X"]:>9} {name[:70]}  | prev: {step[i-1]["Kernel_Name"][:40] if i else ""}')
    prev_end=max(prev_end,en)
print("step span ms",(prev_end-t0)/1e6,"sum gaps ms",gaps/1e6,"kernels",len(step))
PY
rm -f $OUT/tr_kernel_trace.csv
