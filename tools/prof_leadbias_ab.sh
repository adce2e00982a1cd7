cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export SF_LEADBIAS_V1=1; else unset SF_LEADBIAS_V1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_lb_stats_$v -o st -- python3 $GRAFT_REPO_ROOT/tools/probe_leadbias.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r04_lb_stats_$v/**/*kernel_stats.csv",recursive=True)[0]
print("== $v")
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"])>2e5: print(f'{float(r["AverageNs"])/1e3:9.1f} us x {r["Calls"]:>4}  {r["Name"][:90]}')
PY
done
