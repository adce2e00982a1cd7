"""GPU busy fraction of the second half of a rocprofv3 --kernel-trace CSV (union of the kernel intervals / wall window):
    python tools/busy_fraction.py <dir>/t_kernel_trace.csv
Round 4, one box: ConvLSTM 0.961, MetNet 0.972, CloudGAN 0.736 (host-bound under the profiler)."""
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
n=len(ev); lo=ev[n//2][0]; hi=ev[-1][1]
busy=0; cur_s=None; cur_e=None; cnt=0
for s,e in ev:
    if s<lo: continue
    cnt+=1
    if cur_e is None or s>cur_e:
        if cur_e is not None: busy+=cur_e-cur_s
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print(sys.argv[1].split('/')[-2], 'window ms', (hi-lo)/1e6, 'busy frac', round(busy/(hi-lo),4), 'kernels', cnt)
