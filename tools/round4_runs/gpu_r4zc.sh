#!/bin/bash
# kernel trace of the overlapped counter-stream run (row cache on): which refill kernel the step kernel pays for
set -u
export TMPDIR=/tmp
TAG=${1:-r4zc}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.15 > /dev/null 2>&1)
ls $O/prof; python - <<PY
import csv,glob,collections
f=glob.glob("$O/prof/*kernel_trace.csv")[0]
rows=[]
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    k="step" if "bpp_tile_kernel" in n and " 0, 4, 1>" in n else "cut" if "cut_ctr" in n else "sort" if "sort_kernel" in n else "scan" if "scan_kernel" in n else "other"
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),k))
rows.sort()
ref=[(s,e,k) for s,e,k in rows if k in ("cut","sort","scan") and e-s>20000]
steps=[(s,e) for s,e,k in rows if k=="step"]
steps=steps[len(steps)//3:]           # steady state
agg=collections.defaultdict(list)
for s,e in steps:
    ov=collections.Counter()
    for rs,re,k in ref:
        o=min(e,re)-max(s,rs)
        if o>0: ov[k]+=o
    d=e-s
    tag="alone" if not ov else "+".join(sorted(k for k in ov if ov[k]>0.3*d)) or "edge"
    agg[tag].append(d)
for k,v in sorted(agg.items()): print("%-10s n=%5d avg %.2f us" % (k,len(v),sum(v)/len(v)/1e3))
for k in ("cut","sort"):
    v=[e-s for s,e,kk in ref if kk==k]; print(k,"kernels n=%d avg %.1f us"%(len(v),sum(v)/max(1,len(v))/1e3))
PY
cp $O/prof/*kernel_trace.csv $O/kernel_trace.csv; rm -rf $O/prof
