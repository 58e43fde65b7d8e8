#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/${1:-ps}; mkdir -p $O; cd /tmp
for cfg in "16 6" "32 14" "64 30"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$1 -o run -- python $R/bench.py --no-cpu-baseline --stream --stream-depth $1 --stream-refill $2 --steps 600 --warmup 100 > $O/bench_$1.json 2>/dev/null
  python - <<PY
import csv, json
d = json.load(open("$O/bench_$1.json"))
print("depth $1 refill $2: %.1f M steps/s %.2f us/step" % (d["value"]/1e6, d["ms_per_step"]*1e3))
for r in list(csv.DictReader(open("$O/p_$1/run_kernel_stats.csv")))[:6]:
    print("   %-40s calls %6s avg %9.1f us  total %9.1f ms" % (r["Name"].replace("(anonymous namespace)::","")[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"])/1e6))
PY
  cp $O/p_$1/run_kernel_stats.csv $O/kernel_stats_d$1.csv; rm -rf $O/p_$1
done
