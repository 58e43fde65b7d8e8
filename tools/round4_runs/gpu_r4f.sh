#!/bin/bash
# step-kernel duration in ring mode vs ring depth (footprint), serial schedule, counter generator
set -u
export TMPDIR=/tmp
TAG=${1:-r4f}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for cfg in "8 5" "16 13" "32 14" "64 30" "128 62"; do
  set -- $cfg
  BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --stream-depth $1 --stream-refill $2 --gpu-seconds 0.4 > $O/bench_serial_d$1.json 2>/dev/null
  echo "== depth $1 refill $2"
  python - <<PY
import csv
for r in csv.DictReader(open("$O/prof/run_kernel_stats.csv")):
    n = r["Name"]
    for key in ("bpp_tile_kernel", "cut_ctr", "sort_kernel", "scan_kernel"):
        if key in n and int(r["Calls"]) > 5:
            print("   %-16s calls %6s avg %9.1f ns  min %8s max %9s" % (key, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  rm -rf $O/prof
done > $O/summary.txt 2>&1
cat $O/summary.txt
