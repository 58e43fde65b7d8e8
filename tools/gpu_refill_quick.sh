#!/bin/bash
# quick look at the refill pipeline: stream tests, refill latency, kernel stats, three stream benches
set -u
export TMPDIR=/tmp
TAG=${1:-rq}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_stream_supply.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_stream_refill.py --needs 1 2 4 2> $O/err.txt | tee $O/refill_10.json | cut -c1-40,330-
python tools/bench_stream_refill.py --needs 1 --frac 0.11 2>> $O/err.txt | tee -a $O/refill_10.json | cut -c1-40,330-
python tools/bench_stream_refill.py --needs 1 2 --size 20 20 20 --envs 32768 2>> $O/err.txt | tee $O/refill_20.json | cut -c1-40,330-
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/tools/bench_stream_refill.py --needs 1 2 4 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:6]:
    print("%-44s calls %5s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
for cfg in "8 5" "16 6" "32 14" "64 30"; do
  set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --stream --steps 600 --warmup 100 --stream-depth $1 --stream-refill $2 > $O/bench_d$1_r$2.json 2>> $O/err.txt
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_d$1_r$2.json"))
    print("depth %3d refill %3d  %8.1f M env steps/s, %7.2f us/lock-step" % ($1, $2, d["value"] / 1e6, d["ms_per_step"] * 1e3))
except Exception as e:
    print("failed", e)
PY
done
