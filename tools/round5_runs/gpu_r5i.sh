#!/bin/bash
# r5i: (1) FETCH_SIZE calibration in the step kernel's own access widths (tools/ubench calib incl. the narrow reads);
# (2) the slow paths quantified: geometries the tile / prefix-image kernels reject (10x10x30: H > 22; 7x13x8: W*L % 4 != 0)
# and the reference's own 20x20x10 set (dataset/4bins_cut_2.pt) on the tile kernel
set -u
export TMPDIR=/tmp
TAG=${1:-r5i}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib_fetch -o p -- $R/tools/ubench calib > $O/ubench_calib.jsonl 2> $O/calib.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/calib_write -o p -- $R/tools/ubench calib > /dev/null 2>> $O/calib.err
cat $O/ubench_calib.jsonl
python - <<PY
import csv, glob, collections
for what in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/calib_%s/**/*counter_collection.csv" % what, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(what, k[0], k[1], "KiB per launch:", [round(v) for v in acc[k]][:6])
PY
cd $R
for cfg in "generic_10x10x30:--size 10 10 30 --envs 65536" "generic_7x13x8:--size 7 13 8 --envs 65536" "generic_7x13x8_rot:--size 7 13 8 --envs 65536 --rotation" \
           "tile_20x20x10_dataset_4bins:--size 20 20 10 --envs 32768 --pool-file tests/golden/cut2_dataset_4bins_20x20x10.npz" \
           "prefix_rt_12x12x12:--size 12 12 12 --envs 65536"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --gpu-seconds 0.8 $args > $O/bench_$name.json 2>> $O/bench.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d["roofline"]
    print("%-30s %s: %.1f M env steps/s (%.1f M past L3), kernel %.2f us frac %.3f / past L3 %s us frac %s; parity %s" % ("$name", r["kernel"], d["value"] / 1e6, (d["value_past_l3"] or 0) / 1e6, r["launch_us"], r["frac"], r["launch_us_past_l3"], r["frac_past_l3"], d["parity"]["mismatches"]))
except Exception as e:
    print("$name failed", repr(e))
PY
done
tail -n 3 $O/bench.err
