#!/bin/bash
# r5p: candidates spread evenly over the passes (72 candidates: 36 + 36 instead of 64 + 8 -- a pass of <= 8 lanes is the slow case of
# tools/ubench sparse) ALONE, same register cap as the product: -DBPP_AB_EVEN_PASSES against the product, interleaved
set -u
export TMPDIR=/tmp
TAG=${1:-r5p}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_even.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or every_bin or fused or rotating" > $O/pytest_even.log 2>&1; tail -n 1 $O/pytest_even.log
AB_ARGS="--no-cpu-baseline --only-headline --no-parity --steps 300 --warmup 50 --gpu-seconds 0.8"
for v in product even product even; do
  if [ "$v" = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
    name=${cfg%%:*}; args=${cfg#*:}
    python bench.py $AB_ARGS $args > $O/ab_${v}_$name.json 2>> $O/ab.err
    python - <<PY
import json
try:
    d = json.load(open("$O/ab_${v}_$name.json")); r = d["roofline"]
    print("%-10s %-6s kernel %.2f us (frac %.3f)  past L3 %.2f us (frac %.3f)  value %.1f M" % ("$v", "$name", r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0, d["value"] / 1e6))
except Exception as e:
    print("$v $name failed", e)
PY
  done
done
