#!/bin/bash
# r5k: TIMING experiment -- what would storing the observation BEFORE the decision buy (speculative store from the staged
# tile, afterwards only the placed window's rows / the failed bins' full rows)?  -DBPP_AB_EARLY_OBS build (its observations are wrong)
set -u
export TMPDIR=/tmp
TAG=${1:-r5k}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
AB_ARGS="--no-cpu-baseline --only-headline --no-parity --steps 300 --warmup 50 --gpu-seconds 0.8"
for v in product earlyobs product earlyobs; do
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
tail -n 2 $O/ab.err
