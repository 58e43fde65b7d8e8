#!/bin/bash
# r5g: (1) A/B on ONE box: round-4 forms (base) / slot table only (e1) / everything incl. the sparse-EXEC fixes at 78 scalar
# registers (product); (2) drop-in step() with eager_infos; (3) GPU suite subset for the changed paths
set -u
export TMPDIR=/tmp
TAG=${1:-r5g}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or every_bin or fused or rotating or dropin or epsilon" > $O/pytest_subset.log 2>&1; tail -n 2 $O/pytest_subset.log
AB_ARGS="--no-cpu-baseline --only-headline --no-parity --steps 300 --warmup 50 --gpu-seconds 1.2"
for round in 1 2; do
for v in base e1 product; do
  if [ "$v" = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
    name=${cfg%%:*}; args=${cfg#*:}
    python bench.py $AB_ARGS $args > $O/ab${round}_${v}_$name.json 2>> $O/ab.err
    python - <<PY
import json
try:
    d = json.load(open("$O/ab${round}_${v}_$name.json")); r = d["roofline"]
    print("%-10s %-6s kernel %.2f us (frac %.3f)  past L3 %.2f us (frac %.3f)  value %.1f M" % ("$v", "$name", r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0, d["value"] / 1e6))
except Exception as e:
    print("$v $name failed", e)
PY
  done
  if [ $round = 1 ]; then python tools/bench_mask_kernels.py > $O/mask_kernels_$v.json 2>> $O/ab.err; python -c "
import json; d=json.load(open('$O/mask_kernels_$v.json'))
for k,v in d.items(): print('$v', k, {n: x['us'] for n,x in v.items()})"; fi
done
done
unset BPP_HIP_LIB
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/dropin.err; python -c "
import json; d=json.load(open('$O/dropin_step.json')); [print(k, v) for k, v in d.items() if k != 'note']"; tail -n 2 $O/dropin.err
tail -n 3 $O/ab.err
