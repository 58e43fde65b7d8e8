#!/bin/bash
# r5a: first GPU call of round 5 -- (1) the GPU suite on the new build (ABI v14, slot table, fill fast path, bench gate
# tests); (2) the default bench line as the driver runs it; (3) A/B: round-4 forms (base) / slot table only (e1) /
# slot table + fill fast path (product) on the three single-GPU configs + the mask / reset kernels per variant.
set -u
export TMPDIR=/tmp
TAG=${1:-r5a}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -n 5 $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_line.json 2> $O/bench_driver_line.err
tail -n 4 $O/bench_driver_line.err
python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/bench_driver_line.json") if l.startswith("{")][0]); r = d["roofline"]
    print("headline: %.1f M (%.1f M past L3) kernel %.2f / %.2f us frac %.3f / %.3f" % (d["value"]/1e6, d["value_past_l3"]/1e6, r["launch_us"], r["launch_us_past_l3"], r["frac"], r["frac_past_l3"]))
    for n, c in d.get("configs", {}).items():
        r = c["roofline"]; print("%s: %.1f M (%.1f M past L3) kernel %.2f / %.2f us frac %.3f / %.3f parity %s" % (n, c["value"]/1e6, c["value_past_l3"]/1e6, r["launch_us"], r["launch_us_past_l3"], r["frac"], r["frac_past_l3"], c["parity"]["mismatches"]))
    print("parity", {k: v for k, v in d["parity"].items() if k != "per_workload"}); print("eps", d.get("epsilon_variant"))
    c = d["cpu_baseline"]; print("cpu_baseline", c["kind"], c["value"], c["cores"])
except Exception as e:
    print("driver line failed", repr(e))
PY
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
    print("%-8s %-6s kernel %.2f us (frac %.3f)  past L3 %.2f us (frac %.3f)  value %.1f M" % ("$v", "$name", r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0, d["value"] / 1e6))
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
tail -n 3 $O/ab.err
