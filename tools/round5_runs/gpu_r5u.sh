#!/bin/bash
# r5u: 20x20 bins (a wave owns ONE bin): every wave decides its own bin, no deciding wave, no workgroup barriers
# (-DBPP_AB_LOCAL_DECIDE) against the product, interleaved; parity subset first; mask / reset kernels beside it
set -u
export TMPDIR=/tmp
TAG=${1:-r5u}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_local.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or every_bin or fused or rotating or tall" > $O/pytest_local.log 2>&1; tail -n 1 $O/pytest_local.log
AB_ARGS="--no-cpu-baseline --only-headline --no-parity --steps 300 --warmup 50 --gpu-seconds 0.8"
for v in product local product local; do
  if [ "$v" = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "20:--size 20 20 20 --envs 32768 --pool 2048" "20rot:--size 20 20 20 --envs 32768 --pool 2048 --rotation" "20x20x10:--size 20 20 10 --envs 32768 --pool-file tests/golden/cut2_dataset_4bins_20x20x10.npz"; do
    name=${cfg%%:*}; args=${cfg#*:}
    python bench.py $AB_ARGS $args > $O/ab_${v}_$name.json 2>> $O/ab.err
    python - <<PY
import json
try:
    d = json.load(open("$O/ab_${v}_$name.json")); r = d["roofline"]
    print("%-10s %-9s kernel %.2f us (frac %.3f)  past L3 %.2f us (frac %.3f)  value %.1f M" % ("$v", "$name", r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0, d["value"] / 1e6))
except Exception as e:
    print("$v $name failed", e)
PY
  done
done
for v in product local; do
  if [ "$v" = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  python tools/bench_mask_kernels.py > $O/mask_kernels_$v.json 2>> $O/ab.err; python -c "
import json; d=json.load(open('$O/mask_kernels_$v.json'))
for k,v in d.items():
    if isinstance(v, dict) and k.startswith('20'): print('$v', k, {n: x['us'] for n,x in v.items()})"
done
tail -n 2 $O/ab.err
