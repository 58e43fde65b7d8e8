#!/bin/bash
# A/B of library variants (tools/build_variant.sh): step-kernel launch time of the three single-GPU configs per variant.
# usage: tools/gpu_ab.sh <tag> <variant> [<variant> ...]      ("product" = csrc/libbpp_hip.so)
set -u
export TMPDIR=/tmp
TAG=$1; shift
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in "$@"; do
  if [ "$v" = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048" "stream:--stream" "stream20:--stream --size 20 20 20 --envs 32768"; do
    case " ${AB_ONLY:-10 10rot 20} " in *" ${cfg%%:*} "*) ;; *) continue;; esac
    name=${cfg%%:*}; args=${cfg#*:}
    python bench.py --no-cpu-baseline --only-headline --no-parity --steps 300 --warmup 50 --gpu-seconds 1.5 $args > $O/ab_${v}_$name.json 2>> $O/ab.err
    python - <<PY
import json
try:
    d = json.load(open("$O/ab_${v}_$name.json")); r = d["roofline"]
    print("%-12s %-8s kernel %.2f us (frac %.3f)  past L3 %.2f us (frac %.3f)  value %.1f M  %.2f us/lock-step" % ("$v", "$name", r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0, d["value"] / 1e6, d["ms_per_step"] * 1e3))
except Exception as e:
    print("$v $name failed", e)
PY
  done
done
tail -n 2 $O/ab.err
