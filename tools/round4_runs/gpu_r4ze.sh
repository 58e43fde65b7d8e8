#!/bin/bash
# experiment build: every wave of the row-cache step kernel at s_setprio(2) (the refill kernels' waves run at 0)
set -u
export TMPDIR=/tmp
TAG=${1:-r4ze}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for lib in libbpp_hip.so libbpp_hip_prio2.so; do
  for cfg in "counter:--stream-rng counter" "mt19937:"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/$lib python bench.py --no-cpu-baseline --stream --gpu-seconds 0.8 $args > $O/bench_${name}_$lib.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_$lib.json').readline()); print('$lib $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
