#!/bin/bash
# priority of the refill's side stream (experiment build): 0 = greatest (product), 1 = least, 2 = default
set -u
export TMPDIR=/tmp
TAG=${1:-r4zb}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_prio.so
for pr in 0 1 2; do
  for cfg in "counter:--stream-rng counter" "mt19937:"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_EXP_SIDE_PRIO=$pr python bench.py --no-cpu-baseline --stream --gpu-seconds 0.8 $args > $O/bench_${name}_prio$pr.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_prio$pr.json').readline()); print('prio $pr $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
