#!/bin/bash
# grid of the refill's sort kernel (experiment build): 2048 workgroups (product) vs fewer
set -u
export TMPDIR=/tmp
TAG=${1:-r4zd}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_sortgrid.so
for g in 2048 1024 512 256 128; do
  for cfg in "counter:--stream-rng counter" "mt19937:"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_EXP_SORT_GRID=$g python bench.py --no-cpu-baseline --stream --gpu-seconds 0.8 $args > $O/bench_${name}_g$g.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_g$g.json').readline()); print('sort grid $g $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
