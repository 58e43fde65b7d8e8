#!/bin/bash
# r5zc: rows kernel with a raised wave priority (s_setprio 3 / 1) so that its waves leave sooner
set -u
export TMPDIR=/tmp
TAG=${1:-r5zc}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2; do
for v in product prio3 prio1; do
  if [ $v = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.2 $args > $O/bench_stream_${name}_${v}_$rep.json 2>> $O/bench.err
  done
done
done
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
