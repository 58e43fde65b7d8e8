#!/bin/bash
# experiment builds: sort grid 2048 -> 8192 workgroups, counter cut kernel at raised priority
set -u
export TMPDIR=/tmp
TAG=${1:-r4q}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in product sort8k prio both; do
  lib=""; [ $v != product ] && lib=$R/tools/libbpp_exp_$v.so
  for cfg in "ctr_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "mt_d64_r30:--stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_HIP_LIB=$lib python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_${v}_$name.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${v}_$name.json').readline()); print('$v $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
