#!/bin/bash
# r5zd: deeper rings / rarer refills where the refill is narrow but long (20x20x20) and for the exact generator
set -u
export TMPDIR=/tmp
TAG=${1:-r5zd}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for cfg in "counter_20_d64_r30:--stream-rng counter --size 20 20 20 --envs 32768 --stream-depth 64 --stream-refill 30" \
           "counter_20_d128_r60:--stream-rng counter --size 20 20 20 --envs 32768 --stream-depth 128 --stream-refill 60" \
           "mt19937_d128_r60:--stream-depth 128 --stream-refill 60" \
           "mt19937_20_d64_r30:--size 20 20 20 --envs 32768 --stream-depth 64 --stream-refill 30" \
           "counter_rot_d128_r60:--stream-rng counter --rotation --stream-depth 128 --stream-refill 60"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 200 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_$name.json 2>> $O/bench.err
done
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
grep -v amdgpu.ids $O/bench.err | tail -5
