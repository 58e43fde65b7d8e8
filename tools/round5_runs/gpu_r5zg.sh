#!/bin/bash
# r5zg: the stream-supply bench lines of tools/gpu_final.sh again, on the final tree (after r5zf) -> profiles/r5_bench_stream_*
set -u
export TMPDIR=/tmp
TAG=${1:-r5zg}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for cfg in "mt19937_d32_r14:" "mt19937_d64_r30:--stream-depth 64 --stream-refill 30" "counter_d32_r14:--stream-rng counter" \
           "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" \
           "counter_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30" \
           "counter_d128_r60:--stream-rng counter --stream-depth 128 --stream-refill 60" \
           "mt19937_20_d32_r14:--size 20 20 20 --envs 32768" "counter_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 100 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_$name.json 2>> $O/bench.err
done
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
