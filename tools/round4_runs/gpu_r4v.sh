#!/bin/bash
# groups of bins per wave (BPP_TILE_GROUPS) in ring mode: whole-job stream throughput, default (1) vs 2 vs 4; pool mode for reference
set -u
export TMPDIR=/tmp
TAG=${1:-r4v}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for g in 0 2 4; do
  for cfg in "counter_10:--stream --stream-rng counter" "mt_10:--stream" "counter_rot:--stream --stream-rng counter --rotation" \
             "counter_20:--stream --stream-rng counter --size 20 20 20 --envs 32768" "mt_20:--stream --size 20 20 20 --envs 32768" "pool_10:" "pool_20:--size 20 20 20 --envs 32768 --pool 2048"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_TILE_GROUPS=$g python bench.py --no-cpu-baseline --gpu-seconds 0.8 $args > $O/bench_${name}_g$g.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_g$g.json').readline()); print('groups $g $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
