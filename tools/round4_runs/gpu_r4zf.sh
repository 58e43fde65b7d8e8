#!/bin/bash
# row cache on / off at 20x20x20 (32 768 bins), both generators, three repetitions each
set -u
export TMPDIR=/tmp
TAG=${1:-r4zf}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2 3; do
for c in off on; do
  for cfg in "counter_20:--stream-rng counter --size 20 20 20 --envs 32768" "mt19937_20:--size 20 20 20 --envs 32768"; do
    name=${cfg%%:*}; args=${cfg#*:}
    python bench.py --no-cpu-baseline --stream --stream-cache $c --gpu-seconds 0.8 $args > $O/bench_${name}_${c}_$rep.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_${c}_$rep.json').readline()); print('$name cache $c rep $rep: %.1f M env steps/s' % (d['value']/1e6))"
  done
done
done
