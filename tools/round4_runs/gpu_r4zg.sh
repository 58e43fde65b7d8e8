#!/bin/bash
# sort kernel with six rows in flight per wave: grid sweep (BPP_EXP_SORT_GRID), whole-job stream benches + kernel trace summary
set -u
export TMPDIR=/tmp
TAG=${1:-r4zg}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_stream_supply.py tests/test_stream_counter.py -m gpu -x -q -k "matches_oracle or interchangeable or counter" > $O/pytest_stream.log 2>&1; tail -2 $O/pytest_stream.log
for g in 2048 1024 640 512 384 256; do
  for cfg in "counter:--stream-rng counter" "mt19937:"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_EXP_SORT_GRID=$g python bench.py --no-cpu-baseline --stream --gpu-seconds 0.8 $args > $O/bench_${name}_g$g.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_g$g.json').readline()); print('sort grid $g $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
