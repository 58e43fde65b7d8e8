#!/bin/bash
# r5zf: bpp_rollout_uniform_stream without a sampler launch per chunk (the last step of a chunk draws the next chunk's first action)
set -u
export TMPDIR=/tmp
TAG=${1:-r5zf}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 400 python -m pytest tests/test_stream_supply.py tests/test_stream_counter.py -m gpu -q -x -k "rollout or overlap or side or stream_supply_matches or counter_supply" ) > $O/pytest_stream.log 2>&1
tail -3 $O/pytest_stream.log
for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" \
           "counter_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768" "mt19937_d64_r30:--stream-depth 64 --stream-refill 30"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.2 $args > $O/bench_stream_$name.json 2>> $O/bench.err
done
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
