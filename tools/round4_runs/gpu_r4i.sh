#!/bin/bash
# early issue of the look-ahead pool loads: headline pool, 1 M-row pool, stream (counter / mt), rotation, 20^3
set -u
export TMPDIR=/tmp
TAG=${1:-r4i}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for cfg in "pool8192:--pool 8192" "pool1M:--pool 1048576 --no-past-l3" "rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --gpu-seconds 0.6 $args > $O/bench_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').readline()); r=d['roofline']; print('$name: %.1f M env steps/s, kernel %.2f us (b2b %.2f), past L3 %s us' % (d['value']/1e6, r['launch_us'], r['launch_us_back_to_back'], r['launch_us_past_l3']))"
done
for cfg in "ctr_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "mt_d32_r14:"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
