#!/bin/bash
# idle-wave prefetch of the look-ahead lines (experiment build of the product tree: SGPR count above the 8-workgroup limit)
set -u
export TMPDIR=/tmp
TAG=${1:-r4m}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for P in 8192 1048576; do
  python bench.py --no-cpu-baseline --no-past-l3 --pool $P --gpu-seconds 0.4 > $O/bench_pool_$P.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_pool_$P.json').readline()); r=d['roofline']; print('pool $P: kernel %.2f us' % (r['launch_us']))"
done
for cfg in "ctr_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "mt_d64_r30:--stream-depth 64 --stream-refill 30"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
