#!/bin/bash
# counter generator, overlapped schedule: ring depth / refill interval sweep
set -u
export TMPDIR=/tmp
TAG=${1:-r4g}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for cfg in "64 30" "128 62" "256 126" "512 254"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --stream --stream-rng counter --stream-depth $1 --stream-refill $2 --gpu-seconds 1.5 --warmup 300 > $O/bench_ctr_d$1.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_ctr_d$1.json').readline()); print('counter depth $1 refill $2: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
python bench.py --no-cpu-baseline --stream --stream-rng mt19937 --stream-depth 128 --stream-refill 62 --gpu-seconds 1.5 --warmup 300 > $O/bench_mt_d128.json 2>> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_mt_d128.json').readline()); print('mt19937 depth 128 refill 62: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
tail -3 $O/bench.err
