#!/bin/bash
# stand-alone durations of the refill kernels (no overlap with the lock-steps), counter and MT generators
set -u
export TMPDIR=/tmp
TAG=${1:-r4e}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for g in counter mt19937; do
  BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$g -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng $g --gpu-seconds 0.6 > $O/bench_serial_$g.json 2>/dev/null
  cp $O/prof_$g/run_kernel_stats.csv $O/kernel_stats_serial_$g.csv 2>/dev/null; rm -rf $O/prof_$g
  echo "== $g serial"; cut -c1-160 $O/kernel_stats_serial_$g.csv | head -7
  python -c "
import json; d=json.loads(open('$O/bench_serial_$g.json').readline()); print('serial $g: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
