#!/bin/bash
# is the ring-mode step-kernel penalty the random HBM reads of the look-ahead pool entries?  static pool far beyond L2 / MALL
set -u
export TMPDIR=/tmp
TAG=${1:-r4h}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for P in 8192 131072 1048576; do
  ( time python bench.py --no-cpu-baseline --no-past-l3 --pool $P --gpu-seconds 0.5 > $O/bench_pool_$P.json ) 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_pool_$P.json').readline()); r=d['roofline']; print('pool $P rows: %.1f M env steps/s, kernel %.2f us (b2b %.2f)' % (d['value']/1e6, r['launch_us'], r['launch_us_back_to_back']))"
done
grep real $O/bench.err
