#!/bin/bash
# device scheduling flag (hipSetDeviceFlags) and the fences of a 20-step region: bench.py --steps 20
set -u
export TMPDIR=/tmp
TAG=${1:-r4zh}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for f in "" 1 2 4; do
  BPP_BENCH_SCHED=$f python bench.py --no-cpu-baseline --steps 20 --warmup 5 --gpu-seconds 1.0 --no-past-l3 > $O/bench_sched_$f.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_sched_$f.json').readline()); print('sched [$f]: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
tail -3 $O/bench.err
