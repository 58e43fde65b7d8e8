#!/bin/bash
# r5v: the rows pipeline of the counter generator (one lane per sequence, ranked in the lane, redo kernel): stream tests, benches of
# the counter configs with BPP_STREAM_LEGACY=0 (rows) and =2 (round 4's cut per bin + sort), kernel stats overlapped and serial.
set -u
export TMPDIR=/tmp
TAG=${1:-r5v}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_counter.py tests/test_stream_supply.py -m gpu -q -x ) > $O/pytest_stream.log 2>&1
tail -3 $O/pytest_stream.log
for leg in ${LEGS:-0 2}; do
  for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" \
             "counter_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_STREAM_LEGACY=$leg timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_${name}_legacy$leg.json 2>> $O/bench.err
  done
  (cd /tmp && BPP_STREAM_LEGACY=$leg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.8 > /dev/null 2>&1)
  cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_d32_r14_legacy$leg.csv 2>/dev/null; rm -rf $O/prof
  (cd /tmp && BPP_STREAM_LEGACY=$leg BPP_STREAM_OVERLAP=0 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.5 > /dev/null 2>&1)
  cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_d32_r14_serial_legacy$leg.csv 2>/dev/null; rm -rf $O/prof
done
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
for f in $O/kernel_stats_*.csv; do echo $f; head -6 $f | cut -c1-150; done
