#!/bin/bash
# LATE look-ahead patch: A/B on a 1 M-row pool and on streams (BPP_LOOK_AHEAD_LATE=2 turns it off), headline unchanged
set -u
export TMPDIR=/tmp
TAG=${1:-r4k}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --gpu-seconds 0.6 > $O/bench_pool8192.json 2>> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_pool8192.json').readline()); r=d['roofline']; print('pool8192: %.1f M env steps/s, kernel %.2f us, past L3 %s us' % (d['value']/1e6, r['launch_us'], r['launch_us_past_l3']))"
for late in 0 2; do
  BPP_LOOK_AHEAD_LATE=$late python bench.py --no-cpu-baseline --no-past-l3 --pool 1048576 --gpu-seconds 0.5 > $O/bench_pool1M_late$late.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_pool1M_late$late.json').readline()); r=d['roofline']; print('pool1M late=$late: %.1f M env steps/s, kernel %.2f us' % (d['value']/1e6, r['launch_us']))"
  for cfg in "ctr_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "mt_d64_r30:--stream-depth 64 --stream-refill 30" "ctr_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768" "ctr_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_LOOK_AHEAD_LATE=$late python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_${name}_late$late.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_stream_${name}_late$late.json').readline()); print('stream $name late=$late: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or random_rollout or full_size" ) > $O/pytest_parity.log 2>&1
grep -E "passed|failed" $O/pytest_parity.log | tail -2
