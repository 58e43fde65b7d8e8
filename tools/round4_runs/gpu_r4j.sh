#!/bin/bash
# ring rows with look-ahead entries: stream tests (both generators), headline regression, stream benches, kernel stats
set -u
export TMPDIR=/tmp
TAG=${1:-r4j}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_stream_counter.py tests/test_stream_supply.py tests/test_lookahead.py -m gpu -q -x ) > $O/pytest_stream.log 2>&1
tail -4 $O/pytest_stream.log
python bench.py --no-cpu-baseline --gpu-seconds 0.6 > $O/bench_pool8192.json 2>> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_pool8192.json').readline()); r=d['roofline']; print('pool8192: %.1f M env steps/s, kernel %.2f us (b2b %.2f), past L3 %s us' % (d['value']/1e6, r['launch_us'], r['launch_us_back_to_back'], r['launch_us_past_l3']))"
for cfg in "ctr_d32_r14:--stream-rng counter" "ctr_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "mt_d32_r14:" "mt_d64_r30:--stream-depth 64 --stream-refill 30" "ctr_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768" "ctr_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
cd /tmp
for g in counter; do
  BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$g -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng $g --gpu-seconds 0.5 > $O/bench_serial_$g.json 2>/dev/null
  cp $O/prof_$g/run_kernel_stats.csv $O/kernel_stats_serial_$g.csv 2>/dev/null; rm -rf $O/prof_$g
  echo "== $g serial"; cut -c1-150 $O/kernel_stats_serial_$g.csv | head -5
done
