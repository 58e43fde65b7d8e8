#!/bin/bash
# row cache (bpp_batch.seq_cache): stream tests, whole-job stream benches cache off / on, kernel statistics (serial and
# overlapped schedule), pool-mode regression check
set -u
export TMPDIR=/tmp
TAG=${1:-r4y}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_supply.py tests/test_stream_counter.py -m gpu -x -q > $O/pytest_stream.log 2>&1; tail -3 $O/pytest_stream.log
for cfg in "10:" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --gpu-seconds 0.6 $args > $O/bench_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').readline()); r=d['roofline']; print('$name: %.1f M env steps/s, kernel %.2f us, past L3 %.2f us' % (d['value']/1e6, r['launch_us'], r['launch_us_past_l3']))"
done
for c in off on; do
for cfg in "counter_10:--stream-rng counter" "mt19937_10:" "counter_rot:--stream-rng counter --rotation" \
           "counter_20:--stream-rng counter --size 20 20 20 --envs 32768" "mt19937_20:--size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --stream-cache $c --gpu-seconds 0.8 $args > $O/bench_stream_${name}_cache_$c.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_${name}_cache_$c.json').readline()); print('stream $name cache $c: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
done
for c in off on; do
  (cd /tmp && BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --stream-cache $c --gpu-seconds 0.4 > /dev/null 2>&1)
  cp $O/prof_$c/run_kernel_stats.csv $O/kernel_stats_stream_counter_cache_${c}_serial_schedule.csv 2>/dev/null; rm -rf $O/prof_$c
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --stream-cache $c --gpu-seconds 0.4 > /dev/null 2>&1)
  cp $O/prof_$c/run_kernel_stats.csv $O/kernel_stats_stream_counter_cache_${c}.csv 2>/dev/null; rm -rf $O/prof_$c
  echo "== cache $c: step kernel, serial schedule / beside the refills"
  grep "bpp_tile_kernel.*<10, 10, 1, false, 0, 4, 1>" $O/kernel_stats_stream_counter_cache_${c}_serial_schedule.csv | sed "s/.*Params)\",//"
  grep "bpp_tile_kernel.*<10, 10, 1, false, 0, 4, 1>" $O/kernel_stats_stream_counter_cache_${c}.csv | sed "s/.*Params)\",//"
done
