#!/bin/bash
# row cache: kernel statistics (serial / overlapped schedule) + whole-job stream benches, cache off / on
set -u
export TMPDIR=/tmp
TAG=${1:-r4z}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_supply.py tests/test_stream_counter.py -m gpu -x -q > $O/pytest_stream.log 2>&1; tail -2 $O/pytest_stream.log
for c in off on; do
  for cfg in "counter_10:--stream-rng counter" "mt19937_10:" "counter_20:--stream-rng counter --size 20 20 20 --envs 32768"; do
    name=${cfg%%:*}; args=${cfg#*:}
    python bench.py --no-cpu-baseline --stream --stream-cache $c --gpu-seconds 0.8 $args > $O/bench_stream_${name}_cache_$c.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_stream_${name}_cache_$c.json').readline()); print('stream $name cache $c: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
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
