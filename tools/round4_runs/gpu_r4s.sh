#!/bin/bash
# head table (bpp_batch.seq_head) + amdgpu_num_sgpr(80) on the tile kernel: stream tests, stream benches with kernel
# statistics (serial schedule: the step kernel's own time in ring mode), and the three pool-mode configs for regressions
set -u
export TMPDIR=/tmp
TAG=${1:-r4s}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_supply.py tests/test_stream_counter.py -m gpu -x -q > $O/pytest_stream.log 2>&1; tail -3 $O/pytest_stream.log
for cfg in "10:" "rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --gpu-seconds 0.8 $args > $O/bench_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').readline()); r=d['roofline']; print('$name: %.1f M env steps/s, kernel %.2f us, past L3 %.2f us' % (d['value']/1e6, r['launch_us'], r['launch_us_past_l3']))"
done
for cfg in "mt19937_d32_r14:" "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" \
           "counter_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30" \
           "mt19937_20_d32_r14:--size 20 20 20 --envs 32768" "counter_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
for g in counter; do
  (cd /tmp && BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream_$g -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng $g --gpu-seconds 0.5 > /dev/null 2>&1)
  cp $O/prof_stream_$g/run_kernel_stats.csv $O/kernel_stats_stream_${g}_d32_r14_serial_schedule.csv 2>/dev/null; rm -rf $O/prof_stream_$g
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream_$g -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng $g --gpu-seconds 0.5 > /dev/null 2>&1)
  cp $O/prof_stream_$g/run_kernel_stats.csv $O/kernel_stats_stream_${g}_d32_r14.csv 2>/dev/null; rm -rf $O/prof_stream_$g
done
head -8 $O/kernel_stats_stream_counter_d32_r14_serial_schedule.csv | cut -c1-200
head -8 $O/kernel_stats_stream_counter_d32_r14.csv | cut -c1-200
