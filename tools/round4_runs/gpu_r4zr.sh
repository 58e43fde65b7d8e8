#!/bin/bash
# counter generator: the cut waves sort their own rows (no sort kernel): stream tests, stress, whole-job benches, kernel statistics
set -u
export TMPDIR=/tmp
TAG=${1:-r4zr}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_stream_counter.py -m gpu -x -q > $O/pytest_stream.log 2>&1; tail -2 $O/pytest_stream.log
timeout 300 python tools/stress_stream.py --trials 150 --seed 21 > $O/stress.log 2>&1; tail -1 $O/stress.log
for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "counter_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30" "counter_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.4 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats_stream_counter_d32_r14.csv; rm -rf $O/prof; head -6 $O/kernel_stats_stream_counter_d32_r14.csv | cut -c1-170
