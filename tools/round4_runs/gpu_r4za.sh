#!/bin/bash
# row cache: the new GPU tests (cache on == cache off through kernel switches, clones, restores; stream specs with a cache),
# stream soak with both generators, stream benches
set -u
export TMPDIR=/tmp
TAG=${1:-r4za}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_stream_supply.py tests/test_stream_counter.py -m gpu -x -q ) > $O/pytest_stream.log 2>&1; tail -6 $O/pytest_stream.log
( time timeout 900 python tools/soak_stream.py ) > $O/soak_stream.log 2>&1; tail -4 $O/soak_stream.log
for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "counter_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
(cd /tmp && BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.4 > /dev/null 2>&1)
grep "bpp_tile_kernel.*<10, 10, 1, false, 0, 4, 1>" $O/prof/run_kernel_stats.csv | sed "s/.*Params)\",//"; rm -rf $O/prof
