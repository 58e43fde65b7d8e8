#!/bin/bash
# round 4, fourth call: counter-based stream supply (tests, benches, kernel stats), drop-in step with the SoA gather
set -u
export TMPDIR=/tmp
TAG=${1:-r4d}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_stream_counter.py tests/test_stream_supply.py tests/test_gpu_parity.py -m gpu -q -x -k "counter or stream or finished_infos or dropin" ) > $O/pytest_stream.log 2>&1
tail -4 $O/pytest_stream.log
for cfg in "ctr_d32_r14:--stream-rng counter" "ctr_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" "ctr_d16_r6:--stream-rng counter --stream-depth 16 --stream-refill 6" "ctr_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768" "ctr_rot_d32_r14:--stream-rng counter --rotation" "mt_d32_r14:"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 1.0 > $O/bench_stream_ctr_under_rocprof.json 2>/dev/null)
cp $O/prof_stream/run_kernel_stats.csv $O/kernel_stats_stream_ctr_d32_r14.csv 2>/dev/null; rm -rf $O/prof_stream
cut -c1-200 $O/kernel_stats_stream_ctr_d32_r14.csv | head -8
python tools/bench_dropin_step.py --steps 200 > $O/dropin_step.json 2> $O/dropin.err; python -c "
import json; d=json.load(open('$O/dropin_step.json')); print({k: v for k, v in d.items() if k != 'note'})"
