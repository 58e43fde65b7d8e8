#!/bin/bash
# round 4, third call: stream supply with byte outputs / 16-bit lists (tests, bench, kernel stats), drop-in step with the
# compaction gather, store-bandwidth calibration
set -u
export TMPDIR=/tmp
TAG=${1:-r4c}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_stream_supply.py tests/test_gpu_parity.py -m gpu -q -x -k "stream or finished_infos or dropin" ) > $O/pytest_stream.log 2>&1
tail -4 $O/pytest_stream.log
for cfg in "d32_r14:" "d64_r30:--stream-depth 64 --stream-refill 30" "20_d32_r14:--size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 > $O/bench_stream_under_rocprof.json 2>/dev/null)
cp $O/prof_stream/run_kernel_stats.csv $O/kernel_stats_stream_d32_r14.csv 2>/dev/null; rm -rf $O/prof_stream
cut -c1-200 $O/kernel_stats_stream_d32_r14.csv | head -8
python tools/bench_dropin_step.py --steps 200 > $O/dropin_step.json 2> $O/dropin.err; cat $O/dropin_step.json | cut -c1-1200
if [ ! -x tools/ubench ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip 2> $O/ubench_build.err; fi
tools/ubench calib > $O/ubench_calib.jsonl 2>> $O/ubench_build.err; cat $O/ubench_calib.jsonl
