#!/bin/bash
# r5y: rows pipeline, two rows per lane one after the other (a wave is as slow as its slowest lane; the sum of two sequences varies
# less than one) against one row per lane (`one` = the build of commit afc7d1b)
set -u
export TMPDIR=/tmp
TAG=${1:-r5y}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_counter.py -m gpu -q -x ) > $O/pytest_stream.log 2>&1
tail -3 $O/pytest_stream.log
for rep in 1 2; do
for v in product one; do
  if [ $v = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_${name}_${v}_$rep.json 2>> $O/bench.err
  done
done
done
unset BPP_HIP_LIB
(cd /tmp && BPP_STREAM_OVERLAP=0 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.5 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_d32_r14_serial.csv 2>/dev/null; rm -rf $O/prof
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
for f in $O/kernel_stats_*.csv; do echo $f; head -4 $f | cut -d, -f2-; done
