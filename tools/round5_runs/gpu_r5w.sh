#!/bin/bash
# r5w: rows pipeline with packed level counters (11 LDS granules per wave: two waves beside seven step workgroups); variants with
# other list / staging capacities: B 14 / 28 (7 granules, many second attempts), C 20 / 40 (10), D 32 / 64 (14, next to none)
set -u
export TMPDIR=/tmp
TAG=${1:-r5w}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_counter.py tests/test_stream_supply.py -m gpu -q -x ) > $O/pytest_stream.log 2>&1
tail -3 $O/pytest_stream.log
for v in product rowsB rowsC rowsD; do
  if [ $v = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_${name}_$v.json 2>> $O/bench.err
  done
done
unset BPP_HIP_LIB
timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 --stream-rng counter --rotation --stream-depth 64 --stream-refill 30 > $O/bench_stream_counter_rot_d64_r30_product.json 2>> $O/bench.err
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.8 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_d32_r14.csv 2>/dev/null; rm -rf $O/prof
(cd /tmp && BPP_STREAM_OVERLAP=0 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.5 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_d32_r14_serial.csv 2>/dev/null; rm -rf $O/prof
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
for f in $O/kernel_stats_*.csv; do echo $f; head -4 $f | cut -c1-150; done
