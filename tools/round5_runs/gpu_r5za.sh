#!/bin/bash
# r5za: what of the rows kernel costs the lock-steps beside it -- variants that leave out the terminator padding (`nopad`) or every
# store into the ring (`nostores`; results wrong, timing only), and the refills between the lock-steps instead of beside them
set -u
export TMPDIR=/tmp
TAG=${1:-r5za}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in product nopad nostores; do
  if [ $v = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_${name}_$v.json 2>> $O/bench.err
  done
done
unset BPP_HIP_LIB
for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30"; do
  name=${cfg%%:*}; args=${cfg#*:}
  BPP_STREAM_OVERLAP=0 timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_${name}_serial.json 2>> $O/bench.err
done
export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_nostores.so
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.8 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_d32_r14_nostores.csv 2>/dev/null; rm -rf $O/prof
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
for f in $O/kernel_stats_*.csv; do echo $f; head -4 $f | cut -d, -f2-; done
