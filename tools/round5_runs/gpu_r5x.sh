#!/bin/bash
# r5x: rows pipeline, the cut_rows grid bounded to 256 / 512 / 1024 single-wave workgroups (each walks through its share of the
# rows) against one wave per 64 rows; and a deeper ring (D = 128, refill every 60 lock-steps)
set -u
export TMPDIR=/tmp
TAG=${1:-r5x}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in product grid256 grid512 grid1024; do
  if [ $v = product ]; then unset BPP_HIP_LIB; else export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so; fi
  for cfg in "counter_d32_r14:--stream-rng counter" "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_${name}_$v.json 2>> $O/bench.err
  done
done
unset BPP_HIP_LIB
timeout 120 python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 --stream-rng counter --stream-depth 128 --stream-refill 60 > $O/bench_stream_counter_d128_r60_product.json 2>> $O/bench.err
export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_grid512.so
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.8 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_d32_r14_grid512.csv 2>/dev/null; rm -rf $O/prof
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
for f in $O/kernel_stats_*.csv; do echo $f; head -4 $f | cut -d, -f2-; done
