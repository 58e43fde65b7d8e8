#!/bin/bash
# experiment build libbpp_hip_pf.so: a prefetch kernel on a third queue touches the first line of every bin's next ring row
# while the next lock-step runs (BPP_EXP_PREFETCH=1)
set -u
export TMPDIR=/tmp
TAG=${1:-r4x}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_pf.so
for cfg in "off:BPP_EXP_PREFETCH=0" "on:BPP_EXP_PREFETCH=1" "on_hiprio:BPP_EXP_PREFETCH=1 BPP_EXP_PREFETCH_PRIO=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.6 > $O/bench_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').readline()); print('prefetch $name: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  (cd /tmp && env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.4 > /dev/null 2>&1)
  cp $O/prof_$name/run_kernel_stats.csv $O/kernel_stats_$name.csv 2>/dev/null; rm -rf $O/prof_$name
  grep -E "bpp_tile_kernel.* 0, 4|prefetch" $O/kernel_stats_$name.csv | sed "s/.*Params)\",/step: /; s/.*unsigned int\*)\",/prefetch: /"
done
