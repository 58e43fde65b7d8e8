#!/bin/bash
# side stream of the refill confined to a CU mask (experiment build libbpp_hip_cumask.so), counter generator
set -u
export TMPDIR=/tmp
TAG=${1:-r4w}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_cumask.so
for m in 0 32 64 96 128 192; do
  for sp in 1 0; do
    [ $m = 0 ] && [ $sp = 0 ] && continue
    BPP_EXP_SIDE_CU_MASK=$m BPP_EXP_SIDE_CU_SPREAD=$sp python bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.6 > $O/bench_m${m}_s$sp.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_m${m}_s$sp.json').readline()); print('mask $m spread $sp: %.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"
  done
done
