#!/bin/bash
# experiment build libbpp_hip_ntq.so: nontemporal output stores also in the row-cache step kernel of the 10x10 bins (stream mode)
set -u
export TMPDIR=/tmp
TAG=${1:-r4zp}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for lib in libbpp_hip.so libbpp_hip_ntq.so; do
  for cfg in "counter_10:--stream-rng counter" "mt19937_10:" "counter_rot:--stream-rng counter --rotation"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/$lib python bench.py --no-cpu-baseline --stream --gpu-seconds 0.8 $args > $O/bench_${name}_$lib.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_$lib.json').readline()); print('$lib stream $name: %.1f M env steps/s' % (d['value']/1e6))"
  done
done
python tools/bench_mask_kernels.py > $O/mask_and_reset_kernels.json 2>> $O/bench.err
python -c "
import json; d=json.load(open('$O/mask_and_reset_kernels.json'))
for k,v in d.items(): print(k, {n: x['us'] for n,x in v.items()})"
