#!/bin/bash
# experiment builds (tools/libbpp_exp_*.so, not the product): the step kernel on a 1 M-row pool with none / one of the
# three look-ahead loads -- is the +6 us of a pool beyond the caches those loads?
set -u
export TMPDIR=/tmp
TAG=${1:-r4l}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in product noload oneload; do
  lib=""; [ $v != product ] && lib=$R/tools/libbpp_exp_$v.so
  for P in 8192 1048576; do
    BPP_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-past-l3 --pool $P --gpu-seconds 0.4 > $O/bench_${v}_$P.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${v}_$P.json').readline()); r=d['roofline']; print('$v pool $P: kernel %.2f us, mean episode length %s' % (r['launch_us'], d['config']['mean_episode_length']))"
  done
done
