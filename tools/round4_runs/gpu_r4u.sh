#!/bin/bash
# ring-mode step kernel with 2 / 4 groups per wave (workgroups that live longer than the cold read's latency), with and
# without the head table; serial schedule, kernel statistics
set -u
export TMPDIR=/tmp
TAG=${1:-r4u}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() {  # name, env, args
  (cd /tmp && env BPP_STREAM_OVERLAP=0 $2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.4 $3 > $O/bench_$1.json 2>> $O/bench.err)
  cp $O/prof_$1/run_kernel_stats.csv $O/kernel_stats_$1.csv 2>/dev/null; rm -rf $O/prof_$1
  echo "== $1"; grep "bpp_tile_kernel.* 0, 4" $O/kernel_stats_$1.csv | sed "s/.*Params)\",//"
}
run nohead_g2 "BPP_TILE_GROUPS=2" "--stream-head off"
run nohead_g4 "BPP_TILE_GROUPS=4" "--stream-head off"
run head_g2 "BPP_TILE_GROUPS=2" "--stream-head on"
run head_g4 "BPP_TILE_GROUPS=4" "--stream-head on"
