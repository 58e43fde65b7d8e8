#!/bin/bash
# where the ring-mode penalty of the step kernel comes from: serial schedule (refills between the chunks, nothing beside
# the step kernel), kernel statistics for: no head table / head table / head table whose stale lines are NOT re-read from
# the ring (experiment build, wrong items)
set -u
export TMPDIR=/tmp
TAG=${1:-r4t}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() {  # name, env, args
  (cd /tmp && env BPP_STREAM_OVERLAP=0 $2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --gpu-seconds 0.4 $3 > $O/bench_$1.json 2>> $O/bench.err)
  cp $O/prof_$1/run_kernel_stats.csv $O/kernel_stats_$1.csv 2>/dev/null; rm -rf $O/prof_$1
  echo "== $1"; grep bpp_tile_kernel $O/kernel_stats_$1.csv | sed "s/.*Params)\",//"
}


run head_noring "BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_noring.so" "--stream-head on"
