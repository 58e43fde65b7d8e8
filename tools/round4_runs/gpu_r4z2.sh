#!/bin/bash
# where the row-cache kernel's time goes (experiment builds, wrong items): v1 = copier workgroups whose reads hit a cached
# 64-row window, v2 = no copier workgroups, every look-ahead "misses" into a cached 64-row window
set -u
export TMPDIR=/tmp
TAG=${1:-r4z2}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in v1 v2; do
  (cd /tmp && BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_$v.so BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --stream-cache on --gpu-seconds 0.4 > /dev/null 2>&1)
  cp $O/prof_$v/run_kernel_stats.csv $O/kernel_stats_$v.csv 2>/dev/null; rm -rf $O/prof_$v
  echo "== $v"; grep "bpp_tile_kernel.*<10, 10, 1, false, 0, 4, 1>" $O/kernel_stats_$v.csv | sed "s/.*Params)\",//"
done
