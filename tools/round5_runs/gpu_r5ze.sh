#!/bin/bash
# r5ze: 20x20x20 stream supply (counter generator, round 4's pipeline): kernel stats with the refills between the lock-steps
# (every kernel alone) and beside them
set -u
export TMPDIR=/tmp
TAG=${1:-r5ze}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for ov in 0 1; do
(cd /tmp && BPP_STREAM_OVERLAP=$ov timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline --stream --stream-rng counter --size 20 20 20 --envs 32768 --gpu-seconds 0.8 > $O/bench_overlap$ov.json 2>/dev/null)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats_counter_20_d32_r14_overlap$ov.csv 2>/dev/null; rm -rf $O/prof
done
for f in $O/kernel_stats_*.csv; do echo $f; head -7 $f | cut -d, -f1-8 | cut -c1-60,100-220; done
