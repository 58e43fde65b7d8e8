#!/bin/bash
# soaks of this round's new paths: stream supply with both generators (ring rows with look-ahead entries, byte outputs,
# 16-bit lists), randomised parity stress, statistics stress
set -u
export TMPDIR=/tmp
TAG=${1:-r4p}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 900 python tools/soak_stream.py ) > $O/soak_stream.log 2>&1; tail -14 $O/soak_stream.log
( time timeout 600 python tools/stress_parity.py --trials 300 --seed 4 ) > $O/stress_parity.log 2>&1; tail -3 $O/stress_parity.log
( time timeout 400 python tools/stress_stats.py --launches 6000 ) > $O/stress_stats.json 2> $O/stress_stats.err; cut -c1-400 $O/stress_stats.json
