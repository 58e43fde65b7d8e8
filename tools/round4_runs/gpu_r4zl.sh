#!/bin/bash
# randomised stream stress against the oracle (row cache on wherever the schedule allows)
set -u
export TMPDIR=/tmp
TAG=${1:-r4zl}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1200 python tools/stress_stream.py --trials 800 --seed 11 ) > $O/stress_stream.log 2>&1; tail -5 $O/stress_stream.log
