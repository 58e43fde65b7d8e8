#!/bin/bash
# statistics reduction: all threads fetch a chunk of 64 rows per partial, 64 threads add them in row order
set -u
export TMPDIR=/tmp
TAG=${1:-r4zq}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stat or config or reduc or acc" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python tools/bench_acc_reduce.py > $O/acc_reduce.json 2>> $O/err.log; cat $O/acc_reduce.json
timeout 300 python tools/stress_stats.py > $O/stress_stats.log 2>&1; tail -2 $O/stress_stats.log
