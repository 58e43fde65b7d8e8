#!/bin/bash
# r5h: drop-in step() with the multi-workgroup compaction (lazy and eager); GPU suite subset for the gather paths
set -u
export TMPDIR=/tmp
TAG=${1:-r5h}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_live_reference.py -q -x -k "dropin or gather or infos or live" > $O/pytest_subset.log 2>&1; tail -n 2 $O/pytest_subset.log
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/dropin.err; python -c "
import json; d=json.load(open('$O/dropin_step.json')); [print(k, v) for k, v in d.items() if k != 'note']"; tail -n 2 $O/dropin.err
