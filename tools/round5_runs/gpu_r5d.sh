#!/bin/bash
# r5d: VALU issue cost under sparse EXEC masks (tools/ubench exec); drop-in step() with the native collect of host_fin
set -u
export TMPDIR=/tmp
TAG=${1:-r5d}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
tools/ubench exec > $O/ubench_exec_masks.jsonl 2>&1; cat $O/ubench_exec_masks.jsonl
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/dropin.err; python -c "
import json; d=json.load(open('$O/dropin_step.json')); [print(k, v) for k, v in d.items() if k != 'note']"; tail -n 3 $O/dropin.err
