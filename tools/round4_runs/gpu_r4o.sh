#!/bin/bash
# drop-in step() with the gather writing page-locked memory directly; tests of the paths it touches
set -u
export TMPDIR=/tmp
TAG=${1:-r4o}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_live_reference.py -m gpu -q -x -k "dropin or finished_infos or live_reference_infos" ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/dropin.err; python -c "
import json; d=json.load(open('$O/dropin_step.json')); print({k: v for k, v in d.items() if k != 'note'})"
