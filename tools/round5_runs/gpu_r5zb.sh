#!/bin/bash
# r5zb: step_wait() spinning on the step's completion word (bpp_mark / bpp_wait_mark, ABI v15) against hipStreamSynchronize
set -u
export TMPDIR=/tmp
TAG=${1:-r5zb}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dropin or completion" ) > $O/pytest_dropin.log 2>&1
tail -3 $O/pytest_dropin.log
timeout 300 python tools/bench_dropin_step.py --quick > $O/dropin_step_quick.json 2> $O/dropin.err
python - <<PY
import json
d = json.load(open("$O/dropin_step_quick.json"))
for k, v in d.items():
    if k.endswith("us_per_lockstep") or "host_us" in k: print(k, v)
PY
