#!/bin/bash
# r5n: does busy-polling the completion signal (HSA_ENABLE_INTERRUPT=0, set before the HIP runtime starts) shorten step()?
set -u
export TMPDIR=/tmp
TAG=${1:-r5n}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for mode in default poll default poll; do
  if [ $mode = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
  python tools/bench_dropin_step.py --steps 400 > $O/dropin_$mode.json 2>> $O/err.log
  python -c "
import json; d=json.load(open('$O/dropin_$mode.json')); print('$mode', {k: v for k, v in d.items() if k in ('tensors_us_per_lockstep','step_us_per_lockstep','step_host_us_in_step_async','step_host_us_in_step_wait','step+episodes_arrays_us_per_lockstep','step_eager_infos_us_per_lockstep','step+episodes_arrays_eager_infos_us_per_lockstep')})"
done
unset HSA_ENABLE_INTERRUPT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
