#!/bin/bash
# round 4, first call: new live-reference GPU tests, smoke, driver-style bench with the reference baseline
set -u
export TMPDIR=/tmp
TAG=${1:-r4a}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
nproc > $O/nproc.txt; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())" >> $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 1200 python -m pytest tests/test_gpu_vs_live_reference.py -m gpu -q -x ) > $O/pytest_live.log 2>&1
tail -15 $O/pytest_live.log
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench.err
tail -5 $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_driver_style.json').readline()); r=d['roofline']; c=d['cpu_baseline']
print('driver-style bench: %.1f M env steps/s, %.2f us/lock-step, kernel %.2f us frac %.3f, past L3 frac %.3f, reps %d, %.0f ms timed' % (d['value']/1e6, d['ms_per_step']*1e3, r['launch_us'], r['frac'], r['frac_past_l3'], d['reps'], d['timed_gpu_work_ms']))
print('cpu_baseline', c['kind'], c['value'], c['cores'], c.get('reference_as_is_R1',{}).get('value'), c.get('ours_cpu',{}).get('value'))"
