#!/bin/bash
# What the driver runs at round end, in one call: smoke(), the GPU suite, the default bench line.  usage: tools/gpu_suite.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-suite}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench_driver_style.json')); r=d['roofline']
print('driver-style bench: %.1f M env steps/s, %.2f us/lock-step, kernel %.2f us frac %.3f, past L3 frac %.3f, reps %d, %.0f ms timed' % (d['value']/1e6, d['ms_per_step']*1e3, r['launch_us'], r['frac'], r['frac_past_l3'], d['reps'], d['timed_gpu_work_ms']))"
