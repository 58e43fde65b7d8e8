#!/bin/bash
# last check of the round's HEAD: smoke(), the GPU suite, the default bench line
set -u
export TMPDIR=/tmp
TAG=${1:-r4zi}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); print('%.1f M env steps/s' % (d['value']/1e6), d['roofline']['frac'], d['cpu_baseline']['kind'], d['cpu_baseline']['value'])"
