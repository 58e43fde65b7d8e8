#!/bin/bash
# r5b: full GPU suite (host_fin, ABI v14 final); phase ablation of the mask-only and step kernels (profiling build);
# drop-in step() with the finished bins' records written by the step kernel itself
set -u
export TMPDIR=/tmp
TAG=${1:-r5b}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -n 6 $O/pytest_gpu.log
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/dropin.err; cat $O/dropin_step.json | python -c "
import json,sys; d=json.load(sys.stdin); [print(k, v) for k, v in d.items() if k != 'note']"
BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so python tools/ablate_phases.py > $O/ablate_phases.txt 2> $O/ablate.err
cat $O/ablate_phases.txt; tail -n 3 $O/ablate.err
