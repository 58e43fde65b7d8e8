#!/bin/bash
# r5l: the mask-only kernels with TWO groups of bins per wave (loads of both groups up front, the mask stores of the first
# overlap the second's prefix image and candidates): bpp_knobs.tile_groups = 2 against the default one group
set -u
export TMPDIR=/tmp
TAG=${1:-r5l}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "masks or knob" > $O/pytest_subset.log 2>&1; tail -n 2 $O/pytest_subset.log
for g in 1 2 1 2; do
  python tools/bench_mask_kernels.py --groups $g > $O/mask_kernels_groups$g.json 2>> $O/err.log; python -c "
import json; d=json.load(open('$O/mask_kernels_groups$g.json'))
for k,v in d.items():
    if k != 'tile_groups': print('groups $g', k, {n: x['us'] for n,x in v.items()})"
done
tail -n 2 $O/err.log
