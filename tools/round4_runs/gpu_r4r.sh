#!/bin/bash
# candidate rule as three compares + mask logic instead of a per-lane threshold select: three configs + mask kernels
set -u
export TMPDIR=/tmp
TAG=${1:-r4r}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for cfg in "10:" "rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --gpu-seconds 0.8 $args > $O/bench_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').readline()); r=d['roofline']; print('$name: %.1f M env steps/s, kernel %.2f us, past L3 %.2f us' % (d['value']/1e6, r['launch_us'], r['launch_us_past_l3']))"
done
python tools/bench_mask_kernels.py > $O/mask_and_reset_kernels.json 2>> $O/bench.err
python -c "
import json; d=json.load(open('$O/mask_and_reset_kernels.json'))
for k,v in d.items(): print(k, {n: x['us'] for n,x in v.items()})"
