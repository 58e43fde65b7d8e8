#!/bin/bash
# product with nontemporal output stores for the 20x20 bins: parity tests on 20x20, bench of the three configs, mask / reset
# kernels (they share the store phases), stream 20x20
set -u
export TMPDIR=/tmp
TAG=${1:-r4zo}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_live_reference.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for cfg in "10:" "rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --gpu-seconds 1.0 $args > $O/bench_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').readline()); r=d['roofline']; print('$name: %.1f M env steps/s, kernel %.2f us frac %.3f, past L3 %.2f us frac %.3f' % (d['value']/1e6, r['launch_us'], r['frac'], r['launch_us_past_l3'], r['frac_past_l3']))"
done
python tools/bench_mask_kernels.py > $O/mask_and_reset_kernels.json 2>> $O/bench.err
python -c "
import json; d=json.load(open('$O/mask_and_reset_kernels.json'))
for k,v in d.items(): print(k, {n: x['us'] for n,x in v.items()})"
for cfg in "counter_20:--stream-rng counter --size 20 20 20 --envs 32768" "mt19937_20:--size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.0 $args > $O/bench_stream_$name.json 2>> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_stream_$name.json').readline()); print('stream $name: %.1f M env steps/s' % (d['value']/1e6))"
done
