#!/bin/bash
# experiment build libbpp_hip_nt.so: observation and mask written with DWORD stores (a wave on one bin at a time, 256
set -u
export TMPDIR=/tmp
TAG=${1:-r4zm}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_nt.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for lib in libbpp_hip.so libbpp_hip_nt.so; do
  for cfg in "10:" "rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/$lib python bench.py --no-cpu-baseline --gpu-seconds 0.8 $args > $O/bench_${name}_$lib.json 2>> $O/bench.err
    python -c "
import json; d=json.loads(open('$O/bench_${name}_$lib.json').readline()); r=d['roofline']; print('$lib $name: %.1f M env steps/s, kernel %.2f us, past L3 %.2f us' % (d['value']/1e6, r['launch_us'], r['launch_us_past_l3']))"
  done
done
