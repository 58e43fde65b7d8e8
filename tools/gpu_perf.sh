#!/bin/bash
# Perf iteration call: GPU suite (full), the three single-GPU benches, mask-only kernels, drop-in step.  usage: tools/gpu_perf.sh <tag> [notest]
set -u
export TMPDIR=/tmp
TAG=${1:-perf}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "${2:-}" != "notest" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
  tail -4 $O/pytest_gpu.log
fi
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --rotation > $O/bench_rotation.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/bench.err
python tools/bench_mask_kernels.py > $O/mask_and_reset_kernels.json 2>> $O/bench.err
python tools/bench_dropin_step.py > $O/dropin_step.json 2>> $O/bench.err
for f in bench bench_rotation bench_20x20x20; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    r = d["roofline"]
    print("$f: %.1f M env steps/s (%.1f M past L3), %.2f us/lock-step, kernel %.2f us frac %.3f / past L3 %.2f us frac %.3f" % (
        d["value"] / 1e6, (d["value_past_l3"] or 0) / 1e6, d["ms_per_step"] * 1e3, r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0))
except Exception as e:
    print("$f failed", e)
PY
done
cat $O/mask_and_reset_kernels.json $O/dropin_step.json
tail -n 3 $O/bench.err
