#!/bin/bash
# Round-3 validation call: GPU suite, headline bench (default and the driver's --steps 20), statistics stress on the
# product build and on the legacy-atomics diagnostic build, reference-shaped step() cost.  usage: tools/gpu_check.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-check}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --pool-file tests/golden/cut2_dataset_10.npz > $O/bench_primary_pool_cut2_dataset.json 2>> $O/bench.err
timeout 600 python tools/stress_stats.py --launches 12000 > $O/stress_stats_product.json 2> $O/stress.err
if [ -f $R/online-3d-bpp-drl_amd/csrc/libbpp_hip_legacystats.so ]; then
  BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_legacystats.so timeout 600 python tools/stress_stats.py --launches 20000 > $O/stress_stats_legacy_atomics.json 2>> $O/stress.err
fi
python tools/bench_dropin_step.py > $O/dropin_step.json 2>> $O/bench.err
for f in bench bench_steps20 bench_primary_pool_cut2_dataset; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    r = d["roofline"]
    print("$f: %.1f M env steps/s (%.1f M past L3), %.2f us/lock-step, kernel %.2f us frac %.3f / past L3 %.2f us frac %.3f, reps %d" % (
        d["value"] / 1e6, (d["value_past_l3"] or 0) / 1e6, d["ms_per_step"] * 1e3, r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0, d["reps"]))
except Exception as e:
    print("$f failed", e)
PY
done
cat $O/stress_stats_product.json $O/stress_stats_legacy_atomics.json $O/dropin_step.json 2>/dev/null
tail -3 $O/bench.err $O/stress.err
