#!/bin/bash
# The bench lines of the evidence set only (after a change to bench.py itself): default, driver-style, rotation, 20^3, primary pool.
# usage: tools/gpu_benches.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-benches}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --rotation > $O/bench_rotation.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --pool-file tests/golden/cut2_dataset_10.npz > $O/bench_primary_pool_cut2_dataset.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --stream > $O/bench_stream_d32_r14.json 2>> $O/bench.err
BPP_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 100 --warmup 20 2> $O/bench_2ranks.err | tail -n 1 > $O/bench_2ranks_self_launched_one_device.json
timeout 300 python examples/rollout_with_policy.py --envs 16384 --steps 100 > $O/example_rollout_with_policy.txt 2>&1; tail -2 $O/example_rollout_with_policy.txt
for f in bench bench_steps20 bench_rotation bench_20x20x20 bench_primary_pool_cut2_dataset bench_stream_d32_r14 bench_2ranks_self_launched_one_device; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json")); r = d["roofline"]
    print("$f: %.1f M env steps/s, %.2f us/lock-step, kernel %.2f us (b2b %s) frac %.3f / past L3 %s us frac %s, reps %d" % (
        d["value"] / 1e6, d["ms_per_step"] * 1e3, r["launch_us"], r.get("launch_us_back_to_back"), r["frac"], r.get("launch_us_past_l3"), r.get("frac_past_l3"), d["reps"]))
except Exception as e:
    print("$f failed", e)
PY
done
tail -n 2 $O/bench.err
