#!/bin/bash
# Quick perf check of the current build: bench (no CPU baseline) at the three single-GPU configs + rocprofv3 kernel
# stats of the headline config (+ phase timeline when the profiling build is present).  -> gpurun_out/<tag>/
set -u
export TMPDIR=/tmp
TAG=${1:-quick}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --rotation > $O/bench_rotation.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_10 -o run -- \
    python $R/bench.py --no-cpu-baseline --steps 300 --warmup 50 > /dev/null 2>&1)
cp $O/prof_10/run_kernel_stats.csv $O/kernel_stats_10.csv 2>/dev/null
if [ -f $R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so ]; then
  BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so python tools/phase_timeline.py > $O/timeline_10.json 2>> $O/bench.err
fi
for f in bench bench_rotation bench_20x20x20; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    print("%-16s %.1f M env steps/s, %.2f us/lock-step, step kernel %.2f us, frac %.3f" % (
        "$f", d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["launch_us"], d["roofline"]["frac"]))
except Exception as e:
    print("$f failed", e)
PY
done
sed -n 2p $O/kernel_stats_10.csv | cut -d, -f1-4 | cut -c1-80,150-220
python - <<PY
import json
try:
    d = json.load(open("$O/timeline_10.json"))
    for role in ("wave0", "waves1-3"):
        print(role, {k: int(v) for k, v in d[role].items() if v is not None})
except Exception as e:
    print("no timeline", e)
PY
