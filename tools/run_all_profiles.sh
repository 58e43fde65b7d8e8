#!/bin/bash
# Reproduce every artefact under profiles/ for the current tree (run on the GPU box through gpurun):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'tools/run_all_profiles.sh r02'
# Outputs go to gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --no-cpu-baseline --rotation > $OUT/${TAG}_bench_rotation.json 2>> $OUT/${TAG}_bench.err
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $OUT/${TAG}_bench_20x20x20.json 2>> $OUT/${TAG}_bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- \
    python /root/repo/bench.py --no-cpu-baseline > /dev/null 2>&1)
cp $OUT/${TAG}_prof/run_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
tools/profile_pmc.sh $TAG > /dev/null 2>&1
cp $OUT/pmc_$TAG/summary.txt $OUT/${TAG}_pmc_summary.txt 2>/dev/null
python tools/bench_masked_act.py > $OUT/${TAG}_masked_act_timing.json 2>/dev/null
for f in bench bench_rotation bench_20x20x20; do
  python - <<PY
import json
d = json.load(open("$OUT/${TAG}_$f.json"))
print("$f: %.1f M env steps/s, %.2f us/lock-step, step kernel %.2f us, frac %.3f" % (
    d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["launch_us"], d["roofline"]["frac"]))
PY
done
head -3 $OUT/${TAG}_kernel_stats.csv
grep -E "^step +(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|SQ_WAVES)" $OUT/${TAG}_pmc_summary.txt
