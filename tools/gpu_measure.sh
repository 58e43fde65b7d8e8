#!/bin/bash
# One GPU call: (optionally) the -m gpu suite, bench at the three single-GPU configs, rocprofv3 kernel stats and
# the PMC passes for each.  usage: tools/gpu_measure.sh <tag> [notest]   -> gpurun_out/<tag>/
set -u
export TMPDIR=/tmp
TAG=${1:-meas}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "${2:-}" != "notest" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
  tail -3 $O/pytest_gpu.log
fi
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --rotation > $O/bench_rotation.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/bench.err
for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- \
      python $R/bench.py --no-cpu-baseline $args > /dev/null 2>&1)
  cp $O/prof_$name/run_kernel_stats.csv $O/kernel_stats_$name.csv 2>/dev/null
  tools/profile_pmc.sh ${TAG}_$name $args > /dev/null 2>&1
  cp $R/gpurun_out/pmc_${TAG}_$name/summary.txt $O/pmc_summary_$name.txt 2>/dev/null
done
timeout 600 python tools/sweep_bins.py --bins 65536 262144 > $O/sweep_bins_10.jsonl 2> $O/sweep.err
for f in bench bench_steps20 bench_rotation bench_20x20x20; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    print("$f: %.1f M env steps/s, %.2f us/lock-step, step kernel %.2f us, frac %.3f" % (
        d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["launch_us"], d["roofline"]["frac"]))
except Exception as e:
    print("$f failed", e)
PY
done
grep -h -E "^step +(SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_BUSY_CYCLES|WRITE_SIZE|FETCH_SIZE)" $O/pmc_summary_*.txt
head -2 $O/kernel_stats_10.csv | cut -c1-200
cat $O/sweep_bins_10.jsonl
