#!/bin/bash
# GPU call C: -m gpu suite, then the tile kernel at 1 / 2 / 4 groups per wave on the three single-GPU configs
# (bench + rocprofv3 kernel stats), PMC passes for the default.  -> gpurun_out/r02c/
set -u
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
for g in 1 2 4; do
  export BPP_TILE_GROUPS=$g
  python bench.py --no-cpu-baseline > $O/bench_g$g.json 2>> $O/bench.err
  python bench.py --no-cpu-baseline --rotation > $O/bench_rotation_g$g.json 2>> $O/bench.err
  python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20_g$g.json 2>> $O/bench.err
  for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
    name=${cfg%%:*}; args=${cfg#*:}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${name}_g$g -o run -- \
        python $R/bench.py --no-cpu-baseline --steps 200 --warmup 50 $args > /dev/null 2>&1)
    cp $O/prof_${name}_g$g/run_kernel_stats.csv $O/kernel_stats_${name}_g$g.csv 2>/dev/null
  done
done
unset BPP_TILE_GROUPS
for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  tools/profile_pmc.sh r02c_$name $args > /dev/null 2>&1
  cp $R/gpurun_out/pmc_r02c_$name/summary.txt $O/pmc_summary_$name.txt 2>/dev/null
done
timeout 600 python tools/sweep_bins.py --bins 65536 262144 1048576 > $O/sweep_bins_10.jsonl 2> $O/sweep.err
for f in $O/bench_*.json; do
  python - <<PY
import json
try:
    d = json.load(open("$f"))
    print("%-28s %.1f M env steps/s, %.2f us/lock-step, step kernel %.2f us, frac %.3f" % (
        "$f".split("/")[-1], d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["launch_us"], d["roofline"]["frac"]))
except Exception as e:
    print("$f failed", e)
PY
done
for f in $O/kernel_stats_*.csv; do echo $f; sed -n 2p $f | cut -d, -f1-4 | cut -c1-160; done
grep -h -E "^step +(SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT)" $O/pmc_summary_*.txt
cat $O/sweep_bins_10.jsonl
