#!/bin/bash
# Stream-supply check of the current build in one gpurun call: GPU stream tests, bench --stream at several ring
# depths / refill intervals (side-stream overlap on and off, plain kernel), rocprofv3 kernel stats.  -> gpurun_out/<tag>/
set -u
export TMPDIR=/tmp
TAG=${1:-stream}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_supply.py -m gpu -x -q > $O/pytest_stream.txt 2>&1
tail -3 $O/pytest_stream.txt
run() {  # name, env, args...
  local name=$1; shift
  local envs=$1; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline --stream --steps 600 --warmup 100 "$@" > $O/bench_$name.json 2>> $O/bench.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    print("%-28s %8.1f M env steps/s, %7.2f us/lock-step" % ("$name", d["value"] / 1e6, d["ms_per_step"] * 1e3))
except Exception as e:
    print("$name failed", e)
PY
}
run d8_r5 BPP_X=0 --stream-depth 8 --stream-refill 5
run d8_r5_plain BPP_STREAM_LEGACY=1 --stream-depth 8 --stream-refill 5
run d16_r6 BPP_X=0 --stream-depth 16 --stream-refill 6
run d16_r6_serial BPP_STREAM_OVERLAP=0 --stream-depth 16 --stream-refill 6
run d16_r2 BPP_X=0 --stream-depth 16 --stream-refill 2
run d32_r14 BPP_X=0 --stream-depth 32 --stream-refill 14
run d32_r14_serial BPP_STREAM_OVERLAP=0 --stream-depth 32 --stream-refill 14
run d64_r30 BPP_X=0 --stream-depth 64 --stream-refill 30
run 20_d16_r6 BPP_X=0 --stream-depth 16 --stream-refill 6 --size 20 20 20 --envs 32768
run 20_d16_r6_serial BPP_STREAM_OVERLAP=0 --stream-depth 16 --stream-refill 6 --size 20 20 20 --envs 32768
run 20_d8_r5_plain BPP_STREAM_LEGACY=1 --stream-depth 8 --stream-refill 5 --size 20 20 20 --envs 32768
for cfg in "16 6 1" "16 6 0" "8 5 1"; do
  set -- $cfg
  (cd /tmp && BPP_STREAM_OVERLAP=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d$1_r$2_o$3 -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-depth $1 --stream-refill $2 --steps 300 --warmup 50 > /dev/null 2>&1)
  cp $O/prof_d$1_r$2_o$3/run_kernel_stats.csv $O/kernel_stats_d$1_r$2_o$3.csv 2>/dev/null
  echo "== depth $1 refill $2 overlap $3"; head -6 $O/kernel_stats_d$1_r$2_o$3.csv | cut -d, -f1-4,6-7 | cut -c1-60,100-200
  rm -rf $O/prof_d$1_r$2_o$3
done
