#!/bin/bash
# round 4, second call: full GPU suite on the ABI-v11 build, drop-in step bench, acc_reduce timing, driver-style bench,
# instruction counters of the stream-supply kernels
set -u
export TMPDIR=/tmp
TAG=${1:-r4b}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/dropin.err; cat $O/dropin_step.json; tail -3 $O/dropin.err
python tools/bench_acc_reduce.py > $O/acc_reduce.json 2> $O/acc.err; cat $O/acc_reduce.json; tail -3 $O/acc.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_steps20.json').readline()); r=d['roofline']
print('driver-style bench: %.1f M env steps/s, %.2f us/lock-step, kernel %.2f us (b2b %.2f) frac %.3f, past L3 frac %.3f, reps %d, %.0f ms timed; kernel-limited %.1f M' % (d['value']/1e6, d['ms_per_step']*1e3, r['launch_us'], r['launch_us_back_to_back'], r['frac'], r['frac_past_l3'], d['reps'], d['timed_gpu_work_ms'], 65536/r['launch_us_back_to_back']))"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_stream -o p -- \
   python $R/bench.py --stream --no-cpu-baseline --steps 56 --warmup 28 --reps 3 > $O/pmc_stream.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc_stream/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        if row["Counter_Name"] == "SQ_WAVES": n[k] += 1
with open("$O/pmc_stream_summary.txt", "w") as out:
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", 0)):
        line = "%-62s launches %5d  " % (k, n[k]) + "  ".join("%s %.3e" % (c, v) for c, v in sorted(agg[k].items()))
        print(line); out.write(line + "\n")
PY
rm -rf $O/pmc_stream
