#!/bin/bash
# Refill-kernel microbenchmark in one gpurun call: latency vs sequences per bin, per-kernel times, instruction counters.
set -u
export TMPDIR=/tmp
TAG=${1:-refill}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/bench_stream_refill.py --needs 1 2 4 > $O/refill_10.json 2> $O/err.txt; cat $O/refill_10.json
python tools/bench_stream_refill.py --needs 1 --frac 0.11 >> $O/refill_10.json 2>> $O/err.txt; tail -1 $O/refill_10.json
python tools/bench_stream_refill.py --needs 1 2 --size 20 20 20 --envs 32768 > $O/refill_20.json 2>> $O/err.txt; cat $O/refill_20.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/tools/bench_stream_refill.py --needs 1 2 4 > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:5]:
    print("%-44s calls %5s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cd /tmp
for pass in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  set -- $pass
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$name -o p -- \
      python $R/tools/bench_stream_refill.py --needs 1 --reps 3 > $O/pmc_$name.log 2>&1 || echo "pass $name failed"
done
python $R/tools/pmc_summary.py $O/pmc | grep -E "^(cut|scan|sort)" 
cp $O/pmc/summary.txt $O/pmc_summary.txt 2>/dev/null; rm -rf $O/pmc
