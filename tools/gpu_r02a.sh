#!/bin/bash
# GPU call A of round 2: full -m gpu suite, bench at the three single-GPU configs, rocprofv3 kernel stats,
# PMC passes for all three, counter calibration, bins sweep, instruction micro-benchmarks, N>1 bench path
# exercised on one device.  Everything lands in gpurun_out/r02a/.
set -u
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --rotation > $O/bench_rotation.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- \
    python $R/bench.py --no-cpu-baseline > /dev/null 2>&1)
cp $O/prof/run_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
tools/profile_pmc.sh r02a_10 > /dev/null 2>&1
tools/profile_pmc.sh r02a_10rot --rotation > /dev/null 2>&1
tools/profile_pmc.sh r02a_20 --size 20 20 20 --envs 32768 --pool 2048 > /dev/null 2>&1
for t in r02a_10 r02a_10rot r02a_20; do cp $R/gpurun_out/pmc_$t/summary.txt $O/pmc_summary_$t.txt 2>/dev/null; done
# counter calibration on known byte counts
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib_fetch -o p -- $R/tools/ubench calib > $O/calib_fetch.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/calib_write -o p -- $R/tools/ubench calib > $O/calib_write.log 2>&1)
python - <<PY > $O/calib_summary.txt 2>&1
import csv, glob
for tag in ("fetch", "write"):
    for f in glob.glob("$O/calib_%s/*/*counter_collection.csv" % tag) + glob.glob("$O/calib_%s/*counter_collection.csv" % tag):
        for row in csv.DictReader(open(f)):
            print(tag, row["Kernel_Name"][:40], row["Counter_Name"], row["Counter_Value"])
PY
timeout 300 tools/ubench > $O/ubench.jsonl 2>&1
timeout 600 python tools/sweep_bins.py > $O/sweep_bins_10.jsonl 2> $O/sweep.err
timeout 600 python tools/sweep_bins.py --size 20 20 20 --bins 8192 32768 131072 > $O/sweep_bins_20.jsonl 2>> $O/sweep.err
# the N > 1 bench path on one device: 2 gloo ranks sharing GPU 0, and the RCCL branch with a 1-rank group
BPP_BENCH_BACKEND=gloo BPP_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 100 --warmup 20 > $O/bench_2ranks_gloo_one_device.json 2> $O/bench_2ranks.err
BPP_BENCH_FORCE_PG=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
ls -la $O
