#!/bin/bash
# Round evidence set for the current build, one gpurun call -> gpurun_out/<tag>/ (copy what is to be judged into
# profiles/ as r3_*): GPU suite, benches of the three single-GPU configs (+ the driver's --steps 20, + the primary pool),
# rocprofv3 kernel stats per config (headline leg and past-the-Infinity-Cache leg separately), PMC passes per config,
# mask / reset kernels, drop-in step(), statistics stress (product and legacy-atomics build), stream-supply benches,
# 2 ranks on one device without a launcher, RCCL with one rank, bins sweep, phase timelines (ablation build).
# usage: tools/gpu_final.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-final}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --rotation > $O/bench_rotation.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --pool-file tests/golden/cut2_dataset_10.npz > $O/bench_primary_pool_cut2_dataset.json 2>> $O/bench.err
for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for leg in "headline:--no-past-l3" "past_l3:--past-l3-only"; do
    lname=${leg%%:*}; largs=${leg#*:}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${name}_$lname -o run -- \
        python $R/bench.py --no-cpu-baseline $largs $args > $O/bench_under_rocprof_${name}_$lname.json 2>/dev/null)
    cp $O/prof_${name}_$lname/run_kernel_stats.csv $O/kernel_stats_${name}_$lname.csv 2>/dev/null
    rm -rf $O/prof_${name}_$lname
  done
  BENCH_EXTRA="--no-past-l3 --reps 3" tools/profile_pmc.sh ${TAG}_$name $args > /dev/null 2>&1
  cp $R/gpurun_out/pmc_${TAG}_$name/summary.txt $O/pmc_summary_$name.txt 2>/dev/null
done
python tools/pmc_to_json.py 10x10x10_rot0_E65536=$O/pmc_summary_10.txt 10x10x10_rot1_E65536=$O/pmc_summary_10rot.txt \
    20x20x20_rot0_E32768=$O/pmc_summary_20.txt > /dev/null 2>> $O/bench.err
cp profiles/hbm_traffic.json $O/hbm_traffic.json
python tools/bench_mask_kernels.py > $O/mask_and_reset_kernels.json 2>> $O/bench.err
python tools/bench_dropin_step.py > $O/dropin_step.json 2>> $O/bench.err
timeout 600 python tools/stress_stats.py --launches 12000 > $O/stress_stats_product.json 2> $O/stress.err
if [ -f $R/online-3d-bpp-drl_amd/csrc/libbpp_hip_legacystats.so ]; then
  BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_legacystats.so timeout 600 python tools/stress_stats.py --launches 20000 > $O/stress_stats_legacy_atomics.json 2>> $O/stress.err
fi
for cfg in "d32_r14:" "d64_r30:--stream-depth 64 --stream-refill 30" "20_d32_r14:--size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream $args > $O/bench_stream_$name.json 2>> $O/bench.err
done
BPP_STREAM_OVERLAP=0 python bench.py --no-cpu-baseline --stream > $O/bench_stream_d32_r14_serial.json 2>> $O/bench.err
BPP_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 100 --warmup 20 2> $O/bench_2ranks.err | tail -n 1 > $O/bench_2ranks_self_launched_one_device.json
BPP_BENCH_FORCE_PG=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 20 2> $O/bench_rccl_world1.err | head -n 1 > $O/bench_rccl_world1.json
timeout 600 python tools/sweep_bins.py --bins 65536 262144 > $O/sweep_bins_10.jsonl 2> $O/sweep.err
if [ -f $R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so ]; then
  for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so python tools/phase_timeline.py $args > $O/timeline_$name.json 2>> $O/bench.err
  done
fi
for f in bench bench_steps20 bench_rotation bench_20x20x20 bench_primary_pool_cut2_dataset; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json")); r = d["roofline"]
    print("$f: %.1f M env steps/s (%.1f M past L3), %.2f us/lock-step, kernel %.2f us frac %.3f / past L3 %.2f us frac %.3f" % (
        d["value"] / 1e6, (d["value_past_l3"] or 0) / 1e6, d["ms_per_step"] * 1e3, r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0))
except Exception as e:
    print("$f failed", e)
PY
done
head -2 $O/kernel_stats_10_headline.csv | cut -c1-200
ls $O | head -80
