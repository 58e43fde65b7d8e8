#!/bin/bash
# Round evidence set for the current build, one gpurun call -> gpurun_out/<tag>/ (copy what is to be judged into
# profiles/ as r5_*): GPU suite, benches of the three single-GPU configs (+ the driver's --steps 20 with the reference
# baseline, + the primary pool), rocprofv3 kernel stats per config (headline leg and past-the-Infinity-Cache leg
# separately), PMC passes per config, mask / reset kernels, drop-in step(), acc_reduce, stream-supply benches (both
# generators) with kernel stats, 2 ranks on one device without a launcher, RCCL with one rank, bins sweep.
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
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_steps20.json 2> $O/bench.err      # the driver's command: all configs + parity gate + epsilon leg
python bench.py --no-cpu-baseline > $O/bench.json 2>> $O/bench.err                                      # the same line at the default K = 500
python bench.py --no-cpu-baseline --only-headline --rotation > $O/bench_rotation.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --only-headline --size 20 20 20 --envs 32768 --pool 2048 > $O/bench_20x20x20.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --only-headline --pool-file tests/golden/cut2_dataset_10.npz > $O/bench_primary_pool_cut2_dataset.json 2>> $O/bench.err
for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for leg in "headline:--no-past-l3" "past_l3:--past-l3-only"; do
    lname=${leg%%:*}; largs=${leg#*:}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${name}_$lname -o run -- \
        python $R/bench.py --no-cpu-baseline --only-headline --no-parity --gpu-seconds 0.5 $largs $args > $O/bench_under_rocprof_${name}_$lname.json 2>/dev/null)
    cp $O/prof_${name}_$lname/run_kernel_stats.csv $O/kernel_stats_${name}_$lname.csv 2>/dev/null
    rm -rf $O/prof_${name}_$lname
  done
  BENCH_EXTRA="--no-past-l3 --reps 3" tools/profile_pmc.sh ${TAG}_$name $args > /dev/null 2>&1
  cp $R/gpurun_out/pmc_${TAG}_$name/summary.txt $O/pmc_summary_$name.txt 2>/dev/null
done
python tools/pmc_to_json.py --tracked-prefix profiles/${TAG}_ 10x10x10_rot0_E65536=$O/pmc_summary_10.txt 10x10x10_rot1_E65536=$O/pmc_summary_10rot.txt \
    20x20x20_rot0_E32768=$O/pmc_summary_20.txt > /dev/null 2>> $O/bench.err
cp profiles/hbm_traffic.json $O/hbm_traffic.json; cp profiles/${TAG}_pmc_summary_*.txt $O/ 2>/dev/null      # (profiles/ itself does not travel back: gpurun merges gpurun_out/ only)
python tools/bench_mask_kernels.py > $O/mask_and_reset_kernels.json 2>> $O/bench.err
python tools/bench_dropin_step.py > $O/dropin_step.json 2>> $O/bench.err
python tools/bench_acc_reduce.py > $O/acc_reduce.json 2>> $O/bench.err
python tools/bench_dropin_scale.py > $O/dropin_scale.json 2>> $O/bench.err                                 # the drop-in at the reference's own scale
python tools/bench_dropin_scale.py --rotation --envs 16,1024 > $O/dropin_scale_rot.json 2>> $O/bench.err
python tools/bench_masked_act.py > $O/masked_act.json 2>> $O/bench.err
for ck in default_cut_2:"" rotation_cut_2:--rotation; do                                                   # the reference's checkpoints on its whole test set, one batch
  python examples/evaluate_checkpoint.py --checkpoint oracle/_ref/pretrained_models/${ck%%:*}.pt --dataset oracle/_ref/dataset/cut_2.pt ${ck#*:} >> $O/evaluate_checkpoint.txt 2>> $O/bench.err
done
for e in 16 64 1024; do for g in "" "--graph"; do python examples/rollout_with_policy.py --envs $e --steps 1000 $g 2>> $O/bench.err | tail -1 >> $O/rollout_with_policy_graph.txt; done; done
for cfg in "mt19937_d32_r14:" "mt19937_d64_r30:--stream-depth 64 --stream-refill 30" "counter_d32_r14:--stream-rng counter" \
           "counter_d64_r30:--stream-rng counter --stream-depth 64 --stream-refill 30" \
           "counter_rot_d64_r30:--stream-rng counter --rotation --stream-depth 64 --stream-refill 30" \
           "counter_d128_r60:--stream-rng counter --stream-depth 128 --stream-refill 60" \
           "mt19937_20_d32_r14:--size 20 20 20 --envs 32768" "counter_20_d32_r14:--stream-rng counter --size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream --gpu-seconds 1.5 $args > $O/bench_stream_$name.json 2>> $O/bench.err
done
for g in mt19937 counter; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream_$g -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng $g --stream-depth 64 --stream-refill 30 --gpu-seconds 0.8 > /dev/null 2>&1)
  cp $O/prof_stream_$g/run_kernel_stats.csv $O/kernel_stats_stream_${g}_d64_r30.csv 2>/dev/null; rm -rf $O/prof_stream_$g
  (cd /tmp && BPP_STREAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream_$g -o run -- \
      python $R/bench.py --no-cpu-baseline --stream --stream-rng $g --stream-depth 64 --stream-refill 30 --gpu-seconds 0.5 > /dev/null 2>&1)
  cp $O/prof_stream_$g/run_kernel_stats.csv $O/kernel_stats_stream_${g}_d64_r30_serial_schedule.csv 2>/dev/null; rm -rf $O/prof_stream_$g
done
BPP_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 100 --warmup 20 --gpu-seconds 1 --only-headline 2> $O/bench_2ranks.err | tail -n 1 > $O/bench_2ranks_self_launched_one_device.json
BPP_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 8 --envs 8192 --steps 20 --warmup 5 --gpu-seconds 0.3 --no-past-l3 2> $O/bench_8ranks.err | tail -n 1 > $O/bench_8ranks_self_launched_one_device_8192_bins_each.json
BPP_BENCH_FORCE_PG=1 timeout 300 python bench.py --no-cpu-baseline --only-headline --steps 100 --warmup 20 --gpu-seconds 1 2> $O/bench_rccl_world1.err | head -n 1 > $O/bench_rccl_world1.json
timeout 600 python tools/sweep_bins.py --bins 65536 262144 1048576 > $O/sweep_bins_10.jsonl 2> $O/sweep.err      # (a fresh process: late in a long script the same launches measured 10 % slower at 262 144 bins -- see profiles/README.md, r6e)
timeout 600 python tools/sweep_bins.py --rotation --bins 65536 262144 > $O/sweep_bins_10_rot.jsonl 2>> $O/sweep.err
for f in bench_steps20 bench bench_rotation bench_20x20x20 bench_primary_pool_cut2_dataset; do
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/$f.json") if l.startswith("{")][0]); r = d["roofline"]
    print("$f: %.1f M env steps/s (%.1f M past L3), %.2f us/lock-step, kernel %.2f us frac %.3f / past L3 %.2f us frac %.3f" % (
        d["value"] / 1e6, (d["value_past_l3"] or 0) / 1e6, d["ms_per_step"] * 1e3, r["launch_us"], r["frac"], r["launch_us_past_l3"] or 0, r["frac_past_l3"] or 0))
    c = d.get("cpu_baseline")
    if c: print("   cpu_baseline:", c["kind"], "%.0f env steps/s on %d cores; R1 %.0f; C port %.3g" % (c["value"], c["cores"], c.get("reference_as_is_R1", {}).get("value", 0), c.get("ours_cpu", {}).get("value", 0)))
except Exception as e:
    print("$f failed", e)
PY
done
for f in $O/bench_stream_*.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f'.split('bench_stream_')[1][:-5], '%.1f M env steps/s, %.2f us/lock-step' % (d['value']/1e6, d['ms_per_step']*1e3))"; done
head -3 $O/kernel_stats_10_headline.csv | cut -c1-180
ls $O | wc -l
