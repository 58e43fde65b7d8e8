#!/bin/bash
# The round's evidence set for the current build: tools/gpu_measure.sh <tag> (suite, benches, rocprofv3 stats, PMC,
# sweep) plus phase timelines, the stream-mode bench and the multi-rank bench paths on one device.
set -u
TAG=${1:-final}
R=/root/repo
O=$R/gpurun_out/$TAG
cd $R
tools/gpu_measure.sh $TAG
export TMPDIR=/tmp
if [ -f $R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so ]; then
  for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768"; do
    name=${cfg%%:*}; args=${cfg#*:}
    BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so python tools/phase_timeline.py $args > $O/timeline_$name.json 2>> $O/bench.err
  done
fi
# endless device-generated supply: ring depth / refill interval variants, serial schedule, plain kernel; kernel stats;
# refill latency by sequences per bin; instruction counters of the refill kernels
for cfg in "d8_r5:--stream-depth 8 --stream-refill 5" "d16_r6:--stream-depth 16 --stream-refill 6" "d32_r14:" \
           "d64_r30:--stream-depth 64 --stream-refill 30" "20_d32_r14:--size 20 20 20 --envs 32768"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --no-cpu-baseline --stream $args > $O/bench_stream_$name.json 2>> $O/bench.err
done
BPP_STREAM_OVERLAP=0 python bench.py --no-cpu-baseline --stream > $O/bench_stream_d32_r14_serial.json 2>> $O/bench.err
BPP_STREAM_LEGACY=1 python bench.py --no-cpu-baseline --stream --stream-depth 8 --stream-refill 5 > $O/bench_stream_d8_r5_plain.json 2>> $O/bench.err
BPP_STREAM_LEGACY=1 python bench.py --no-cpu-baseline --stream --stream-depth 8 --stream-refill 5 --size 20 20 20 --envs 32768 --steps 100 --warmup 20 > $O/bench_stream_20_d8_r5_plain.json 2>> $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o run -- \
    python $R/bench.py --no-cpu-baseline --stream > /dev/null 2>&1)
cp $O/prof_stream/run_kernel_stats.csv $O/kernel_stats_stream_d32_r14.csv 2>/dev/null
python tools/bench_stream_refill.py --needs 1 2 4 > $O/refill_latency.jsonl 2>> $O/bench.err
python tools/bench_stream_refill.py --needs 1 --frac 0.11 >> $O/refill_latency.jsonl 2>> $O/bench.err
python tools/bench_stream_refill.py --needs 1 2 --size 20 20 20 --envs 32768 >> $O/refill_latency.jsonl 2>> $O/bench.err
BPP_STREAM_LEGACY=1 python tools/bench_stream_refill.py --needs 1 2 >> $O/refill_latency.jsonl 2>> $O/bench.err
for pass in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  set -- $pass
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_refill/$name -o p -- \
      python $R/tools/bench_stream_refill.py --needs 1 --reps 3 > /dev/null 2>&1) || echo "pass $name failed"
done
python tools/pmc_summary.py $O/pmc_refill > /dev/null 2>&1
cp $O/pmc_refill/summary.txt $O/refill_pmc_summary.txt 2>/dev/null
rm -rf $O/pmc_refill $O/prof_stream
BPP_BENCH_BACKEND=gloo BPP_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 100 --warmup 20 2> $O/bench_2ranks.err | tail -n 1 > $O/bench_2ranks_gloo_one_device.json
BPP_BENCH_FORCE_PG=1 timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 20 2> $O/bench_rccl_world1.err | head -n 1 > $O/bench_rccl_world1.json
timeout 600 python tools/sweep_bins.py --size 20 20 20 --bins 32768 131072 > $O/sweep_bins_20.jsonl 2>> $O/sweep.err
python tools/bench_mask_kernels.py > $O/mask_and_reset_kernels.json 2>> $O/bench.err
python tools/bench_masked_act.py > $O/masked_act_timing.json 2>> $O/bench.err
ls $O | head -50
