#!/bin/bash
# PMC passes for the step kernel (run on the GPU box through gpurun).  Counters are collected in their
# own runs, with --kernel-trace only (never with sys/hip/hsa tracing).  Results -> gpurun_out/pmc_<tag>/.
# usage: tools/profile_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- \
      python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-headline --no-parity ${BENCH_EXTRA:-} "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
python /root/repo/tools/pmc_summary.py $OUT
