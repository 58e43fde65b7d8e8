#!/bin/bash
# Attribute the step kernel's LDS bank-conflict cycles to its phases: the profiling build (tools/build_variant.sh abl
# -DBPP_ENABLE_ABLATION) skips a phase per BPP_ABLATE bit (1 prefix image, 2 candidates, 4 mask store, 8 observation
# store; results are wrong then, only the counters matter); one rocprofv3 PMC pass each.  -> gpurun_out/<tag>/lds_conflicts_by_phase.txt
set -u
export TMPDIR=/tmp
TAG=${1:-lds}; shift || true
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
export BPP_HIP_LIB=$R/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so
: > $O/lds_conflicts_by_phase.txt
for cfg in "10:" "10rot:--rotation" "20:--size 20 20 20 --envs 32768 --pool 2048"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for abl in 0 1 2 3 4 8; do
    rm -rf $O/pmc_tmp
    (cd /tmp && BPP_ABLATE=$abl timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_tmp/p -o p -- \
        python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-headline --no-parity --no-past-l3 --reps 3 $args > /dev/null 2>&1)
    python $R/tools/pmc_summary.py $O/pmc_tmp > /dev/null 2>&1
    echo "config $name BPP_ABLATE=$abl" >> $O/lds_conflicts_by_phase.txt
    grep "^step" $O/pmc_tmp/summary.txt >> $O/lds_conflicts_by_phase.txt
  done
done
rm -rf $O/pmc_tmp
cat $O/lds_conflicts_by_phase.txt
