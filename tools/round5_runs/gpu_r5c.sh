#!/bin/bash
# r5c: drop-in step() with host_fin; what the SQ counters say about the mask-only kernel (is it issue-bound at all?):
# counters of 200 bpp_mask_from_hmap launches per config, separate passes; available counter list kept for reference
set -u
export TMPDIR=/tmp
TAG=${1:-r5c}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/bench_dropin_step.py > $O/dropin_step.json 2> $O/dropin.err; python -c "
import json; d=json.load(open('$O/dropin_step.json')); [print(k, v) for k, v in d.items() if k != 'note']"
cd /tmp
rocprofv3 --list-avail > $O/counters_avail.txt 2>&1
grep -c "" $O/counters_avail.txt
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -o p -- python $R/tools/bench_mask_kernels.py > $O/pmc_$name.log 2>&1 || echo "pass $name failed"
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
pass c SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT
pass d SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
for name in "abcd":
    files = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % name, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "bpp_tile_kernel" not in k: continue
            short = "rot" if "Lb1E" in k else ("20" if "Li20E" in k else "10")
            mode = k.split("ELb")[1][2:].split("E")[0] if "ELb" in k else "?"
            acc[(short, k[k.find("ILi"):k.find("EEEv")])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key in sorted(acc):
        print(name, key, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[key].items())}, "n=%d" % max(len(v) for v in acc[key].values()))
PY
