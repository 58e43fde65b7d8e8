#!/bin/bash
# Build a diagnostic / experimental variant of the product library next to it: tools/build_variant.sh <name> [-D...]
#   -> online-3d-bpp-drl_amd/csrc/libbpp_hip_<name>.so; load it with BPP_HIP_LIB=<path> (never built implicitly).
# Known variant: abl = -DBPP_ENABLE_ABLATION (phase ablation / timestamps) -- the only switch the shipped source carries.
# The A/B switches of rounds 2-3 (BPP_EXP_*, BPP_TILE_ACC_MODE / _SPEC / _LDS_PAD, BPP_LEGACY_STATS_ATOMICS) were removed
# from the source in round 4; their measurements are profiles/archive/r3x_* and profiles/archive/r3_stress_stats_legacy_atomics.json, the
# code is in the history (commit c585660); round 5's (BPP_AB_*) likewise, see tools/round5_runs/README.md.  New experiments: patch a copy, build it here, load it with BPP_HIP_LIB.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../online-3d-bpp-drl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -Wno-cuda-compat "$@" -fPIC -shared \
    -o libbpp_hip_$NAME.so bpp_kernels.hip
echo built $(pwd)/libbpp_hip_$NAME.so
