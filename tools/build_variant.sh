#!/bin/bash
# Build a diagnostic / experimental variant of the product library next to it: tools/build_variant.sh <name> [-D...]
#   -> online-3d-bpp-drl_amd/csrc/libbpp_hip_<name>.so; load it with BPP_HIP_LIB=<path> (never built implicitly).
# Known variants: abl = -DBPP_ENABLE_ABLATION (phase ablation / timestamps), legacystats = -DBPP_LEGACY_STATS_ATOMICS
# (tools/stress_stats.py: rounds 1-2's slotted float64 atomics kept beside the per-bin accumulators).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../online-3d-bpp-drl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -Wno-cuda-compat "$@" -fPIC -shared \
    -o libbpp_hip_$NAME.so bpp_kernels.hip
echo built $(pwd)/libbpp_hip_$NAME.so
