#!/bin/bash
# Build libbpp_hip.so with the phase-ablation hooks compiled in (profiling only; see BPP_ABL in
# csrc/bpp_kernels.hip).  Restore the product build with `BPP_FORCE_BUILD=1 python __graft_entry__.py`.
set -e
cd "$(dirname "$0")/../online-3d-bpp-drl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -DBPP_ENABLE_ABLATION -fPIC -shared \
    -o libbpp_hip.so bpp_kernels.hip
