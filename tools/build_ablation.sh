#!/bin/bash
# Build csrc/libbpp_hip_abl.so: the product library with the phase-ablation / phase-timestamp hooks compiled in
# (profiling only; see BPP_ABL / BPP_STAMP in csrc/bpp_kernels.hip).  Use it with BPP_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../online-3d-bpp-drl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -Wno-cuda-compat -DBPP_ENABLE_ABLATION -fPIC -shared \
    -o libbpp_hip_abl.so bpp_kernels.hip
echo built $(pwd)/libbpp_hip_abl.so
