export BPP_HIP_LIB=/root/repo/online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so
mkdir -p gpurun_out/r02i
for abl in 0 512 1024; do
  BPP_ABLATE=$abl python bench.py --no-cpu-baseline --steps 300 > gpurun_out/r02i/bench_abl$abl.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r02i/bench_abl$abl.json')); print('ablate $abl: %.2f us/step'%(d['ms_per_step']*1e3))"
done
