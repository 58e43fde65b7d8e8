"""Marginal cost of the tile kernel's phases (profiling build only: tools/build_variant.sh abl -DBPP_ENABLE_ABLATION, then
BPP_HIP_LIB=.../libbpp_hip_abl.so): the mask-only kernel (fixed inputs -- skipping a phase does not change what the others
see) and the step kernel (skipping the prefix image / the candidates changes the masks and therefore the episodes: those rows
are indicative only) with phases switched off through bpp_knobs.ablate, 200 launches back to back between one pair of HIP
events.  Bits: 1 prefix image, 2 candidate passes, 4 mask store, 8 observation + heightmap store, 16 everything (empty
kernel), 32 deciding wave, 64 staging loads (step kernel).  Bit 32 in the FULL kernel is not "the cost of deciding": without
decisions no bin gets an item and the candidate passes fall away too (step abl=32 ~ step abl=3); compare abl=15 with abl=111."""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ctypes, json
import torch, bpp_amd
from bpp_amd import _lib
lib = _lib.lib()
N = 200
out = {}
cfgs = [((10, 10, 10), 65536, False), ((10, 10, 10), 65536, True), ((20, 20, 20), 32768, False)]
if len(sys.argv) > 1:
    cfgs = cfgs[:int(sys.argv[1])]
for size, E, rot in cfgs:
    pool = bpp_amd.sequences.cut2_pool(size, 256, seed=0)
    env = bpp_amd.BppVecEnv(E, size, enable_rotation=rot, pool=pool); env.reset()
    actions = torch.empty(E, dtype=torch.int64, device=env.device)
    env.rollout_uniform_sets(1, 0, 30, actions)
    A = size[0] * size[1]; M = A * (2 if rot else 1)
    hm = env.heightmaps().reshape(E, -1).contiguous(); items = env.preview(1)[:, 0, :].contiguous()
    mask = torch.empty(E, M, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    W, L, H = size
    keep = (env.hmap.clone(), env.state.clone(), actions.clone())
    res = {}

    def timed(fn, n=N):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return round(a.elapsed_time(b) * 1e3 / n, 2)

    for abl in (0, 1, 2, 3, 4, 7, 16):
        _lib.set_knobs(ablate=abl)
        res["mask_from_hmap abl=%d" % abl] = timed(lambda: lib.bpp_mask_from_hmap(hm.data_ptr(), items.data_ptr(), mask.data_ptr(), E, W, L, H, int(rot), 0, st))
    for abl in (0, 8, 4, 12, 1, 2, 3, 15, 32, 64, 96, 111, 16):
        _lib.set_knobs(ablate=abl)
        env.hmap.copy_(keep[0]); env.state.copy_(keep[1]); actions.copy_(keep[2])
        env.rollout_uniform_sets(1, 30, 10, actions, resume=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env.rollout_uniform_sets(1, 40, N, actions, resume=True)
        b.record()
        torch.cuda.synchronize()
        res["step abl=%d" % abl] = round(a.elapsed_time(b) * 1e3 / N, 2)
    _lib.set_knobs(ablate=0)
    out["%dx%dx%d%s_E%d" % (size + (" rot" if rot else "", E))] = res
    print("%dx%dx%d%s" % (size + (" rot" if rot else "",)), json.dumps(res), flush=True)
