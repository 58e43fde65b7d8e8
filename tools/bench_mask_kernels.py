"""Mask-only entry points and bpp_reset alone: 200 launches enqueued back to back through the C ABI (ctypes call
~3 us, well below the kernels' duration, so the queue never runs dry) between ONE pair of HIP events; 65 536 / 32 768
bins after 30 lock-steps of the uniform policy.  Two byte counts per kernel: `algorithmic` = SURVEY.md 8(d)'s contract
figure (the whole [E][4A] float32 observation is the input of get_possible_position), `real` = what the kernel touches
(plane 0 of the observation + the three item scalars, i.e. three more 64-byte lines per bin, + the mask) -- the honest
bandwidth."""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ctypes, json
import torch, bpp_amd
from bpp_amd import _lib
lib = _lib.lib()
if len(sys.argv) > 2 and sys.argv[1] == "--groups":     # bpp_knobs.tile_groups: groups of bins a wave of the tile kernel walks through
    _lib.set_knobs(tile_groups=int(sys.argv[2]))
out = {"tile_groups": _lib.get_knobs()["tile_groups"]}
N = 200
for size, E, rot in (((10, 10, 10), 65536, False), ((10, 10, 10), 65536, True), ((20, 20, 20), 32768, False)):
    pool = bpp_amd.sequences.cut2_pool(size, 256, seed=0)
    env = bpp_amd.BppVecEnv(E, size, enable_rotation=rot, pool=pool); env.reset()
    env.rollout_uniform(1, 0, 30)
    obs = env._res.obs.clone(); A = size[0] * size[1]; M = A * (2 if rot else 1)
    hm = env.heightmaps().reshape(E, -1).contiguous(); items = env.preview(1)[:, 0, :].contiguous()
    mask = torch.empty(E, M, device='cuda')
    want = env.location_masks.clone()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    W, L, H = size
    o = _lib.StepOut.from_buffer_copy(env._out)
    calls = (("mask_from_obs", lambda: lib.bpp_mask_from_obs(obs.data_ptr(), mask.data_ptr(), E, W, L, H, int(rot), 0, st), 16 * A + 4 * M, 4 * A + 3 * 64 + 4 * M),
             ("mask_from_hmap", lambda: lib.bpp_mask_from_hmap(hm.data_ptr(), items.data_ptr(), mask.data_ptr(), E, W, L, H, int(rot), 0, st), 4 * A + 12 + 4 * M, 4 * A + 12 + 4 * M),
             ("reset", lambda: lib.bpp_reset(env._batch_ref, 1, ctypes.byref(o), st), A + 16 * A + 4 * M + 48, A + 16 * A + 4 * M + 48))
    res = {}
    for name, fn, bytes_, real in calls:
        for _ in range(10):
            assert fn() == 0
        if name != "reset":
            assert torch.equal(mask, want), name
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(N):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / N
        res[name] = {"us": round(us, 2), "algorithmic_GBps": round(E * bytes_ / us / 1e3, 1), "real_bytes_per_bin": real,
                     "real_GBps": round(E * real / us / 1e3, 1), "real_frac_of_8TBps": round(E * real / us / 1e3 / 8000, 3)}
    out["%dx%dx%d%s_E%d" % (size + (" rot" if rot else "", E))] = res
print(json.dumps(out))
