#!/usr/bin/env python3
"""Per-phase shader-clock timeline of bpp_tile_kernel inside a full-size launch (profiling build only).

    tools/build_variant.sh abl -DBPP_ENABLE_ABLATION && BPP_HIP_LIB=online-3d-bpp-drl_amd/csrc/libbpp_hip_abl.so python tools/phase_timeline.py [--groups N]

Lane 0 of every wave of every 8th workgroup stamps s_memtime at the phase boundaries of ONE launch; printed: median
cycles between consecutive stamps, for the deciding wave (wave 0) and the other waves, plus wave lifetime."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAMES = ["start->staged", "barrier1 wait", "decide (wave0) / idle", "barrier2 wait", "to loop", "placement", "obs store issue",
         "prefix image", "candidates+draw", "mask store issue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[10, 10, 10])
    ap.add_argument("--rotation", action="store_true")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--groups", type=int, default=0)
    args = ap.parse_args()
    import torch
    import bpp_amd
    lib = bpp_amd._lib.lib()
    if not hasattr(lib, "bpp_debug_stamps"):
        raise SystemExit("needs the profiling build: tools/build_variant.sh abl -DBPP_ENABLE_ABLATION, then BPP_HIP_LIB=.../libbpp_hip_abl.so")
    size = tuple(args.size)
    pool = bpp_amd.sequences.cut2_pool(size, 2048, seed=0)
    bpp_amd._lib.set_knobs(tile_groups=args.groups)
    env = bpp_amd.BppVecEnv(args.envs, size, enable_rotation=args.rotation, pool=pool)
    env.reset()
    actions = torch.empty(args.envs, dtype=torch.int64, device=env.device)
    env.rollout_uniform(seed=1, step0=0, nsteps=40, actions=actions)
    torch.cuda.synchronize()
    bpp_amd._lib.set_knobs(ablate=256, tile_groups=args.groups)
    n = 4096 * 16
    buf = np.zeros(n, np.uint64)
    env.step_tensors(actions, sample=(1, 41, actions))
    torch.cuda.synchronize()
    lib.bpp_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    bpp_amd._lib.check(lib.bpp_debug_stamps(buf.ctypes.data, n, 1))
    env.step_tensors(actions, sample=(1, 42, actions))        # the measured launch (stamps cleared before it)
    torch.cuda.synchronize()
    bpp_amd._lib.check(lib.bpp_debug_stamps(buf.ctypes.data, n, 1))
    bpp_amd._lib.set_knobs(ablate=0, tile_groups=0)
    st = buf.reshape(4096, 16).astype(np.int64)
    ok = st[:, 0] > 0
    st = st[ok]
    wid = np.flatnonzero(ok) % 4
    out = {"size": list(size), "rotation": bool(args.rotation), "envs": args.envs, "groups": args.groups,
           "launch": bpp_amd._lib.launch_info(args.envs, size, args.rotation), "sampled_waves": int(ok.sum())}
    t0 = st[:, 0].min()
    for role, sel in (("wave0", wid == 0), ("waves1-3", wid != 0)):
        s = st[sel]
        d = {}
        for k in range(10):
            a, b = s[:, k], s[:, k + 1]
            good = (a > 0) & (b > 0)
            d[NAMES[k]] = float(np.median((b - a)[good])) if good.any() else None
        last = np.where(s[:, 10] > 0, s[:, 10], s[:, 9])
        d["lifetime_median"] = float(np.median(last - s[:, 0]))
        d["start_offset_median"] = float(np.median(s[:, 0] - t0))
        d["start_offset_p90"] = float(np.percentile(s[:, 0] - t0, 90))
        d["end_offset_max"] = float((last - t0).max())
        out[role] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
