#!/usr/bin/env python3
"""Stress the episode statistics (GPU box).  Rounds 1-2 added finished episodes into 256 shared slots with float64
L2 atomics (slot = workgroup index >> 3); one GPU-suite run lost 0.05 - 0.7 % of those adds while every per-bin output
stayed bit-exact (profiles/archive/r03f_pytest_gpu.log).  The product now keeps one accumulator row per bin (plain
read-modify-write by the bin's own lane, fixed-order reduction) -- exact by construction.  This tool

  * runs >= --launches lock-steps over alternating env objects (so buffers come and go through the caching allocator),
    alternating xcd_remap, interleaved allocator churn and a second stream hammering the L2s;
  * after every chunk compares the per-bin rows and their fixed-order reduction with the ORACLE bit for bit
    (assert_array_equal) -- the exactness claim of the new path;
  * (round 3 only: a diagnostic build that kept the OLD slotted atomics beside the rows was compared as well --
    profiles/archive/r3_stress_stats_legacy_atomics.json: 20 000 launches, no add lost; that switch has since been removed from
    the source, commit c585660 has it.  The `legacy` branches below only act when a library exports
    bpp_debug_legacy_slots.)

    python tools/stress_stats.py --launches 12000          [BPP_HIP_LIB=.../libbpp_hip_legacystats.so]
Prints one JSON line."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bpp_amd
from oracle import oracle as orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=12000)
    ap.add_argument("--chunk", type=int, default=50)
    ap.add_argument("--bins", type=int, default=65536)
    ap.add_argument("--oracle-bins", type=int, default=4096, help="a second, small env is checked against the oracle")
    args = ap.parse_args()
    lib = bpp_amd._lib.lib()
    legacy = hasattr(lib, "bpp_debug_legacy_slots")
    if legacy:
        lib.bpp_debug_legacy_slots.argtypes = [ctypes.c_void_p, ctypes.c_int]
    size = (10, 10, 10)
    pool = bpp_amd.sequences.cut2_pool(size, 2048, seed=0)
    dev = torch.device("cuda", 0)
    orc.build()

    def legacy_read(clear=True):
        buf = np.zeros((256, 4), np.float64)
        bpp_amd._lib.check(lib.bpp_debug_legacy_slots(buf.ctypes.data, int(clear)))
        return buf

    envs, t_of = {}, {}

    def get_env(key, E, rot):
        if key not in envs:
            envs[key] = bpp_amd.BppVecEnv(E, size, enable_rotation=rot, pool=pool, device=dev)
            envs[key].reset()
            t_of[key] = 0
        return envs[key]

    # the oracle-checked env
    ref = orc.OracleEnv(pool, size, True, args.oracle_bins)
    ref.reset()
    small = get_env("small", args.oracle_bins, True)
    side = torch.cuda.Stream(device=dev)
    noise = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
    done_launches, chunks, mismatches_rows, legacy_bad, episodes = 0, 0, 0, [], 0.0
    if legacy:
        legacy_read(True)
    t_start = time.time()
    rng = np.random.RandomState(1)
    while done_launches < args.launches:
        chunks += 1
        key = ("big", chunks % 3)                      # three full-size env objects take turns ...
        if chunks % 7 == 0 and key in envs:            # ... and are dropped / re-created now and then (allocator reuse)
            del envs[key]
            torch.cuda.empty_cache() if chunks % 21 == 0 else None
        env = get_env(key, args.bins, bool(chunks % 2))
        bpp_amd._lib.set_knobs(xcd_remap=chunks % 2)
        acts = torch.empty(env.E, dtype=torch.int64, device=dev)
        junk = [torch.empty(int(rng.randint(1, 64)) << 20, dtype=torch.uint8, device=dev) for _ in range(3)]   # allocator churn
        with torch.cuda.stream(side):                  # a second stream keeps the L2s / fabric busy beside the lock-steps
            for _ in range(4):
                noise.add_(1)
        before, before_small = env.ep_acc.clone(), small.ep_acc.clone()
        if legacy:
            torch.cuda.synchronize()
            legacy_read(True)
        env.rollout_uniform(seed=3, step0=t_of[key], nsteps=args.chunk, actions=acts)
        t_of[key] += args.chunk
        # the same number of lock-steps on the small env, compared with the oracle
        small.rollout_uniform(seed=3, step0=t_of["small"], nsteps=args.chunk)
        orc.rollout_uniform(ref, 3, t_of["small"], args.chunk)
        t_of["small"] += args.chunk
        done_launches += 2 * args.chunk
        torch.cuda.synchronize()
        try:
            np.testing.assert_array_equal(small.ep_acc.cpu().numpy(), ref.ep_acc)
            np.testing.assert_array_equal(small.episode_stats().cpu().numpy(), ref.episode_stats())
        except AssertionError:
            mismatches_rows += 1
        delta = (env.ep_acc - before).cpu().numpy()
        dsm = (small.ep_acc - before_small).cpu().numpy()
        episodes += float(delta[:, 3].sum())
        if legacy:      # both envs' launches since the clear added to the slots: counts and lengths are integers
            slots = legacy_read(True).sum(0)
            want_cnt, want_len = delta[:, 3].sum() + dsm[:, 3].sum(), delta[:, 2].sum() + dsm[:, 2].sum()
            if slots[3] != want_cnt or slots[2] != want_len:
                legacy_bad.append({"chunk": chunks, "episodes_rows": float(want_cnt), "episodes_slots": float(slots[3]),
                                   "length_rows": float(want_len), "length_slots": float(slots[2])})
        del junk
    out = {"launches": done_launches, "chunks": chunks, "bins": args.bins, "episodes_finished": episodes,
           "seconds": round(time.time() - t_start, 1),
           "per_bin_rows_vs_oracle": "bit-exact in every chunk" if mismatches_rows == 0 else "%d chunks DIFFER" % mismatches_rows,
           "legacy_slotted_atomics_build": legacy,
           "legacy_slotted_atomics": (None if not legacy else
                                      ("no add lost or gained in %d chunks (counts and lengths are integers: exact compare)" % chunks
                                       if not legacy_bad else legacy_bad[:20]))}
    print(json.dumps(out))
    return 1 if mismatches_rows else 0


if __name__ == "__main__":
    sys.exit(main())
