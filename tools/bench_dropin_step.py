#!/usr/bin/env python3
"""Cost of the REFERENCE-SHAPED step() (acktr/envs.py:170-193 types: obs on the device, reward CPU float32 [E,1],
done numpy bool, lazily built infos) per lock-step, next to the tensor-native step_tensors().  The action source is a
separate bpp_sample_feasible launch per step (a policy would stand there).  Variants: what the loop touches afterwards.

    python tools/bench_dropin_step.py [--envs 65536] [--steps 300]
One JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    import torch
    import bpp_amd
    size = (10, 10, 10)
    pool = bpp_amd.sequences.cut2_pool(size, 8192, seed=0)
    out = {"envs": args.envs, "steps": args.steps}
    for fresh in (False, True):
        env = bpp_amd.BppVecEnv(args.envs, size, pool=pool, fresh_outputs=fresh)
        env.reset()
        a = env.sample_feasible(seed=1, step=0)

        def run(kind, n, t0):
            nonlocal a
            for t in range(t0, t0 + n):
                if kind == "tensors":
                    env.step_tensors(a)
                else:
                    obs, rew, done, infos = env.step(a)
                    if kind == "step+finished_infos":
                        idx = infos.done_indices()
                        if idx.size:
                            infos[int(idx[0])]["episode"]["r"]
                    elif kind == "step+running_info":
                        infos[0]["ratio"]
                env.sample_feasible(seed=1, step=t + 1, out=a)

        for kind in ("tensors", "step", "step+finished_infos", "step+running_info"):
            run(kind, 30, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(kind, args.steps, 30)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / args.steps * 1e6
            out["%s%s_us_per_lockstep" % (kind, "_fresh_outputs" if fresh else "")] = round(us, 1)
        del env
        torch.cuda.empty_cache()
    out["note"] = ("step = step kernel (also writing reward + done, 5 bytes per bin, into page-locked host memory) + separate "
                   "action-sampling launch + stream sync; "
                   "+finished_infos: gather and copy the finished bins' (r, l, ratio, counter); +running_info: copy counter / ratio "
                   "of all bins (12 bytes per bin)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
