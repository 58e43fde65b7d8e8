#!/usr/bin/env python3
"""Throughput of the fused step kernel vs bins per GPU, with the OUTPUT buffers rotated so that nothing the
kernel writes can stay resident in the 256 MiB Infinity Cache (MI355X_MICROARCH.md: L3 hits are counted by
FETCH/WRITE_SIZE and a <100 MB working set can hide HBM behaviour).  Per point: R sets of (obs, mask, scalars)
with R x bytes-per-set >= --span-gb, lock-step t writes set t % R; the same loop with ONE set for comparison.
Prints one JSON line per point (kept under profiles/ as the artefact behind DESIGN.md's "flat in E" claim).

    python tools/sweep_bins.py [--size 10 10 10] [--rotation] [--bins 16384 65536 262144 1048576]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[10, 10, 10])
    ap.add_argument("--rotation", action="store_true")
    ap.add_argument("--bins", type=int, nargs="+", default=[16384, 65536, 262144, 1048576])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--span-gb", type=float, default=1.2)
    args = ap.parse_args()
    import torch
    import bpp_amd
    from bpp_amd.vec_env import StepTensors
    size = tuple(args.size)
    A = size[0] * size[1]
    M = A * (2 if args.rotation else 1)
    pool = bpp_amd.sequences.cut2_pool(size, 2048 if A > 100 else 8192, seed=0)
    for E in args.bins:
        env = bpp_amd.BppVecEnv(E, size, enable_rotation=args.rotation, pool=pool)
        env.reset()
        set_bytes = E * (16 * A + 4 * M + 29)
        for rotate in (True, False):
            R = max(2, int(args.span_gb * 1e9 / set_bytes) + 1) if rotate else 1
            sets = [env._alloc() for _ in range(R)]
            actions = env.sample_feasible(seed=1, step=0)

            def run(n, t0):
                for t in range(t0, t0 + n):
                    env._bufs, env._out = sets[t % R]
                    env._res = env._bufs
                    env.step_tensors(actions, sample=(1, t + 1, actions))

            run(60, 0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(args.steps, 60)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.steps
            print(json.dumps({"size": list(size), "rotation": bool(args.rotation), "bins": E, "output_sets": R,
                              "output_span_MB": round(R * set_bytes / 1e6, 1), "us_per_lockstep": round(us, 2),
                              "env_steps_per_s": E / us * 1e6, "steps": args.steps}), flush=True)
            del sets
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
