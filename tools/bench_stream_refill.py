#!/usr/bin/env python3
"""Latency of one bpp_stream_refill as a function of the sequences every bin needs (tools, not the benchmark):
all bins are moved on by `need` episodes, then one refill is timed with HIP events on the launch stream.
usage: bench_stream_refill.py [--size W L H] [--envs E] [--needs 1 2 4] [--frac 1.0] [--reps 5]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bpp_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[10, 10, 10])
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--needs", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--frac", type=float, default=1.0, help="fraction of the bins that need sequences")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    E = args.envs
    depth = max(args.needs) + 4
    env = bpp_amd.BppVecEnv(E, tuple(args.size), stream=dict(bound=(2, 5), seed=0, depth=depth, refill_every=1))
    env.reset()
    gen = torch.Generator(device="cpu").manual_seed(1)
    out = {"size": args.size, "envs": E, "frac": args.frac, "knobs": bpp_amd._lib.get_knobs(), "refill_us": {}}
    for need in args.needs:
        ts = []
        for _ in range(args.reps + 1):
            sel = (torch.rand(E, generator=gen) < args.frac).to(env.device)
            env.state[:, 1] += sel.to(torch.int32) * need        # bpp_env_state.episode
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.refill()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        out["refill_us"][str(need)] = round(sorted(ts[1:])[len(ts[1:]) // 2], 1)
    assert int(env.stream_overflow.item()) == 0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
