#!/usr/bin/env python3
"""bpp_episode_acc_reduce, one workgroup vs the many-workgroup form (scratch buffer): microseconds per call, HIP events
around 200 calls each, on rows that a real rollout filled; both results compared bit for bit.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bpp_amd
    out = {}
    for E in (65536, 262144):
        env = bpp_amd.BppVecEnv(E, (10, 10, 10), pool=bpp_amd.sequences.cut2_pool((10, 10, 10), 2048, seed=0))
        env.reset()
        a = torch.empty(E, dtype=torch.int64, device=env.device)
        env.rollout_uniform_sets(1, 0, 60, a)
        one, wide = env.episode_stats(wide=False), env.episode_stats(wide=True)
        assert torch.equal(one, wide) and float(one[3]) > 0
        for name, w in (("one_workgroup", False), ("wide", True)):
            acc = torch.zeros(4, dtype=torch.float64, device=env.device)
            for _ in range(10):
                env.episode_stats(wide=w, out=acc)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                env.episode_stats(wide=w, out=acc)
            e1.record()
            torch.cuda.synchronize()
            out["E%d_%s_us" % (E, name)] = round(e0.elapsed_time(e1) / 200 * 1e3, 2)
        del env
    print(json.dumps(out))


if __name__ == "__main__":
    main()
