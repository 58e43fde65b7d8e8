"""Randomised stress of the endless device supply on the GPU against the oracle (tools, not part of the suite): random bin
counts (incl. ones that are no multiple of anything), geometries on all three kernel paths, ring depths and refill periods
on both sides of what the row cache needs (bpp_batch.seq_cache: on wherever depth - refill >= 4), both generators, native
rollouts and stepwise play with forced failures.  Every output of every compared step, all state records and the
generators' progress must equal the oracle's (tests/test_stream_supply.py: spec_check).
    python tools/stress_stream.py --trials 60 --seed 1"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle
oracle.build()
import test_stream_supply as T
import torch
import bpp_amd


class Env(object):
    def __init__(self, sz, r, n, base, spec):
        self.env = bpp_amd.BppVecEnv(n, sz, enable_rotation=r, stream=spec, env_id_base=base, env_id_total=base + n + 3)

    def reset(self):
        obs = self.env.reset()
        return obs.cpu().numpy(), self.env.location_masks.cpu().numpy()

    def _out(self, r):
        out = {k: getattr(r, k).cpu().numpy() for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len")}
        out["reward"] = r.reward.cpu().numpy()[:, 0]
        return out

    def step(self, a):
        return self._out(self.env.step_tensors(np.asarray(a)))

    def rollout(self, seed, step0, n):
        acts = torch.empty(self.env.E, dtype=torch.int64, device=self.env.device)
        r = self.env.rollout_uniform(seed, step0, n, actions=acts)
        return self._out(r), acts.cpu().numpy()

    def state_records(self):
        assert int(self.env.stream_overflow.item()) == 0
        return self.env.state_numpy()


ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rng = np.random.RandomState(args.seed)
sizes = [((10, 10, 10), False), ((10, 10, 10), True), ((20, 20, 20), False), ((6, 6, 6), False), ((6, 6, 6), True), ((7, 9, 8), False),
         ((12, 12, 12), False)]
t0 = time.time()
cached = 0
for trial in range(args.trials):
    size, rot = sizes[rng.randint(len(sizes))]
    big = size[0] * size[1] >= 400
    E = int(rng.choice([1, 3, 63, 64, 65, 255, 1023, 1024, 1025, 2049, 4099])) if not big else int(rng.choice([1, 5, 64, 130, 513]))
    depth = int(rng.randint(5, 24))
    refill = int(rng.randint(1, depth - 2))
    gen = ("mt19937", "counter")[rng.randint(2)]
    native = bool(rng.randint(2)) and E >= 64      # (spec_check wants at least one finished episode: a handful of bins under the
                                                   # uniform policy may not get there in 30 steps; stepwise play forces failures)
    steps = int(rng.randint(30, 160 if not big else 90))
    cached += depth - refill >= 4
    T.spec_check(Env, oracle, size, rot, E, steps, depth, refill, native, gen=gen)
    print("ok", trial, size, rot, E, steps, depth, refill, gen, "native" if native else "stepwise", flush=True)
print("stress stream ok: %d trials (%d with the row cache), %.0f s" % (args.trials, cached, time.time() - t0))
