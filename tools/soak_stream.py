"""Long soak of the endless device supply on the GPU against the oracle (tools, not part of the suite: ~40 s on the box):
thousands of lock-steps per bin, i.e. hundreds of sequences and dozens of laps of every bin's MT19937 state, for both generators
(argv: mt19937 / counter); every output of
the last step, all state records and the generators' progress must equal the oracle's, the items shown Python's `random`."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle
oracle.build()
import test_stream_supply as T
import torch, bpp_amd

class Env(object):
    def __init__(self, sz, r, n, base, spec):
        self.env = bpp_amd.BppVecEnv(n, sz, enable_rotation=r, stream=spec, env_id_base=base, env_id_total=base + n + 3)
    def reset(self):
        obs = self.env.reset(); return obs.cpu().numpy(), self.env.location_masks.cpu().numpy()
    def _out(self, r):
        out = {k: getattr(r, k).cpu().numpy() for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len")}
        out["reward"] = r.reward.cpu().numpy()[:, 0]; return out
    def step(self, a): return self._out(self.env.step_tensors(np.asarray(a)))
    def rollout(self, seed, step0, n):
        acts = torch.empty(self.env.E, dtype=torch.int64, device=self.env.device)
        r = self.env.rollout_uniform(seed, step0, n, actions=acts); return self._out(r), acts.cpu().numpy()
    def state_records(self):
        assert int(self.env.stream_overflow.item()) == 0; return self.env.state_numpy()

gens = sys.argv[1:] or ["mt19937", "counter"]
for gen in gens:
  for (size, rot, E, steps, depth, refill, native) in [((10,10,10), False, 4096, 2000, 16, 6, True), ((10,10,10), True, 3000, 1500, 32, 14, True),
                                                     ((10,10,10), False, 2048, 500, 8, 5, False), ((20,20,20), False, 512, 1200, 16, 6, True),
                                                     ((6,6,6), False, 4096, 1500, 19, 6, True), ((10,10,10), False, 20000, 600, 64, 30, True)]:
    t0 = time.time()
    T.spec_check(Env, oracle, size, rot, E, steps, depth, refill, native, gen=gen)
    print("soak ok", gen, size, rot, E, steps, depth, refill, native, "%.1f s" % (time.time() - t0), flush=True)
