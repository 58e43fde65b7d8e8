"""SURVEY.md 8(f2), second half: the endless CUT-2 supply (include/bpp_abi.h: bpp_stream) that removes the finite
pool.  Episode k of global bin g must play the k-th sequence that random.Random(seed + g) yields through the
reference's MDlayerBoxCreator -- checked three ways: the product's device generator (compiled for the host SIMT
emulator here, on the MI355X in the `-m gpu` test) against (1) the oracle library's independent plain-C generator
driven through the same ring, and (2) Python's own `random` module feeding the restatement that is pinned to the
reference fixtures (tests/test_sequence_generators.py).  No sequence is ever replayed: the soak checks that every
episode of every bin shows the expected, distinct sequence."""
import random

import numpy as np
import pytest

from bpp_amd import sequences


class EmuStreamEnv(object):
    def __init__(self, emu, size, rot, E, base, spec):
        self.emu = emu
        self.env = emu.OracleEnv(None, size, rot, E, env_id_base=base, env_id_total=base + E + 3, stream=spec)

    def reset(self):
        return self.env.reset()

    def step(self, a):
        return self.env.step(a)

    def rollout(self, seed, step0, n):
        return self.emu.rollout_uniform(self.env, seed, step0, n)

    def state_records(self):
        return self.env.state


@pytest.mark.parametrize("size,rot,E,steps,depth,refill,native", [((10, 10, 10), False, 70, 60, 8, 5, False),
                                                                  ((10, 10, 10), True, 33, 40, 4, 1, False),
                                                                  ((10, 10, 10), False, 50, 80, 6, 3, True),
                                                                  ((20, 20, 20), False, 5, 30, 5, 2, True)])
def test_emulated_stream_supply_matches_oracle_and_python_random(emu, oracle, size, rot, E, steps, depth, refill, native):
    spec_check(lambda sz, r, n, base, spec: EmuStreamEnv(emu, sz, r, n, base, spec), oracle, size, rot, E, steps, depth, refill, native)


def spec_check(make_env, oracle, size, rot, E, steps, depth, refill, native):
    base, seed = 1000, 77
    spec = dict(bound=(2, 5), seed=seed, depth=depth, refill_every=refill)
    env = make_env(size, rot, E, base, spec)
    ref = oracle.OracleEnv(None, size, rot, E, env_id_base=base, env_id_total=base + E + 3, stream=spec)
    obs, mask = env.reset()
    robs, rmask = ref.reset()
    np.testing.assert_array_equal(obs, robs)
    np.testing.assert_array_equal(mask, rmask)
    A = size[0] * size[1]
    firsts = [[tuple(int(robs[e, (p + 1) * A]) for p in range(3))] for e in range(E)]   # first item of every episode shown
    if native:
        r, ra = env.rollout(9, 0, steps)
        o, oa = oracle.rollout_uniform(ref, 9, 0, steps)
        np.testing.assert_array_equal(ra, oa)
        for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=k)
    else:
        rng = np.random.RandomState(seed)
        for t in range(steps):
            a = oracle.sample_feasible(rmask, 5, t, env_id_base=base)
            a[rng.rand(E) < 0.15] = -1                    # many failures: bins race through their episodes
            r, o = env.step(a), ref.step(a)
            for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(r[k], o[k], err_msg="%s t=%d" % (k, t))
            rmask = o["mask"]
            for e in np.flatnonzero(o["done"]):
                firsts[e].append(tuple(int(o["obs"][e, (p + 1) * A]) for p in range(3)))
    st, rst = env.state_records(), ref.state
    for f in ("cursor", "episode", "n_boxes", "vol_sum", "ep_ret", "ep_len", "seq", "item_cur", "item_next", "item_reset"):
        np.testing.assert_array_equal(st[f], rst[f], err_msg=f)
    assert int(ref.overflow[0]) == 0
    # independent of both native generators: Python's own random.Random through the restatement pinned to the reference
    for e in sorted(set((0, E // 2, E - 1))):
        n_ep = int(rst["episode"][e])
        rng = random.Random(seed + base + e)
        seqs = [sequences.cut2_sequence(size, (2, 5), rng) for _ in range(n_ep + 1)]
        if not native:
            assert firsts[e] == [s[0] for s in seqs], e                  # every episode played the next sequence of the stream
        cur = int(rst["cursor"][e])
        want = seqs[n_ep][cur] if cur < len(seqs[n_ep]) else tuple(size)
        ic = int(rst["item_cur"][e])
        assert (ic & 255, (ic >> 8) & 255, (ic >> 16) & 255) == tuple(want), e
    assert int(rst["episode"].max()) >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("size,rot,E,steps,depth,refill,native", [((10, 10, 10), False, 4099, 120, 8, 5, False),
                                                                  ((10, 10, 10), True, 2000, 300, 8, 5, True),
                                                                  ((10, 10, 10), False, 65536, 60, 8, 5, True),
                                                                  ((20, 20, 20), False, 300, 400, 6, 3, True)])
def test_gpu_stream_supply_matches_oracle_and_python_random(oracle, size, rot, E, steps, depth, refill, native):
    import torch
    import bpp_amd

    class GpuStreamEnv(object):
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

    spec_check(GpuStreamEnv, oracle, size, rot, E, steps, depth, refill, native)


@pytest.mark.gpu
def test_gpu_stream_env_checkpoint_resume_and_factory():
    """state_dict() of a streaming env carries the ring and every bin's generator: a fresh env resumes identically;
    make_vec_envs accepts stream=... in place of a pool."""
    import types
    import torch
    import bpp_amd
    size, E = (10, 10, 10), 300
    spec = dict(bound=(2, 5), seed=5, depth=6, refill_every=3)
    env = bpp_amd.BppVecEnv(E, size, enable_rotation=True, stream=spec)
    env.reset()
    env.rollout_uniform(seed=3, step0=0, nsteps=17)
    ckpt = env.state_dict()
    first = env.rollout_uniform(seed=3, step0=17, nsteps=23)
    want = {k: getattr(first, k).clone() for k in ("obs", "mask", "counter", "ratio", "ep_ret")}
    other = bpp_amd.BppVecEnv(E, size, enable_rotation=True, stream=spec)
    other.load_state_dict(ckpt)
    again = other.rollout_uniform(seed=3, step0=17, nsteps=23)
    for k, v in want.items():
        assert torch.equal(getattr(again, k), v), k
    assert torch.equal(other.state, env.state) and torch.equal(other.gen_next, env.gen_next)
    args = types.SimpleNamespace(container_size=size, enable_rotation=False, data_type="cut2", box_size_set=None)
    envs = bpp_amd.make_vec_envs("Bpp-v0", 1, 16, 1.0, None, "cuda:0", False, args=args, stream=spec)
    obs = envs.reset()
    assert tuple(obs.shape) == (16, 400) and envs._stream is not None
