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
                                                                  ((20, 20, 20), False, 5, 30, 5, 2, True),
                                                                  # depth >= 2 * refill + 3: the driver's side-stream schedule
                                                                  # (at most two sequences per bin and refill unless urgent)
                                                                  ((10, 10, 10), False, 70, 50, 9, 3, True),
                                                                  ((10, 10, 10), True, 40, 64, 20, 8, True),
                                                                  ((20, 20, 20), False, 5, 30, 9, 3, True),
                                                                  # a bin that holds two or three items: episodes shorter than
                                                                  # the refill interval, bins fall behind and become urgent
                                                                  ((6, 6, 6), False, 70, 60, 19, 6, True),
                                                                  ((6, 6, 6), True, 33, 45, 24, 8, True),
                                                                  # tightest schedules with a row cache (depth - refill == 4)
                                                                  ((10, 10, 10), False, 70, 60, 8, 4, False),
                                                                  ((10, 10, 10), True, 33, 40, 5, 1, False),
                                                                  ((20, 20, 20), False, 5, 30, 6, 2, True),
                                                                  ((6, 6, 6), False, 70, 60, 10, 6, True),
                                                                  # W*L % 4 != 0: the generic cell-scan kernel drops the cache lines
                                                                  ((7, 9, 8), False, 20, 40, 9, 3, False)])
def test_emulated_stream_supply_matches_oracle_and_python_random(emu, oracle, size, rot, E, steps, depth, refill, native):
    spec_check(lambda sz, r, n, base, spec: EmuStreamEnv(emu, sz, r, n, base, spec), oracle, size, rot, E, steps, depth, refill, native)


def spec_check(make_env, oracle, size, rot, E, steps, depth, refill, native, gen="mt19937"):
    base, seed = 1000, 77
    # row cache (bpp_batch.seq_cache) wherever the schedule leaves the row of look-ahead it costs
    spec = dict(bound=(2, 5), seed=seed, depth=depth, refill_every=refill, rng=gen, cache=depth - refill >= 4)
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
        if gen == "counter":    # a sequence is a function of (seed, stream id = global bin id, episode)
            seqs = [sequences.cut2_sequence(size, (2, 5), sequences.CounterRandom(seed, base + e, k)) for k in range(n_ep + 1)]
        else:
            pyrng = random.Random(seed + base + e)
            seqs = [sequences.cut2_sequence(size, (2, 5), pyrng) for _ in range(n_ep + 1)]
        if not native:
            assert firsts[e] == [s[0] for s in seqs], e                  # every episode played the next sequence of the stream
        cur = int(rst["cursor"][e])
        want = seqs[n_ep][cur] if cur < len(seqs[n_ep]) else tuple(size)
        ic = int(rst["item_cur"][e])
        assert (ic & 255, (ic >> 8) & 255, (ic >> 16) & 255) == tuple(want), e
    assert int(rst["episode"].max()) >= 1


def knob_check(front, oracle, set_knobs, size, E, depth, steps, pattern, refill=None, seed=7, base=100, fail=0.2, gen="mt19937"):
    """Fast pipeline (scan / cut / sort) and the one-lane-per-bin kernel are interchangeable refill by refill: `pattern(t)`
    chooses which one serves lock-step t.  Everything a step returns and, at the end, the whole ring must equal the
    oracle's (its generator is the plain-C one of include/bpp_gen.inl)."""
    refill = refill or max(1, depth - 3)
    spec = dict(bound=(2, 5), seed=seed, depth=depth, refill_every=refill, rng=gen, cache=depth - refill >= 4)
    set_knobs(stream_legacy=pattern(0))
    try:
        env = front(size, E, base, spec)
        ref = oracle.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E + 3, stream=spec)
        (obs, mask), (robs, rmask) = env.reset(), ref.reset()
        np.testing.assert_array_equal(obs, robs)
        rng = np.random.RandomState(3)
        for t in range(steps):
            set_knobs(stream_legacy=pattern(t + 1))
            a = oracle.sample_feasible(rmask, 5, t, env_id_base=base)
            a[rng.rand(E) < (fail if E >= 8 else 0.8)] = -1   # failures make bins race through their episodes
            r, o = env.step(a), ref.step(a)
            for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(r[k], o[k], err_msg="%s t=%d" % (k, t))
            rmask = o["mask"]
        np.testing.assert_array_equal(env.ring(), ref.pool)
        np.testing.assert_array_equal(env.gen_next_host(), ref.gen_next)
        assert int(ref.overflow[0]) == 0 and int(ref.state["episode"].max()) >= 1
    finally:
        set_knobs(stream_legacy=0)


class EmuKnobEnv(object):
    def __init__(self, emu, size, E, base, spec):
        self.env = emu.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E + 3, stream=spec)

    def reset(self):
        return self.env.reset()

    def step(self, a):
        return self.env.step(a)

    def ring(self):
        return self.env.pool

    def gen_next_host(self):
        return self.env.gen_next


@pytest.mark.parametrize("size,E,depth,steps,pattern", [
    ((10, 10, 10), 130, 8, 40, "fast"), ((10, 10, 10), 130, 8, 40, "alternate"), ((10, 10, 10), 67, 5, 30, "thirds"),
    ((10, 10, 10), 70, 6, 60, "plain"),          # the one-lane-per-bin kernel alone (twists its generators in place)
    ((20, 20, 20), 9, 5, 12, "fast"),            # ~1100 outputs per sequence: every job twists its generator
    ((20, 20, 20), 9, 5, 12, "alternate"),
    ((8, 12, 9), 40, 6, 30, "fast"),
    ((30, 30, 18), 2, 4, 2, "fast"),             # pending lists beyond the LDS part (peak ~120 entries, 80 in LDS)
])
def test_emulated_fast_and_plain_refill_interchangeable(emu, oracle, size, E, depth, steps, pattern):
    pat = {"fast": lambda t: 0, "alternate": lambda t: t % 2, "thirds": lambda t: (t // 3) % 2, "plain": lambda t: 1}[pattern]
    knob_check(lambda sz, n, base, spec: EmuKnobEnv(emu, sz, n, base, spec), oracle,
               lambda **kw: emu.set_knobs(**kw), size, E, depth, steps, pat)


def test_emulated_refill_after_many_episodes_without_refill(emu, oracle):
    """A caller that refills too rarely loses rows (contract: refill_every <= depth - 3) but never the stream position:
    generators advance by every sequence, the last `depth` ones are in the ring."""
    size, E, D, base = (10, 10, 10), 70, 4, 5
    spec = dict(bound=(2, 5), seed=11, depth=D, refill_every=1)
    env = emu.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E, stream=spec)
    ref = oracle.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E, stream=spec)
    for o in (env, ref):
        o.reset()
        o.refill_every = 10 ** 9                     # from here on only the explicit refill below
        o.state["episode"][: E // 2] += np.arange(E // 2, dtype=np.int32) % 9   # up to 8 episodes behind, more than depth
        o.refill()
    np.testing.assert_array_equal(env.pool, ref.pool)
    np.testing.assert_array_equal(env.gen_next, ref.gen_next)


@pytest.mark.gpu
@pytest.mark.parametrize("size,rot,E,steps,depth,refill,native", [((10, 10, 10), False, 4099, 120, 8, 5, False),
                                                                  ((10, 10, 10), True, 2000, 300, 8, 5, True),
                                                                  ((10, 10, 10), False, 65536, 60, 8, 5, True),
                                                                  ((20, 20, 20), False, 300, 400, 6, 3, True),
                                                                  ((6, 6, 6), False, 5000, 200, 19, 6, True),
                                                                  ((10, 10, 10), False, 20000, 150, 32, 14, True),
                                                                  # depth - refill >= 4: with the row cache (copier workgroups,
                                                                  # E not a multiple of their 1024 bins, forced failures)
                                                                  ((10, 10, 10), False, 4099, 120, 9, 4, False),
                                                                  ((10, 10, 10), True, 2000, 300, 12, 5, True),
                                                                  ((10, 10, 10), False, 65536, 60, 16, 6, True),
                                                                  ((20, 20, 20), False, 300, 400, 10, 3, True)])
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


class GpuKnobEnv(object):
    def __init__(self, size, E, base, spec):
        import bpp_amd
        self.env = bpp_amd.BppVecEnv(E, size, enable_rotation=False, stream=spec, env_id_base=base, env_id_total=base + E + 3)

    def reset(self):
        obs = self.env.reset()
        return obs.cpu().numpy(), self.env.location_masks.cpu().numpy()

    def step(self, a):
        r = self.env.step_tensors(np.asarray(a))
        out = {k: getattr(r, k).cpu().numpy() for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len")}
        out["reward"] = r.reward.cpu().numpy()[:, 0]
        return out

    def ring(self):
        return self.env.pool.cpu().numpy()

    def gen_next_host(self):
        return self.env.gen_next.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("size,E,depth,steps,pattern", [
    ((10, 10, 10), 5000, 8, 60, "fast"), ((10, 10, 10), 5000, 8, 60, "alternate"), ((10, 10, 10), 777, 5, 40, "thirds"),
    ((10, 10, 10), 3000, 6, 120, "plain"),
    ((20, 20, 20), 200, 5, 40, "fast"), ((20, 20, 20), 130, 5, 20, "alternate"), ((8, 12, 9), 300, 6, 40, "fast"),
    ((30, 30, 18), 70, 4, 3, "fast"),
])
def test_gpu_fast_and_plain_refill_interchangeable(oracle, size, E, depth, steps, pattern):
    import bpp_amd
    pat = {"fast": lambda t: 0, "alternate": lambda t: t % 2, "thirds": lambda t: (t // 3) % 2, "plain": lambda t: 1}[pattern]
    knob_check(GpuKnobEnv, oracle, lambda **kw: bpp_amd._lib.set_knobs(**kw), size, E, depth, steps, pat)


@pytest.mark.gpu
@pytest.mark.parametrize("size,E,depth,refill,steps", [((10, 10, 10), 65536, 16, 6, 64), ((10, 10, 10), 3001, 9, 3, 100),
                                                       ((20, 20, 20), 1024, 12, 4, 80)])
def test_gpu_refill_beside_the_lock_steps_changes_nothing(oracle, size, E, depth, refill, steps):
    """bpp_rollout_uniform_stream with depth >= 2 * refill_every + 3: refills run on the library's side stream while the
    next lock-steps execute.  Outputs, state records, generator progress and ring equal the serial schedule's and the
    oracle's; a knob turns the side stream off."""
    import torch
    import bpp_amd
    spec = dict(bound=(2, 5), seed=21, depth=depth, refill_every=refill)
    results = []
    for overlap in (1, 0):
        old = bpp_amd._lib.set_knobs(stream_overlap=overlap)
        try:
            env = bpp_amd.BppVecEnv(E, size, enable_rotation=False, stream=spec, env_id_base=17, env_id_total=17 + E)
            env.reset()
            acts = torch.empty(E, dtype=torch.int64, device=env.device)
            r = env.rollout_uniform(4, 0, steps, actions=acts)
            env.refill()
            torch.cuda.synchronize()
            results.append(dict(obs=r.obs.cpu().numpy(), mask=r.mask.cpu().numpy(), ep_ret=r.ep_ret.cpu().numpy(),
                                acts=acts.cpu().numpy(), state=env.state.cpu().numpy(), ring=env.pool.cpu().numpy(),
                                gen_next=env.gen_next.cpu().numpy()))
            assert int(env.stream_overflow.item()) == 0
        finally:
            bpp_amd._lib.set_knobs(**old)
    for k, v in results[0].items():
        np.testing.assert_array_equal(v, results[1][k], err_msg=k)
    ref = oracle.OracleEnv(None, size, False, E, env_id_base=17, env_id_total=17 + E, stream=spec)
    ref.reset()
    o, oa = oracle.rollout_uniform(ref, 4, 0, steps)
    ref.refill()
    np.testing.assert_array_equal(results[0]["acts"], oa)
    np.testing.assert_array_equal(results[0]["obs"], o["obs"])
    np.testing.assert_array_equal(results[0]["ring"], ref.pool)
    np.testing.assert_array_equal(results[0]["gen_next"], ref.gen_next)


def test_emulated_streaming_clone_continues_the_sources_item_stream(emu):
    """a bin cloned in streaming mode must keep playing the SOURCE's item stream from its own ring
    column.  Clone bin 0 into bin 1, let bin 0 fail through more than `depth` episodes (its refills rewrite its column),
    then play bin 1: it must show exactly the items random.Random(seed + id of bin 0) yields for the episode it was
    cloned in and the following ones.  Runs the product's own host-side copy (vec_env.copy_bin_records) on the
    emulated kernels' buffers."""
    import torch
    from bpp_amd.vec_env import copy_bin_records
    size, E, base, seed, depth = (10, 10, 10), 6, 40, 5, 6
    NOOP = -2 ** 63
    spec = dict(bound=(2, 5), seed=seed, depth=depth, refill_every=1)
    # mask_rule = 1 (Space.check_box): a position the mask shows is one the placement accepts
    env = emu.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E, stream=spec, mask_rule=1)
    obs, mask = env.reset()
    A = 100

    def shown(o, e):
        return tuple(int(o[e, (p + 1) * A]) for p in range(3))

    # bin 0 places two items, then gets cloned into bin 1
    for t in range(2):
        a = emu.sample_feasible(mask, 1, t, env_id_base=base)
        o = env.step(a)
        mask = o["mask"]
    assert not o["done"][0] and int(env.state["cursor"][0]) == 2 and int(env.state["episode"][0]) == 0
    hm = torch.from_numpy(env.hmap)
    st = torch.from_numpy(env.state.view(np.int32).reshape(E, 12))
    ring = torch.from_numpy(env.pool)
    mt = torch.from_numpy(env._mt.view(np.int32).reshape(E, -1))
    gn = torch.from_numpy(env.gen_next)
    copy_bin_records(hm, st, torch.tensor([0]), torch.tensor([1]), ring=ring, mt=mt, gen_next=gn, depth=depth)
    assert env.seq_cache is not None
    env.reset_seq_cache()          # state and ring were written behind the library's back (BppVecEnv.copy_bins does the same)
    assert int(env.state["seq"][1]) == int(env.state["seq"][0]) + 1        # same ring row index, the copy's own column
    # bin 0 fails 2 * depth times in a row (refill after every lock-step rewrites its column); every other bin waits
    for t in range(2 * depth):
        a = np.full(E, NOOP, np.int64)
        a[0] = -1
        o = env.step(a)
        assert o["done"][0]
    assert int(env.state["episode"][0]) == 2 * depth and int(env.state["episode"][1]) == 0
    # now play the clone at the first position its mask shows (all-ones when nothing fits: that placement fails and the
    # next sequence of the stream starts): every item it is shown must be the source stream's
    rng = random.Random(seed + base + 0)
    seqs = [sequences.cut2_sequence(size, (2, 5), rng) for _ in range(8)]
    a = np.full(E, NOOP, np.int64)
    o = env.step(a)                                   # observe
    episode, cursor = 0, 2
    for k in range(45):
        want = seqs[episode][cursor] if cursor < len(seqs[episode]) else tuple(size)
        assert shown(o["obs"], 1) == tuple(want), (k, episode, cursor)
        a[1] = int(np.flatnonzero(o["mask"][1])[0])
        o = env.step(a)
        episode, cursor = (episode + 1, 0) if o["done"][1] else (episode, cursor + 1)
    assert episode >= 2 and int(env.state["episode"][1]) == episode


@pytest.mark.gpu
def test_gpu_streaming_clone_and_preview_follow_the_sources_stream():
    """BppVecEnv.clone_into / preview in streaming mode on the HIP path (same scenario as the emulated test above, 64
    sources cloned at once while they race ahead by more than the ring depth)."""
    import torch
    import bpp_amd
    size, E, base, seed, depth = (10, 10, 10), 256, 900, 12, 8
    spec = dict(bound=(2, 5), seed=seed, depth=depth, refill_every=1)
    env = bpp_amd.BppVecEnv(E, size, stream=spec, env_id_base=base, env_id_total=base + E, mask_rule="space")
    env.reset()
    for t in range(2):
        env.step_tensors(env.sample_feasible(seed=1, step=t))
    src, dst = torch.arange(0, 64), torch.arange(128, 192)
    env.clone_into(src, dst)
    keep = env.state_numpy().copy()
    fail = torch.full((E,), env.NOOP, dtype=torch.int64)
    fail[src] = -1
    for t in range(2 * depth + 3):                     # the sources burn through > depth episodes, refilled every step
        r = env.step_tensors(fail)
    assert bool(r.done[src].all()) and int(env.state_numpy()["episode"][:64].min()) == 2 * depth + 3
    A = 100
    for b in (0, 17, 63):
        g = 128 + b
        rng = random.Random(seed + base + b)           # the SOURCE's generator
        seqs = [sequences.cut2_sequence(size, (2, 5), rng) for _ in range(6)]
        episode, cursor = int(keep["episode"][b]), int(keep["cursor"][b])
        assert episode == 0
        pv = env.preview(4)[g].cpu().numpy()
        want_pv = [(seqs[0] + [tuple(size)] * 4)[cursor + k] for k in range(4)]
        assert [tuple(int(v) for v in row) for row in pv] == want_pv
    r = env.observe()
    a = torch.full((E,), env.NOOP, dtype=torch.int64)
    track = {b: [0, int(keep["cursor"][b])] for b in (0, 17, 63)}
    gens = {b: random.Random(seed + base + b) for b in track}
    seqs = {b: [sequences.cut2_sequence(size, (2, 5), gens[b]) for _ in range(8)] for b in track}
    for k in range(40):
        obs, mask = r.obs.cpu().numpy(), r.mask.cpu().numpy()
        for b, (ep, cur) in track.items():
            g = 128 + b
            want = seqs[b][ep][cur] if cur < len(seqs[b][ep]) else tuple(size)
            assert tuple(int(obs[g, (p + 1) * A]) for p in range(3)) == tuple(want), (b, k)
            a[g] = int(np.flatnonzero(mask[g])[0])
        r = env.step_tensors(a)
        done = r.done.cpu().numpy()
        for b in track:
            track[b] = [track[b][0] + 1, 0] if done[128 + b] else [track[b][0], track[b][1] + 1]
    assert min(t[0] for t in track.values()) >= 1


@pytest.mark.gpu
def test_gpu_stream_checkpoints_are_validated_on_load():
    """a streaming checkpoint only loads into a streaming env of the same geometry, depth, seed and shard."""
    import bpp_amd
    size = (10, 10, 10)
    spec = dict(bound=(2, 5), seed=5, depth=6, refill_every=3)
    env = bpp_amd.BppVecEnv(64, size, stream=spec)
    env.reset()
    sd = env.state_dict()
    pool_env = bpp_amd.BppVecEnv(64, size, pool=bpp_amd.sequences.cut2_pool(size, 8, seed=0))
    with pytest.raises(ValueError, match="streaming"):
        pool_env.load_state_dict(sd)
    with pytest.raises(ValueError, match="pool-based"):
        env.load_state_dict(pool_env.state_dict())
    for other in (dict(spec, seed=6), dict(spec, depth=8, refill_every=3), dict(spec, bound=(2, 4))):
        with pytest.raises(ValueError, match="stream_spec"):
            bpp_amd.BppVecEnv(64, size, stream=other).load_state_dict(sd)
    with pytest.raises(ValueError, match="stream_spec"):
        bpp_amd.BppVecEnv(64, size, stream=spec, env_id_base=64, env_id_total=128).load_state_dict(sd)
    bpp_amd.BppVecEnv(64, size, stream=spec).load_state_dict(sd)      # the matching env loads
    # the ring / generator LAYOUT is versioned too -- a checkpoint that SAYS it has another layout is refused even
    # when every buffer happens to have the same shape.  ADVICE r5: a checkpoint without the key (ABI 12 / 13 wrote layout 2
    # without recording it; layout 1's generator records and rows have other widths) is told apart by its buffer shapes
    assert sd["stream_layout"] == bpp_amd.vec_env.STREAM_LAYOUT
    with pytest.raises(ValueError, match="layout"):
        bpp_amd.BppVecEnv(64, size, stream=spec).load_state_dict(dict(sd, stream_layout=1))
    keyless = {k: v for k, v in sd.items() if k != "stream_layout"}
    bpp_amd.BppVecEnv(64, size, stream=spec).load_state_dict(keyless)                 # same shapes: this build's layout
    with pytest.raises(ValueError, match="layout"):                                   # layout 1's wider generator records
        bpp_amd.BppVecEnv(64, size, stream=spec).load_state_dict(dict(keyless, stream_mt=sd["stream_mt"][:, :-8].clone()))
    # ... and the automatic row cache is only switched on where the tile step kernel (10x10 / 20x20 bins) keeps it
    assert bpp_amd.BppVecEnv(64, size, stream=dict(spec, depth=8, refill_every=3)).stream_spec["cache"] is True
    odd = bpp_amd.BppVecEnv(64, (7, 13, 8), stream=dict(bound=(2, 4), seed=5, depth=8))
    assert odd.stream_spec["cache"] is False and odd.refill_every == 5


@pytest.mark.gpu
def test_gpu_two_streaming_envs_in_one_process_own_their_side_streams():
    """ABI v14: the side stream + events of the overlapped refill schedule belong to the env (bpp_side_create), not to a per-device
    set inside the library: two streaming envs of one process, driven in turns, each equal what it does alone; close() releases."""
    import torch
    import bpp_amd
    size = (10, 10, 10)

    def make(seed):
        return bpp_amd.BppVecEnv(2048, size, stream=dict(bound=(2, 5), seed=seed, depth=32, refill_every=14, rng="counter"))

    def drive(envs, chunks=3, n=45):
        for e in envs:
            e.reset()
        for c in range(chunks):
            for e in envs:
                e.rollout_uniform(seed=3, step0=c * n, nsteps=n)
        torch.cuda.synchronize()
        return [(e.hmap.cpu().numpy().copy(), e.state.cpu().numpy().copy(), e.ep_acc.cpu().numpy().copy()) for e in envs]

    a, b = make(1), make(2)
    both = drive([a, b])
    assert a._side is not None and b._side is not None and a._side.value != b._side.value
    alone = drive([make(1)]) + drive([make(2)])
    for got, want in zip(both, alone):
        for x, y in zip(got, want):
            np.testing.assert_array_equal(x, y)
    assert int(both[0][1][:, 1].sum()) > 2048          # episodes went by
    a.close(), b.close()
    assert a._side is None and b._side is None


def test_emulated_row_cache_answers_the_look_aheads(emu, oracle):
    """bpp_batch.seq_cache is not just harmless, it works: under the benchmark's policy (fused uniform-feasible draw) all
    but the first look-aheads of a rollout are answered by the bins' cache lines -- the ring is read only while the first
    requests are on their way -- and the results still equal the oracle's, which knows nothing of the cache.  A zeroed
    cache (what a checkpoint restore leaves) recovers the same way."""
    import ctypes
    size, E, base = (10, 10, 10), 96, 500
    spec = dict(bound=(2, 5), seed=3, depth=12, refill_every=4, rng="counter", cache=True)
    env = emu.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E, stream=spec)
    ref = oracle.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E, stream=dict(spec, cache=False))
    env.reset(), ref.reset()
    stat_fn = emu.lib().emu_cache_stat
    stat_fn.restype = ctypes.POINTER(ctypes.c_longlong)
    stat = stat_fn()

    def run(step0, n):
        stat[0] = stat[1] = 0
        r, ra = emu.rollout_uniform(env, 9, step0, n)
        o, oa = oracle.rollout_uniform(ref, 9, step0, n)
        np.testing.assert_array_equal(ra, oa)
        for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=k)
        return int(stat[0]), int(stat[1])

    miss, hit = run(0, 40)
    assert miss + hit == 40 * E
    assert miss <= 2 * E + E // 8, (miss, hit)          # two lock-steps until the first lines arrive, then next to nothing
    miss, hit = run(40, 40)
    assert miss + hit == 40 * E and miss <= E // 8, (miss, hit)
    assert int(ref.state["episode"].min()) >= 3        # every bin moved through several rows meanwhile
    env.reset_seq_cache()                              # restore / clone: the caller zeroes the cache
    miss, hit = run(80, 30)
    assert 2 * E <= miss <= 2 * E + E // 8 and hit >= 27 * E - E // 8, (miss, hit)


@pytest.mark.gpu
@pytest.mark.parametrize("size,rot,E,gen", [((10, 10, 10), False, 5000, "counter"), ((10, 10, 10), True, 3000, "mt19937"),
                                            ((20, 20, 20), False, 1500, "counter")])
def test_gpu_row_cache_on_equals_off_through_kernel_switches_clones_and_restores(size, rot, E, gen):
    """bpp_batch.seq_cache changes nothing but speed: two streaming envs, one with and one without the row cache, stepped
    with the same actions -- one bin in six fails on purpose, so bins race through their rows (two rows in two steps: the
    case a line cannot answer), some bins are left alone (NOOP) -- while the launch shape changes under them (the generic
    and runtime-geometry kernels drop the cache lines, the tile kernel has to recover), bins are cloned and a checkpoint is
    restored (the host zeroes the cache).  Every output of every step and the final records must be equal."""
    import torch
    import bpp_amd
    from bpp_amd import _lib
    spec = dict(bound=(2, 5), seed=13, depth=12, refill_every=4, rng=gen)
    on = bpp_amd.BppVecEnv(E, size, enable_rotation=rot, stream=dict(spec, cache=True), env_id_base=7, env_id_total=7 + E)
    off = bpp_amd.BppVecEnv(E, size, enable_rotation=rot, stream=dict(spec, cache=False), env_id_base=7, env_id_total=7 + E)
    assert on._seq_cache is not None and off._seq_cache is None
    np.testing.assert_array_equal(on.reset().cpu().numpy(), off.reset().cpu().numpy())
    rng = np.random.RandomState(5)
    old = _lib.get_knobs()
    try:
        for t in range(150):
            _lib.set_knobs(force_generic=int(t % 23 in (7, 8)), legacy_fast=int(t % 31 == 12), tile_groups=(0, 2, 4)[(t // 40) % 3])
            a = on.sample_feasible(3, t).cpu().numpy()
            a[rng.rand(E) < 1.0 / 6] = -1
            a[rng.rand(E) < 0.05] = bpp_amd.BppVecEnv.NOOP
            r1, r0 = on.step_tensors(a), off.step_tensors(a)
            for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(getattr(r1, k).cpu().numpy(), getattr(r0, k).cpu().numpy(), err_msg="%s t=%d" % (k, t))
            if t == 60:
                src, dst = np.arange(0, 40), np.arange(100, 140)
                on.copy_bins(src, dst), off.copy_bins(src, dst)
            if t == 90:
                on.load_state_dict(on.state_dict())
    finally:
        _lib.set_knobs(**old)
    np.testing.assert_array_equal(on.state.cpu().numpy(), off.state.cpu().numpy())
    np.testing.assert_array_equal(on.pool.cpu().numpy(), off.pool.cpu().numpy())
    assert int(on.state[:, 1].max()) >= 8        # bpp_env_state.episode: bins went through many rows


@pytest.mark.parametrize("size,rot,E,gen", [((10, 10, 10), False, 70, "counter"), ((20, 20, 20), False, 9, "mt19937")])
def test_emulated_row_cache_survives_kernel_switches(emu, oracle, size, rot, E, gen):
    """The CPU half of test_gpu_row_cache_on_equals_off_...: the emulated product with a row cache against the oracle
    while the launch shape changes (kernels that drop the cache lines in between), with forced failures and NOOPs."""
    base = 40
    spec = dict(bound=(2, 5), seed=13, depth=12, refill_every=4, rng=gen, cache=True)
    env = emu.OracleEnv(None, size, rot, E, env_id_base=base, env_id_total=base + E, stream=spec)
    ref = oracle.OracleEnv(None, size, rot, E, env_id_base=base, env_id_total=base + E, stream=dict(spec, cache=False))
    (obs, mask), (robs, rmask) = env.reset(), ref.reset()
    np.testing.assert_array_equal(obs, robs)
    rng = np.random.RandomState(5)
    try:
        for t in range(70):
            emu.set_knobs(force_generic=int(t % 23 in (7, 8)), legacy_fast=int(t % 31 == 12), tile_groups=(0, 2, 4)[(t // 20) % 3])
            a = oracle.sample_feasible(rmask, 3, t, env_id_base=base)
            a[rng.rand(E) < 1.0 / 6] = -1
            a[rng.rand(E) < 0.05] = -2 ** 63
            r, o = env.step(a), ref.step(a)
            for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(r[k], o[k], err_msg="%s t=%d" % (k, t))
            rmask = o["mask"]
    finally:
        emu.set_knobs()
    for f in ("cursor", "episode", "seq", "item_cur", "item_next", "item_reset"):
        np.testing.assert_array_equal(env.state[f], ref.state[f], err_msg=f)
    assert int(ref.state["episode"].max()) >= 6
