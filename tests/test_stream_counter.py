"""The counter-based item supply (include/bpp_abi.h: BPP_STREAM_RNG_COUNTER): the reference's cutting algorithm
(envs/bpp0/mdCreator.py:59-138) on a stateless generator -- distribution parity, which is SURVEY 8(f2)'s bar for this row
("RNG streams can't match Python's random"; the exact MT19937 supply of tests/test_stream_supply.py exceeds it).

Three statements of the generator are compared bit for bit -- the device kernels (host SIMT emulator here, MI355X in the
`-m gpu` tests), the oracle library's plain C, and sequences.CounterRandom feeding the cutting restatement that is pinned to
the reference fixtures under MT19937 -- and the DISTRIBUTION is compared with the exact generator's."""
import random

import numpy as np
import pytest

from bpp_amd import sequences
from test_stream_supply import EmuKnobEnv, EmuStreamEnv, GpuKnobEnv, knob_check, spec_check


def test_counter_random_is_what_the_oracle_library_cuts(oracle):
    """Episode k of bin g == cut2_sequence(CounterRandom(seed, g, k)); Lemire's mapping is uniform on its range."""
    size, E, base, seed, D = (10, 10, 10), 9, 123456789012, 2 ** 40 + 17, 6
    spec = dict(bound=(2, 5), seed=seed, depth=D, refill_every=1, rng="counter")
    env = oracle.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E, stream=spec)
    env.reset()
    ring = env.pool.reshape(D, E, -1, 4)
    for e in range(E):
        for k in range(D):
            want = sequences.cut2_sequence(size, (2, 5), sequences.CounterRandom(seed, base + e, k))
            row = [tuple(int(v) for v in it[:3]) for it in ring[k, e][2:]]      # (behind the two look-ahead entries)
            assert row[:len(want)] == want and all(it == size for it in row[len(want):]), (e, k)
            nxt = sequences.cut2_sequence(size, (2, 5), sequences.CounterRandom(seed, base + e, k + 1))
            nn = sequences.cut2_sequence(size, (2, 5), sequences.CounterRandom(seed, base + e, k + 2))
            if k + 1 < D:   # the look-ahead entries: item 1 of the next row, item 0 of the row after (cut so far: rows 0 .. D - 1)
                assert tuple(int(v) for v in ring[k, e][0][:3]) == nxt[1]
            if k + 2 < D:
                assert tuple(int(v) for v in ring[k, e][1][:3]) == nn[0]
    r = sequences.CounterRandom(5, 7, 0)
    draws = np.array([r.below(7) for _ in range(70000)])
    assert set(draws) == set(range(7)) and abs(np.bincount(draws) / 10000.0 - 1.0).max() < 0.04
    assert [sequences.CounterRandom(1, 2, 3).randint(1, 10) for _ in range(3)] == [sequences.CounterRandom(1, 2, 3).randint(1, 10)] * 3


def test_counter_generator_has_the_exact_generators_distribution():
    """Same algorithm, same uniform draws -> same distribution of sequences: 6 000 sequences from each generator; sequence
    length agrees within ~4 sigma, and the total-variation distance between the item-size histograms (all items / first
    items) of the counter generator and the exact one is what two independent samples of the EXACT generator show among
    themselves (the noise floor, measured in the same test: ~0.014 / ~0.055)."""
    size, N = (10, 10, 10), 6000
    a = [sequences.cut2_sequence(size, (2, 5), sequences.CounterRandom(99, g, 0)) for g in range(N)]
    b = [sequences.cut2_sequence(size, (2, 5), random.Random(1000 + g)) for g in range(N)]
    c = [sequences.cut2_sequence(size, (2, 5), random.Random(50000 + g)) for g in range(N)]
    la, lb = np.array([len(s) for s in a]), np.array([len(s) for s in b])
    assert all(sum(x * y * z for x, y, z in s) == 1000 for s in a)
    assert abs(la.mean() - lb.mean()) < 4 * np.sqrt(la.var() / N + lb.var() / N) + 1e-9
    assert abs(la.std() - lb.std()) < 0.25

    def hist(seqs, first):
        h = np.zeros((6, 6, 6))
        for s in seqs:
            for x, y, z in (s[:1] if first else s):
                h[x, y, z] += 1
        return h / h.sum()
    for first in (False, True):
        tv_ab = 0.5 * np.abs(hist(a, first) - hist(b, first)).sum()
        tv_bc = 0.5 * np.abs(hist(b, first) - hist(c, first)).sum()
        assert tv_ab < 1.5 * tv_bc + 0.003, (first, tv_ab, tv_bc)


def test_counter_generator_words_are_uncorrelated_across_bins_episodes_and_draws():
    """the per-stream key goes through ONE 32-bit hash and every word is one fmix32 permutation of a 32-bit input,
    so all bins draw from the same 2^32-point function at structured offsets: quality rests on fmix32's avalanche.  What the
    structure could break is independence BETWEEN neighbours -- adjacent bins (stream ids g, g + 1), adjacent episodes,
    adjacent draws of one sequence.  Checked on 40 000 streams: the correlation of the raw words and the chi-square of the
    joint table of the draws' values (`below(7)`, what the cutting algorithm consumes) are at the noise level."""
    M = np.uint64(0xFFFFFFFF)

    def fmix(x):
        x = x & M
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x85ebca6b)) & M
        x ^= x >> np.uint64(13)
        x = (x * np.uint64(0xc2b2ae35)) & M
        x ^= x >> np.uint64(16)
        return x

    def words(seed0, sid, k, n):          # include/bpp_abi.h: word(n, 0) of episode k of stream id sid
        sid, k, n = (np.asarray(v, dtype=np.uint64) for v in (sid, k, n))
        h = fmix(np.uint64(seed0 & 0xFFFFFFFF) + np.uint64(0x9E3779B9))
        h = fmix(h ^ np.uint64(seed0 >> 32))
        h = fmix(h ^ (sid & M))
        h = fmix(h ^ (sid >> np.uint64(32)))
        klo = fmix(h ^ k)
        khi = fmix(klo + np.uint64(0x7F4A7C15) + k)
        return fmix(((klo + n * np.uint64(0x9E3779B9)) & M) ^ (khi & M))

    N = 40000
    g = np.arange(N, dtype=np.uint64)
    # the vectorised restatement is the generator the product uses
    for sid, k in ((0, 0), (7, 3), (2 ** 33 + 5, 11)):
        r = sequences.CounterRandom(99, sid, k)
        assert [int(words(99, [sid], [k], [n])[0]) for n in range(3)] == [r._word(0), (setattr(r, "n", 1), r._word(0))[1], (setattr(r, "n", 2), r._word(0))[1]]
    pairs = {"adjacent bins": (words(99, g, 0, 0), words(99, g + np.uint64(1), 0, 0)),
             "bins a stride of 65 536 apart (same local bin on the next GPU)": (words(99, g, 0, 0), words(99, g + np.uint64(65536), 0, 0)),
             "adjacent episodes": (words(99, g, 4, 0), words(99, g, 5, 0)),
             "adjacent draws": (words(99, g, 2, 6), words(99, g, 2, 7)),
             "bin g draw 1 vs bin g + 1 draw 0": (words(99, g, 0, 1), words(99, g + np.uint64(1), 0, 0))}
    for name, (a, b) in pairs.items():
        rho = np.corrcoef(a.astype(np.float64), b.astype(np.float64))[0, 1]
        assert abs(rho) < 4.5 / np.sqrt(N), (name, rho)
        da, db = (a * np.uint64(7)) >> np.uint64(32), (b * np.uint64(7)) >> np.uint64(32)     # below(7) without the rejection step
        table = np.zeros((7, 7))
        np.add.at(table, (da.astype(int), db.astype(int)), 1)
        chi2 = ((table - N / 49.0) ** 2 / (N / 49.0)).sum()
        assert chi2 < 48 + 5 * np.sqrt(2 * 48), (name, chi2)       # 48 degrees of freedom, 5 sigma
        assert abs(np.bincount((a >> np.uint64(31)).astype(int), minlength=2)[1] / N - 0.5) < 4.5 * 0.5 / np.sqrt(N), name


# which refill serves lock-step t (the stream_legacy knob): 0 = the default pipeline (counter generator: rows), 1 = the plain
# one-lane-per-bin kernel, 2 = scan / cut per bin / sort, 3 = rows with most rows going through the redo kernel
PATTERNS = {"fast": lambda t: 0, "alternate": lambda t: t % 2, "plain": lambda t: 1, "bins": lambda t: 2, "redo": lambda t: 3,
            "all four": lambda t: (0, 3, 2, 1)[t % 4]}


@pytest.mark.parametrize("size,rot,E,steps,depth,refill,native", [((10, 10, 10), False, 70, 60, 8, 5, False),
                                                                  ((10, 10, 10), True, 40, 64, 20, 8, True),
                                                                  ((20, 20, 20), False, 5, 30, 9, 3, True),
                                                                  ((6, 6, 6), False, 70, 60, 19, 6, True)])
def test_emulated_counter_supply_matches_oracle_and_python(emu, oracle, size, rot, E, steps, depth, refill, native):
    spec_check(lambda sz, r, n, base, spec: EmuStreamEnv(emu, sz, r, n, base, spec), oracle, size, rot, E, steps, depth, refill, native,
               gen="counter")


@pytest.mark.parametrize("size,E,depth,steps,pattern", [
    ((10, 10, 10), 130, 8, 40, "fast"), ((10, 10, 10), 130, 8, 40, "alternate"), ((10, 10, 10), 70, 6, 60, "plain"),
    ((20, 20, 20), 9, 5, 12, "alternate"), ((8, 12, 9), 40, 6, 30, "fast"),
    ((30, 30, 18), 2, 4, 2, "fast"),             # pending lists beyond the LDS part
    ((10, 10, 10), 130, 8, 40, "bins"),          # round 4's scan / cut per bin / sort
    ((10, 10, 10), 130, 8, 40, "redo"),          # rows pipeline with capacities that send most rows to its redo kernel
    ((12, 12, 12), 70, 7, 30, "redo"), ((10, 10, 10), 200, 9, 60, "all four"), ((15, 15, 15), 20, 6, 16, "fast"),
    ((16, 4, 4), 70, 6, 30, "fast"), ((16, 4, 4), 70, 6, 30, "redo"),      # a side >= 16: 8-bit box fields (the rows kernel's other instantiation)
])
def test_emulated_counter_fast_and_plain_refill_interchangeable(emu, oracle, size, E, depth, steps, pattern):
    pat = PATTERNS[pattern]
    knob_check(lambda sz, n, base, spec: EmuKnobEnv(emu, sz, n, base, spec), oracle, lambda **kw: emu.set_knobs(**kw), size, E, depth,
               steps, pat, gen="counter")


def test_emulated_counter_clone_continues_the_sources_stream(emu):
    """A bin cloned in counter mode carries the SOURCE's stream id with its record: the copy plays hash(seed, source id,
    episode) sequences, not its own."""
    import torch
    from bpp_amd.vec_env import copy_bin_records
    size, E, base, seed, depth = (10, 10, 10), 6, 40, 5, 6
    NOOP = -2 ** 63
    spec = dict(bound=(2, 5), seed=seed, depth=depth, refill_every=1, rng="counter")
    env = emu.OracleEnv(None, size, False, E, env_id_base=base, env_id_total=base + E, stream=spec, mask_rule=1)
    obs, mask = env.reset()
    A = 100
    for t in range(2):
        o = env.step(emu.sample_feasible(mask, 1, t, env_id_base=base))
        mask = o["mask"]
    assert not o["done"][0]
    hm, st = torch.from_numpy(env.hmap), torch.from_numpy(env.state.view(np.int32).reshape(E, 12))
    ring, mt, gn = torch.from_numpy(env.pool), torch.from_numpy(env._mt.view(np.int32).reshape(E, -1)), torch.from_numpy(env.gen_next)
    assert mt.shape[1] == 4
    copy_bin_records(hm, st, torch.tensor([0]), torch.tensor([1]), ring=ring, mt=mt, gen_next=gn, depth=depth)
    env.reset_seq_cache()                          # (BppVecEnv.copy_bins does the same)
    for t in range(2 * depth):                      # the source burns through > depth episodes; everybody else waits
        a = np.full(E, NOOP, np.int64)
        a[0] = -1
        assert env.step(a)["done"][0]
    seqs = [sequences.cut2_sequence(size, (2, 5), sequences.CounterRandom(seed, base + 0, k)) for k in range(8)]
    a = np.full(E, NOOP, np.int64)
    o = env.step(a)
    episode, cursor = 0, 2
    for k in range(45):
        want = seqs[episode][cursor] if cursor < len(seqs[episode]) else tuple(size)
        assert tuple(int(o["obs"][1, (p + 1) * A]) for p in range(3)) == tuple(want), (k, episode, cursor)
        a[1] = int(np.flatnonzero(o["mask"][1])[0])
        o = env.step(a)
        episode, cursor = (episode + 1, 0) if o["done"][1] else (episode, cursor + 1)
    assert episode >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("size,rot,E,steps,depth,refill,native", [((10, 10, 10), False, 4099, 120, 8, 5, False),
                                                                  ((10, 10, 10), True, 2000, 300, 8, 5, True),
                                                                  ((10, 10, 10), False, 65536, 60, 8, 5, True),
                                                                  ((20, 20, 20), False, 300, 400, 6, 3, True),
                                                                  ((6, 6, 6), False, 5000, 200, 19, 6, True),
                                                                  ((10, 10, 10), False, 20000, 150, 32, 14, True)])
def test_gpu_counter_supply_matches_oracle_and_python(oracle, size, rot, E, steps, depth, refill, native):
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

    spec_check(GpuStreamEnv, oracle, size, rot, E, steps, depth, refill, native, gen="counter")


@pytest.mark.gpu
@pytest.mark.parametrize("size,E,depth,steps,pattern", [
    ((10, 10, 10), 5000, 8, 60, "fast"), ((10, 10, 10), 5000, 8, 60, "alternate"), ((10, 10, 10), 3000, 6, 120, "plain"),
    ((20, 20, 20), 130, 5, 20, "alternate"), ((8, 12, 9), 300, 6, 40, "fast"), ((30, 30, 18), 70, 4, 3, "fast"),
    ((10, 10, 10), 5000, 8, 60, "bins"), ((10, 10, 10), 5000, 8, 60, "redo"), ((12, 12, 12), 700, 7, 40, "redo"),
    ((10, 10, 10), 4000, 9, 80, "all four"), ((15, 15, 15), 300, 6, 24, "fast"), ((10, 10, 10), 70000, 8, 20, "fast"),
    ((16, 4, 4), 900, 6, 40, "fast"), ((16, 4, 4), 900, 6, 40, "redo"),
])
def test_gpu_counter_fast_and_plain_refill_interchangeable(oracle, size, E, depth, steps, pattern):
    import bpp_amd
    pat = PATTERNS[pattern]
    knob_check(GpuKnobEnv, oracle, lambda **kw: bpp_amd._lib.set_knobs(**kw), size, E, depth, steps, pat, gen="counter")


@pytest.mark.gpu
def test_gpu_counter_checkpoint_and_stream_identity():
    """A counter-mode checkpoint resumes identically and does not load into an MT19937 env (and vice versa)."""
    import torch
    import bpp_amd
    size, E = (10, 10, 10), 300
    spec = dict(bound=(2, 5), seed=5, depth=6, refill_every=3, rng="counter")
    env = bpp_amd.BppVecEnv(E, size, stream=spec)
    assert tuple(env._mt.shape) == (E, 4)
    env.reset()
    env.rollout_uniform(seed=3, step0=0, nsteps=17)
    ckpt = env.state_dict()
    first = env.rollout_uniform(seed=3, step0=17, nsteps=23)
    want = {k: getattr(first, k).clone() for k in ("obs", "mask", "counter", "ratio", "ep_ret")}
    other = bpp_amd.BppVecEnv(E, size, stream=spec)
    other.load_state_dict(ckpt)
    again = other.rollout_uniform(seed=3, step0=17, nsteps=23)
    for k, v in want.items():
        assert torch.equal(getattr(again, k), v), k
    with pytest.raises(ValueError, match="stream_spec"):
        bpp_amd.BppVecEnv(E, size, stream=dict(spec, rng="mt19937")).load_state_dict(ckpt)
