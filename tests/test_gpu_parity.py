"""Parity tests proper: the HIP path (through the C ABI, via bpp_amd) vs the reference's golden
vectors and vs the oracle.  Everything is bit-exact: masks, heightmaps, observations, dones, counters
(integer/index work) AND rewards/ratios/episode returns (float64 arithmetic in the reference's
operation order; float32 only as the final cast), so every comparison is assert_array_equal."""
import os

import numpy as np
import pytest

from conftest import MASK_CASES, ROLLOUT_CASES, load_golden
from test_oracle_golden import check_masks, check_rollout

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bpp():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bpp_amd
    bpp_amd._lib.lib()
    return bpp_amd


@pytest.fixture(params=["tile", "rt", "generic"])
def kernel_path(request, bpp):
    """tile: the default dispatch (bpp_tile_kernel for the 10x10 / 20x20 bins, the runtime-geometry prefix-image
    kernel for other areas divisible by 4, the cell-scan kernel otherwise); rt: bpp_fast_kernel with runtime
    geometry wherever it applies; generic: the cell-scan kernel for everything."""
    old = bpp._lib.set_knobs(bins_per_wave=0, waves_per_group=0, xcd_remap=1, force_generic=int(request.param == "generic"),
                             legacy_fast=int(request.param == "rt"))
    yield request.param
    bpp._lib.set_knobs(**old)


@pytest.fixture
def knobs(bpp):
    """Set launch-shape knobs for one test (bpp_set_knobs), restored afterwards."""
    saved = bpp._lib.get_knobs()
    yield lambda **kw: bpp._lib.set_knobs(**kw)
    bpp._lib.set_knobs(**saved)


class GpuEnv(object):
    """numpy-in/numpy-out adapter so the golden replay helper can drive BppVecEnv."""

    def __init__(self, bpp, pool, size, rot, E, rule, **kw):
        self.env = bpp.BppVecEnv(E, size, enable_rotation=bool(rot), pool=pool,
                                 mask_rule="space" if rule else "utils", **kw)

    def reset(self):
        obs = self.env.reset()
        return obs.cpu().numpy(), self.env.location_masks.cpu().numpy()

    def step(self, actions):
        r = self.env.step_tensors(np.asarray(actions))
        out = {k: getattr(r, k).cpu().numpy() for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len")}
        out["reward"] = r.reward.cpu().numpy()[:, 0]
        return out


@pytest.mark.parametrize("case", ROLLOUT_CASES)
def test_gpu_rollout_matches_reference_golden(bpp, kernel_path, case):
    check_rollout(lambda pool, size, rot, E, rule: GpuEnv(bpp, pool, size, rot, E, rule), load_golden(case))


@pytest.mark.parametrize("case", MASK_CASES)
def test_gpu_masks_match_reference_golden(bpp, kernel_path, case):
    rules = {0: "utils", 1: "space"}
    check_masks(lambda obs, size, rot, rule: bpp.batched_mask_from_obs(obs, size, bool(rot), rules[rule]).cpu().numpy(),
                lambda hm, it, size, rot, rule: bpp.batched_mask_from_hmap(hm, it, size, bool(rot), rules[rule]).cpu().numpy(),
                load_golden(case))


def test_gpu_dropin_mask_functions(bpp):
    """acktr.utils-shaped single-row helpers: list[int] / int32 ndarray like the reference returns."""
    g = load_golden("masks_10")
    A = 100
    k = 5
    obs = np.concatenate([g["hmap"][k], np.full(A, g["items"][k, 0]), np.full(A, g["items"][k, 1]),
                          np.full(A, g["items"][k, 2])]).astype(np.float32)
    m = bpp.get_possible_position(obs, (10, 10, 10))
    assert isinstance(m, list) and m == g["mask_utils"][k].tolist()
    import torch
    mr = bpp.get_rotation_mask(torch.from_numpy(obs), (10, 10, 10))
    assert mr.dtype == np.int32 and np.array_equal(mr, g["mask_utils_rot"][k])


GEOMS = [((10, 10, 10), False, 1024, 11), ((10, 10, 10), True, 1000, 12), ((20, 20, 20), False, 160, 13),
         ((7, 13, 8), True, 333, 14), ((5, 4, 6), False, 77, 15), ((32, 32, 40), True, 9, 16),
         ((20, 20, 10), True, 130, 17), ((20, 20, 22), True, 67, 18), ((10, 10, 7), True, 203, 19),
         ((10, 10, 11), False, 50, 20), ((2, 2, 5), True, 40, 21), ((1, 3, 4), False, 33, 22), ((3, 1, 4), True, 20, 23),
         ((1, 1, 3), True, 17, 24), ((8, 128, 10), True, 50, 25), ((4, 255, 10), False, 40, 26), ((14, 72, 12), True, 9, 27),
         ((10, 10, 30), True, 300, 28), ((12, 8, 46), False, 90, 29), ((10, 10, 47), False, 40, 30)]      # K = 4 (round 6); H = 47: cell scan


@pytest.mark.parametrize("size,rot,E,seed", GEOMS)
def test_gpu_vs_oracle_random_rollout(bpp, oracle, kernel_path, size, rot, E, seed):
    """Longer seeded rollouts on many bins (incl. bin counts that are not a multiple of the per-wave
    group and non-multiple-of-4 areas): device sampler == oracle sampler, all outputs and the final
    state bit-exact."""
    rng = np.random.RandomState(seed)
    lo, hi = 1, max(2, min(size) // 2)
    seqs = [[tuple(rng.randint(lo, hi + 1, size=3)) for _ in range(rng.randint(3, 60))] for _ in range(37)]
    seqs[3][1] = (size[0], size[1], 1)       # bin-sized footprints: windows far above 31 cells
    seqs[5][0] = (size[0], max(1, size[1] - 1), 2)
    pool = bpp.sequences.pad_pool(seqs, size)
    for rule in ("utils", "space"):
        env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool, mask_rule=rule, env_id_base=5, env_id_total=E + 9)
        ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=5, env_id_total=E + 9,
                               mask_rule=1 if rule == "space" else 0)
        obs = env.reset()
        robs, rmask = ref.reset()
        np.testing.assert_array_equal(obs.cpu().numpy(), robs)
        np.testing.assert_array_equal(env.location_masks.cpu().numpy(), rmask)
        M = env.act_len
        for t in range(60):
            a = env.sample_feasible(seed=seed, step=t).cpu().numpy()
            np.testing.assert_array_equal(a, oracle.sample_feasible(rmask, seed, t, env_id_base=5))
            bad = rng.rand(E) < 0.05
            a[bad] = rng.randint(-2, M + 3, size=int(bad.sum()))
            r = env.step_tensors(a)
            o = ref.step(a)
            for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), o[k], err_msg="%s t=%d" % (k, t))
            np.testing.assert_array_equal(r.reward.cpu().numpy()[:, 0], o["reward"])
            rmask = o["mask"]
            if t == 30:   # VecEnv.reset() mid-run: every bin moves on to its next sequence
                np.testing.assert_array_equal(env.reset().cpu().numpy(), ref.reset()[0])
                rmask = ref.out["mask"].copy()
                np.testing.assert_array_equal(env.location_masks.cpu().numpy(), rmask)
        np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
        st = env.state_numpy()
        for f in ("cursor", "episode", "n_boxes", "vol_sum", "ep_ret", "ep_len", "seq", "item_cur", "item_next", "item_reset"):
            np.testing.assert_array_equal(st[f], ref.state[f], err_msg=f)


def test_gpu_reward_table_all_volumes(bpp, oracle):
    """float32(float64(vol/binvol)*10) for every reachable volume of the 10^3 and 20^3 item sets."""
    for size in ((10, 10, 10), (20, 20, 20)):
        items = [(x, y, z) for x in range(1, 8) for y in range(1, 8) for z in range(1, 8)]
        pool = bpp.sequences.pad_pool([[it] for it in items], size)
        E = len(items)
        env = bpp.BppVecEnv(E, size, pool=pool)
        ref = oracle.OracleEnv(pool, size, False, E)
        env.reset(), ref.reset()
        r = env.step_tensors(np.zeros(E, np.int64))
        o = ref.step(np.zeros(E, np.int64))
        np.testing.assert_array_equal(r.reward.cpu().numpy()[:, 0], o["reward"])
        np.testing.assert_array_equal(r.ratio.cpu().numpy(), o["ratio"])
        binvol = float(np.prod(size))
        expect = np.array([np.float32(np.float64(x * y * z / binvol) * 10) for x, y, z in items], np.float32)
        np.testing.assert_array_equal(o["reward"], expect)


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), False, 65536), ((10, 10, 10), True, 65536),
                                         ((20, 20, 20), False, 32768)])
def test_gpu_full_size_properties_and_slices(bpp, oracle, kernel_path, size, rot, E):
    """BASELINE.json's full sizes: size-independent invariants on ALL bins plus bit-exact oracle
    replays of three 192-bin slices (bins are independent and sequences are keyed by global bin id,
    so a slice can be replayed in isolation with env_id_base/env_id_total)."""
    import torch
    W, L, H = size
    A = W * L
    pool = bpp.sequences.cut2_pool(size, 64 if A > 100 else 512, seed=1)
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool)
    slices = [0, E // 2 - 96, E - 192]
    refs = [oracle.OracleEnv(pool, size, rot, 192, env_id_base=s, env_id_total=E) for s in slices]
    obs = env.reset()
    for ref, s in zip(refs, slices):
        robs, rmask = ref.reset()
        np.testing.assert_array_equal(obs[s:s + 192].cpu().numpy(), robs)
        np.testing.assert_array_equal(env.location_masks[s:s + 192].cpu().numpy(), rmask)
    ret_sum = torch.zeros(E, dtype=torch.float64, device=obs.device)
    n_done = 0
    for t in range(24):
        a = env.sample_feasible(seed=9, step=t)
        if t % 5 == 4:
            a[::3] = A - 1      # mostly infeasible corner placements
        prev_mask = env.location_masks.clone()
        r = env.step_tensors(a)
        for ref, s in zip(refs, slices):
            o = ref.step(a[s:s + 192].cpu().numpy())
            for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(getattr(r, k)[s:s + 192].cpu().numpy(), o[k], err_msg="%s t=%d" % (k, t))
            np.testing.assert_array_equal(r.reward[s:s + 192, 0].cpu().numpy(), o["reward"])
        done = r.done.bool()
        o4 = r.obs.view(E, 4, A)
        # plane 0 is the int32 heightmap, planes 1-3 are constant = next item
        assert torch.equal(o4[:, 0], env.hmap.float())   # uint8 state == float32 plane 0
        assert bool((o4[:, 1:] == o4[:, 1:, :1]).all())
        assert int(env.hmap.max()) <= H and int(env.hmap.min()) >= 0
        # masks are 0/1, never all-zero
        assert bool(((r.mask == 0) | (r.mask == 1)).all()) and bool((r.mask.sum(1) > 0).all())
        # a finished bin shows an empty map; reward is 0 exactly on terminal steps (items have volume)
        assert bool((env.hmap[done] == 0).all())
        assert torch.equal(r.reward[:, 0] == 0, done)
        # the action taken was feasible under rule U for everything we did not overwrite; rule U and
        # rule S coincide for items <= 5x5 so those steps cannot terminate unless the
        # mask was the all-ones fallback -- or unless the action was index A, "rotated at (0,0)", which
        # the mask allows but bin3D.py:102's strict `idx > area` decodes as an un-rotated out-of-bounds
        # drop
        if t % 5 != 4:
            fallback = prev_mask.sum(1) == prev_mask.shape[1]
            assert bool((~done | fallback | (a == A)).all())
        # Monitor: episode return == sum of rewards; == 10 * final ratio up to float64 rounding
        ret_sum += r.reward[:, 0].double()
        if bool(done.any()):
            assert torch.allclose(r.ep_ret[done], 10.0 * r.ratio[done], rtol=0, atol=1e-9)
            assert torch.allclose(r.ep_ret[done], ret_sum[done], rtol=0, atol=1e-5)  # float32 rewards summed
            n_done += int(done.sum())
        ret_sum[done] = 0
    assert n_done > 0


def test_gpu_fused_episode_stats_and_standalone_kernel(bpp, oracle):
    """The statistics kept inside bpp_step (per-bin accumulator rows, no atomics, fixed-order reduction) and the
    stand-alone bpp_episode_stats kernel equal the oracle's sums BIT FOR BIT (include/bpp_abi.h fixes the order)."""
    size, E = (10, 10, 10), 3000
    pool = bpp.sequences.cut2_pool(size, 64, seed=5)
    env = bpp.BppVecEnv(E, size, enable_rotation=True, pool=pool)
    ref = oracle.OracleEnv(pool, size, True, E)
    env.reset(), ref.reset()
    standalone = bpp.EpisodeStats(env.device)
    acts = []
    for t in range(30):
        a = env.sample_feasible(seed=2, step=t)
        r = env.step_tensors(a)
        standalone.update(r)
        acts.append(a.cpu().numpy())
        ref.step(acts[-1])
    np.testing.assert_array_equal(env.ep_acc.cpu().numpy(), ref.ep_acc)       # every bin's own row
    fused = env.episode_stats().cpu().numpy()
    want = ref.episode_stats()
    assert want[3] > 100
    np.testing.assert_array_equal(fused, want)
    assert want[3] == ref.ep_acc[:, 3].sum() and want[2] == ref.ep_acc[:, 2].sum()
    np.testing.assert_allclose(want[:2], ref.ep_acc[:, :2].sum(0), rtol=1e-12)
    # the stand-alone kernel reduces the finished bins of every step in the same fixed order
    acc = np.zeros(4)
    ref2 = oracle.OracleEnv(pool, size, True, E)
    ref2.reset()
    for t in range(30):
        o = ref2.step(acts[t])
        oracle.episode_stats(o["done"], o["ep_ret"], o["ratio"], o["ep_len"], acc)
    np.testing.assert_array_equal(standalone.acc.cpu().numpy(), acc)
    s = bpp.EpisodeStats(env.device).collect(env).summary()
    assert s["episodes"] == int(want[3]) and abs(s["mean_ratio"] - want[1] / want[3]) < 1e-12
    assert float(env.episode_stats().sum()) == 0.0                      # collect() cleared the accumulator


def test_gpu_native_rollout_driver_matches_oracle(bpp, oracle):
    """bpp_rollout_uniform (N lock-steps enqueued by one native call) == the oracle's same driver."""
    size, E = (10, 10, 10), 4096
    pool = bpp.sequences.cut2_pool(size, 128, seed=7)
    for rot in (False, True):
        env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool)
        ref = oracle.OracleEnv(pool, size, rot, E)
        env.reset(), ref.reset()
        r = env.rollout_uniform(seed=11, step0=3, nsteps=37)
        o, last_a = oracle.rollout_uniform(ref, 11, 3, 37)
        for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), o[k], err_msg=k)
        np.testing.assert_array_equal(r.reward.cpu().numpy()[:, 0], o["reward"])
        np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
        np.testing.assert_array_equal(env.state_numpy()["episode"], ref.state["episode"])


@pytest.mark.parametrize("size,rot,E,steps", [((20, 20, 20), False, 4096, 150), ((20, 20, 20), True, 1024, 110), ((20, 20, 22), False, 1024, 110)])
def test_gpu_tall_20x20_bins_two_phase_scan_matches_oracle(bpp, oracle, size, rot, E, steps):
    """The 20x20 kernel holds a ONE-word prefix image: a bin taller than 11 is scanned in two phases (upper word, then
    lower word).  Long rollouts so that a large share of the bins is tall at the end, every lock-step's final state and
    outputs compared with the oracle; the share of tall bins is asserted so that the path cannot go untested."""
    pool = bpp.sequences.cut2_pool(size, 64, seed=13)
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool)
    ref = oracle.OracleEnv(pool, size, rot, E)
    env.reset(), ref.reset()
    tall_seen, done_steps = 0, 0
    for chunk in (steps // 3, steps // 3, steps - 2 * (steps // 3)):
        r = env.rollout_uniform(seed=17, step0=done_steps, nsteps=chunk)
        o, _ = oracle.rollout_uniform(ref, 17, done_steps, chunk)
        done_steps += chunk
        for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), o[k], err_msg="%s after %d" % (k, done_steps))
        np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
        st = env.state_numpy()
        np.testing.assert_array_equal(st["hmax"], ref.state["hmax"])
        np.testing.assert_array_equal(st["hmax"], ref.hmap.max(1))          # the record's hmax IS the map's maximum
        tall_seen = max(tall_seen, int((st["hmax"] > 11).sum()))
    np.testing.assert_array_equal(env.episode_stats().cpu().numpy(), ref.episode_stats())
    assert tall_seen > E // 20, tall_seen


def test_gpu_dropin_make_vec_envs_returns_reference_types(bpp, tmp_path):
    """The reference-shaped entry point: make_vec_envs(...) -> step() -> (obs device f32, reward CPU f32
    [N,1], done numpy bool, infos of dicts) exactly like VecPyTorch (acktr/envs.py:170-193), replayed
    against the golden rollout recorded from the reference stack."""
    import types
    import torch
    g = load_golden("rollout_cut2_10_rot")
    E = g["actions"].shape[1]
    args = types.SimpleNamespace(container_size=(10, 10, 10), enable_rotation=True, data_type="cut2", box_size_set=None)
    log_dir = str(tmp_path / "log")
    envs = bpp.make_vec_envs("Bpp-v0", 1, E, 1.0, log_dir, "cuda:0", False, args=args, pool=g["pool"])
    assert envs.num_envs == E and envs.action_space.n == 200 and envs.observation_space.shape == (400,)
    assert envs.action_space.__class__.__name__ == "Discrete"
    obs = envs.reset()
    assert obs.dtype == torch.float32 and obs.is_cuda and tuple(obs.shape) == (E, 400)
    held = []
    for t in range(40):
        obs, reward, done, infos = envs.step(torch.from_numpy(g["actions"][t]).unsqueeze(1))
        held.append(obs)
        assert reward.dtype == torch.float32 and not reward.is_cuda and tuple(reward.shape) == (E, 1)
        assert isinstance(done, np.ndarray) and done.dtype == bool and len(infos) == E
        np.testing.assert_array_equal(obs.cpu().numpy(), g["obs"][t].astype(np.float32))
        np.testing.assert_array_equal(reward.numpy()[:, 0], g["reward"][t])
        np.testing.assert_array_equal(done, g["done"][t].astype(bool))
        for e in range(E):
            i = infos[e]
            assert i["counter"] == g["counter"][t][e] and i["ratio"] == g["ratio"][t][e]
            assert ("episode" in i.keys()) == bool(g["done"][t][e]) and "bad_transition" not in i.keys()
            if done[e]:
                assert i["episode"]["r"] == g["ep_r"][t][e] and i["episode"]["l"] == g["ep_l"][t][e]
                assert i["mask"].shape == (200,)
        m = bpp.get_rotation_mask(obs[0], (10, 10, 10))
        np.testing.assert_array_equal(m, g["mask"][t][0])
    # fresh_outputs (default of the factory): results of earlier steps are still intact
    np.testing.assert_array_equal(held[5].cpu().numpy(), g["obs"][5].astype(np.float32))
    envs.close()
    # log_dir (acktr/envs.py:54-58): <log_dir>/0.monitor.csv holds Monitor's row of every finished episode (monitor.py:58-72)
    lines = open(os.path.join(log_dir, "0.monitor.csv"), newline="").read().split("\r\n")
    assert lines[0].startswith("# {") and lines[0].endswith("r,l,t,bin") and lines[-1] == ""
    rows = [ln.split(",") for ln in lines[1:-1]]
    want = [(g["ep_r"][t][e], g["ep_l"][t][e], e) for t in range(40) for e in range(E) if g["done"][t][e]]
    assert len(rows) == len(want) > 0
    for row, (r, l, e) in zip(rows, want):
        assert float(row[0]) == r and int(row[1]) == l and int(row[3]) == e and float(row[2]) >= 0.0


def test_gpu_dropin_infos_read_late_are_still_the_steps_own(bpp):
    """a loop that collects `infos` over a rollout and reads them afterwards.  With the factory's
    fresh_outputs=True every infos object keeps describing ITS step (finished and running bins alike, fetched lazily
    long after later steps ran); with shared output buffers a late first access raises instead of showing another
    step's numbers."""
    import types
    import torch
    g = load_golden("rollout_cut2_10_rot")
    E = g["actions"].shape[1]
    args = types.SimpleNamespace(container_size=(10, 10, 10), enable_rotation=True, data_type="cut2", box_size_set=None)
    envs = bpp.make_vec_envs("Bpp-v0", 1, E, 1.0, None, "cuda:0", False, args=args, pool=g["pool"])
    envs.reset()
    kept = []
    for t in range(30):
        _, reward, done, infos = envs.step(torch.from_numpy(g["actions"][t]).unsqueeze(1))
        kept.append((reward, done, infos))
    assert sum(int(d.sum()) for _, d, _ in kept) > 0
    for t, (reward, done, infos) in enumerate(kept):            # nothing was looked at until now
        np.testing.assert_array_equal(reward.numpy()[:, 0], g["reward"][t])
        np.testing.assert_array_equal(done, g["done"][t].astype(bool))
        for e in range(E):
            i = infos[e]
            assert i["counter"] == g["counter"][t][e] and i["ratio"] == g["ratio"][t][e], (t, e)
            if done[e]:
                assert i["episode"]["r"] == g["ep_r"][t][e] and i["episode"]["l"] == g["ep_l"][t][e]
    shared = bpp.BppVecEnv(E, (10, 10, 10), enable_rotation=True, pool=g["pool"])       # fresh_outputs=False
    shared.reset()
    _, _, _, first = shared.step(torch.from_numpy(g["actions"][0]))
    _, _, _, second = shared.step(torch.from_numpy(g["actions"][1]))
    assert second[0]["counter"] == g["counter"][1][0]
    with pytest.raises(RuntimeError, match="fresh_outputs"):
        first[0]


def test_gpu_dropin_finished_infos_through_the_native_gather(bpp, oracle):
    """step() on 8 192 bins with the fused draw: infos.episodes() (bpp_gather_finished, one launch) and the dicts of the
    finished bins equal the oracle's numbers of the same step; the staging cap and the legacy-checkpoint path of ADVICE r3."""
    import torch
    size, E = (10, 10, 10), 8192
    pool = bpp.sequences.cut2_pool(size, 256, seed=3)
    env = bpp.BppVecEnv(E, size, pool=pool, fresh_outputs=True)
    ref = oracle.OracleEnv(pool, size, False, E)
    env.reset()
    ref.reset()
    a = env.sample_feasible(seed=9, step=0)
    seen = 0
    for t in range(25):
        a_np = a.cpu().numpy().copy()
        obs, reward, done, infos = env.step(a, sample=(9, t + 1, a))
        o = ref.step(a_np)
        np.testing.assert_array_equal(done, o["done"].astype(bool))
        np.testing.assert_array_equal(reward.numpy()[:, 0], o["reward"])
        np.testing.assert_array_equal(a.cpu().numpy(), oracle.sample_feasible(o["mask"], 9, t + 1))
        ep = infos.episodes()
        d = np.flatnonzero(o["done"])
        np.testing.assert_array_equal(ep["bins"], d)
        np.testing.assert_array_equal(ep["r"], np.round(o["ep_ret"][d], 6))
        np.testing.assert_array_equal(ep["l"], o["ep_len"][d])
        np.testing.assert_array_equal(ep["ratio"], o["ratio"][d])
        np.testing.assert_array_equal(ep["counter"], o["counter"][d])
        for i in d[:5]:
            assert infos[int(i)]["episode"]["r"] == round(float(o["ep_ret"][i]), 6) and infos[int(i)]["ratio"] == o["ratio"][i]
        seen += d.size
    assert seen > E
    # a loop that keeps every step's reward / done views: page-locked staging stops growing at MAX_STAGING, results stay right
    kept = []
    with pytest.warns(RuntimeWarning, match="page-locked"):
        for t in range(env.MAX_STAGING + 4):
            a_np = a.cpu().numpy().copy()
            kept.append(env.step(a, sample=(9, 100 + t, a))[1:3])
            o = ref.step(a_np)
            np.testing.assert_array_equal(kept[-1][0].numpy()[:, 0], o["reward"])
            np.testing.assert_array_equal(kept[-1][1], o["done"].astype(bool))
    assert len(env._stage_pool) == env.MAX_STAGING
    # checkpoints: a legacy one (no format, hmax word zero, slotted `stats`) is repaired on load
    sd = env.state_dict()
    assert sd["format"] == 1
    legacy = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items() if k not in ("format", "ep_acc")}
    legacy["state"][:, 11] = 0
    legacy["stats"] = torch.zeros(4)
    env2 = bpp.BppVecEnv(E, size, pool=pool)
    with pytest.warns(RuntimeWarning, match="legacy checkpoint"):
        env2.load_state_dict(legacy)
    np.testing.assert_array_equal(env2.state_numpy()["hmax"], ref.state["hmax"])
    np.testing.assert_array_equal(env2.state_numpy()["hmax"], env.hmap.max(1).values.cpu().numpy())
    assert float(env2.ep_acc.abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("eager", [False, True])
def test_gpu_dropin_step_waits_on_its_completion_word(bpp, oracle, eager):
    """step_wait() spins on the word bpp_mark leaves in the step's page-locked buffer behind the step kernel (and the eager
    gather) instead of synchronising the stream (ABI v15): same reward / done / infos as the oracle and as an env that
    synchronises, step after step; a word nobody marked is reported, not waited for."""
    import ctypes
    import torch
    size, E = (10, 10, 10), 20000
    pool = bpp.sequences.cut2_pool(size, 512, seed=5)
    envs = [bpp.BppVecEnv(E, size, pool=pool, eager_infos=eager) for _ in range(2)]
    envs[1].spin_wait = False
    ref = oracle.OracleEnv(pool, size, False, E)
    for e in envs:
        e.reset()
    ref.reset()
    a = envs[0].sample_feasible(seed=4, step=0)
    for t in range(40):
        a_np = a.cpu().numpy().copy()
        o = ref.step(a_np)
        outs = [e.step(torch.from_numpy(a_np).to(e.device)) for e in envs]
        for obs, reward, done, infos in outs:
            np.testing.assert_array_equal(done, o["done"].astype(bool))
            np.testing.assert_array_equal(reward.numpy()[:, 0], o["reward"])
            ep = infos.episodes()
            d = np.flatnonzero(o["done"])
            np.testing.assert_array_equal(ep["bins"], d)
            np.testing.assert_array_equal(ep["ratio"], o["ratio"][d])
        assert envs[0]._mark == t + 1 and envs[1]._mark == 0
        a = torch.from_numpy(oracle.sample_feasible(o["mask"], 4, t + 1)).to(envs[0].device)
    lib = bpp._lib.lib()
    word = torch.zeros(2, dtype=torch.int32).pin_memory()
    torch.cuda.synchronize()
    assert lib.bpp_wait_mark(word.data_ptr(), 5, envs[0]._stream_ptr()) != 0 and b"flag" in lib.bpp_last_error()
    assert lib.bpp_mark(word.data_ptr(), 5, envs[0]._stream_ptr()) == 0
    assert lib.bpp_wait_mark(word.data_ptr(), 5, envs[0]._stream_ptr()) == 0 and int(word[0]) == 5 and int(word[1]) == 0


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), False, 4099), ((10, 10, 10), True, 1000), ((20, 20, 20), False, 301),
                                         ((20, 20, 10), True, 130), ((7, 13, 8), True, 97)])
def test_gpu_fused_next_action_equals_standalone_sampler(bpp, oracle, kernel_path, size, rot, E):
    """bpp_step_out.next_action (drawn inside the step kernel from the LDS mask) == bpp_sample_feasible
    on the mask the step wrote == the oracle, on both kernel paths; and whole rollouts driven by it match."""
    import torch
    pool = bpp.sequences.cut2_pool(size, 32, seed=3, bound=(2, min(5, min(size) // 2)))
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool, env_id_base=7, env_id_total=E + 7)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=7, env_id_total=E + 7)
    env.reset(), ref.reset()
    a = env.sample_feasible(seed=4, step=0)
    nxt = torch.empty_like(a)
    for t in range(25):
        r = env.step_tensors(a, sample=(4, t + 1, nxt))
        o = ref.step(a.cpu().numpy())
        np.testing.assert_array_equal(r.mask.cpu().numpy(), o["mask"])
        np.testing.assert_array_equal(nxt.cpu().numpy(), oracle.sample_feasible(o["mask"], 4, t + 1, env_id_base=7))
        np.testing.assert_array_equal(nxt.cpu().numpy(), env.sample_feasible(seed=4, step=t + 1).cpu().numpy())
        a, nxt = nxt, a
    r = env.rollout_uniform(seed=9, step0=100, nsteps=12)
    o, _ = oracle.rollout_uniform(ref, 9, 100, 12)
    for k in ("obs", "mask", "done", "counter", "ep_ret"):
        np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), o[k], err_msg=k)


def test_gpu_lookahead_support_copy_bins_and_preview(bpp, oracle):
    """branching a bin (copy_bins) reproduces deepcopy semantics -- the copy steps exactly
    like an oracle env that replays the same history -- and preview(k) lists the upcoming items."""
    import torch
    size, E = (10, 10, 10), 64
    pool = bpp.sequences.cut2_pool(size, 16, seed=8)
    env = bpp.BppVecEnv(E, size, enable_rotation=True, pool=pool)
    ref = oracle.OracleEnv(pool, size, True, E)
    env.reset(), ref.reset()
    for t in range(6):
        a = env.sample_feasible(seed=1, step=t)
        env.step_tensors(a), ref.step(a.cpu().numpy())
    pv = env.preview(3).cpu().numpy()
    st = ref.state
    for e in range(E):
        for j in range(3):
            c = min(int(st["cursor"][e]) + j, pool.shape[1] - 1)
            assert pv[e, j].tolist() == pool[st["seq"][e], c, :3].tolist()
    assert torch.equal(env.heightmaps().reshape(E, -1), env.hmap.int())
    # branch: bins 32..63 become copies of bins 0..31, then both halves take the same actions
    env.copy_bins(torch.arange(32), torch.arange(32, 64))
    a = env.sample_feasible(seed=2, step=0)   # masks of the copies are stale; draw for the sources only
    a[32:] = a[:32]
    r = env.step_tensors(a)
    o = ref.step(np.concatenate([a[:32].cpu().numpy(), np.zeros(32, np.int64)]))
    for k in ("obs", "mask", "done", "counter", "ratio"):
        v = getattr(r, k).cpu().numpy()
        np.testing.assert_array_equal(v[32:], v[:32], err_msg=k)
        np.testing.assert_array_equal(v[:32], o[k][:32], err_msg=k)


@pytest.mark.parametrize("size,rot,E,steps", [((10, 10, 10), False, 2048, 1500), ((10, 10, 10), True, 1024, 800),
                                               ((20, 20, 20), False, 256, 400)])
def test_gpu_long_soak_matches_oracle(bpp, oracle, size, rot, E, steps):
    """Long horizon: thousands of lock-steps (hundreds of episodes per bin, pool rows wrapping around many
    times) through the native driver with the in-kernel draw; final observation, mask, byte heightmaps,
    complete state records (incl. the float64 Monitor sums and the item cache) and the episode statistics
    must equal the oracle's."""
    pool = bpp.sequences.cut2_pool(size, 7, seed=21)          # tiny pool -> heavy wrap-around
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool, env_id_base=3, env_id_total=E + 11)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=3, env_id_total=E + 11)
    env.reset(), ref.reset()
    done_steps = 0
    for chunk in (steps // 3, steps - steps // 3):
        r = env.rollout_uniform(seed=5, step0=done_steps, nsteps=chunk)
        o, _ = oracle.rollout_uniform(ref, 5, done_steps, chunk)
        done_steps += chunk
        for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), o[k], err_msg=k)
    np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
    st = env.state_numpy()
    for f in st.dtype.names:
        if f != "pad":
            np.testing.assert_array_equal(st[f], ref.state[f], err_msg=f)
    np.testing.assert_array_equal(env.ep_acc.cpu().numpy(), ref.ep_acc)
    got, want = env.episode_stats().cpu().numpy(), ref.episode_stats()
    np.testing.assert_array_equal(env.episode_stats(wide=False).cpu().numpy(), want)   # one-workgroup form: the same bits
    np.testing.assert_array_equal(got, want)
    assert want[3] > E * steps / 60


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), False, 4099), ((10, 10, 10), True, 1027), ((20, 20, 20), False, 515)])
def test_gpu_rollout_over_rotating_output_sets(bpp, oracle, size, rot, E):
    """bpp_rollout_uniform_sets (bench.py's driver, incl. its past-the-Infinity-Cache leg): lock-step t writes output set
    t mod n, every lock-step draws the next actions in-kernel, resume=True enqueues step kernels only == the oracle."""
    import torch
    pool = bpp.sequences.cut2_pool(size, 64, seed=8)
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool, env_id_base=11, env_id_total=E + 11)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=11, env_id_total=E + 11)
    env.reset(), ref.reset()
    actions = torch.empty(E, dtype=torch.int64, device=env.device)
    ra, r_last, t = None, None, 0
    for n, nsets in ((7, 3), (5, 2), (9, 1), (6, 4)):
        sets = env.output_sets(nsets) if nsets > 1 else None
        r = env.rollout_uniform_sets(5, t, n, actions, sets=sets, resume=t > 0)
        rs, ra = oracle.rollout_uniform_sets(ref, 5, t, n, nsets, resume=t > 0, actions=ra,
                                             first_mask=r_last["mask"] if r_last else None)
        r_last = rs[(n - 1) % nsets]
        t += n
        np.testing.assert_array_equal(actions.cpu().numpy(), ra)
        for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), r_last[k], err_msg=k)
        if sets is not None:
            for j in range(min(n, nsets)):
                np.testing.assert_array_equal(sets[j][0]["mask"].cpu().numpy(), rs[j]["mask"])
    np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
    np.testing.assert_array_equal(env.ep_acc.cpu().numpy(), ref.ep_acc)
    np.testing.assert_array_equal(env.episode_stats().cpu().numpy(), ref.episode_stats())


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), False, 4099), ((10, 10, 10), True, 1027), ((20, 20, 20), False, 515)])
def test_gpu_epsilon_variant_of_the_rollout(bpp, oracle, size, rot, E):
    """SURVEY 8d's failure-path variant (bench.py's epsilon leg): bpp_rollout_uniform_sets with BPP_ROLLOUT_EPS == the
    oracle's statement of it; bpp_epsilon_override alone == the oracle's; eps = 0 enqueues nothing."""
    import torch
    pool = bpp.sequences.cut2_pool(size, 64, seed=8)
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool, env_id_base=11, env_id_total=E + 11)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=11, env_id_total=E + 11)
    a0 = torch.arange(E, dtype=torch.int64, device=env.device) % env.act_len
    np.testing.assert_array_equal(env.epsilon_override(a0.clone(), 7, 3, 0.0).cpu().numpy(), a0.cpu().numpy())
    got = env.epsilon_override(a0.clone(), 7, 3, 0.2).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.epsilon_override(a0.cpu().numpy(), env.act_len, 7, 3, 0.2, env_id_base=11))
    assert 0.1 * E < np.count_nonzero(got != a0.cpu().numpy()) < 0.3 * E
    env.reset(), ref.reset()
    actions = torch.empty(E, dtype=torch.int64, device=env.device)
    ra, r_last, t = None, None, 0
    for n, nsets in ((7, 3), (9, 1)):
        sets = env.output_sets(nsets) if nsets > 1 else None
        r = env.rollout_uniform_sets(5, t, n, actions, sets=sets, resume=t > 0, eps=0.05)
        rs, ra = oracle.rollout_uniform_sets(ref, 5, t, n, nsets, resume=t > 0, actions=ra,
                                             first_mask=r_last["mask"] if r_last else None, eps=0.05)
        r_last = rs[(n - 1) % nsets]
        t += n
        np.testing.assert_array_equal(actions.cpu().numpy(), ra)
        for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), r_last[k], err_msg=k)
    np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
    np.testing.assert_array_equal(env.ep_acc.cpu().numpy(), ref.ep_acc)


def test_gpu_masks_property_random_geometries(bpp, oracle):
    """Random small geometries / heightmaps / items (incl. items larger than the bin, heights above H):
    both mask kernels' entry points == the oracle for both rules and both rotation settings."""
    rng = np.random.RandomState(99)
    for trial in range(60):
        W, L, H = rng.randint(1, 14), rng.randint(1, 14), rng.randint(1, 15)
        n = rng.randint(1, 40)
        hm = rng.randint(0, H + 2, size=(n, W * L)).astype(np.int32)          # some cells above H
        hm[rng.rand(n) < 0.4] = rng.randint(0, H + 1)                         # some flat maps
        items = np.stack([rng.randint(1, W + 2, n), rng.randint(1, L + 2, n), rng.randint(1, H + 1, n)], 1).astype(np.int32)
        size = (W, L, H)
        A = W * L
        obs = np.concatenate([hm, np.repeat(items[:, 0:1], A, 1), np.repeat(items[:, 1:2], A, 1),
                              np.repeat(items[:, 2:3], A, 1)], 1).astype(np.float32)
        for rot in (False, True):
            for rule, rname in ((0, "utils"), (1, "space")):
                want = oracle.mask_from_hmap(hm, items, size, rot, rule)
                np.testing.assert_array_equal(bpp.batched_mask_from_hmap(hm, items, size, rot, rname).cpu().numpy(), want,
                                              err_msg="hmap %r rot=%d rule=%d" % (size, rot, rule))
                np.testing.assert_array_equal(bpp.batched_mask_from_obs(obs, size, rot, rname).cpu().numpy(), want,
                                              err_msg="obs %r rot=%d rule=%d" % (size, rot, rule))


def test_gpu_config0_shape_rs_16_envs_through_the_factory(bpp, oracle):
    """BASELINE.json configs[0] shape: 10x10x10, --item-seq rs, 16 envs, driven through the reference-shaped
    factory and step() with a loop written like main.py:148-174 (per-row mask helper, infos scan); every
    value equals the oracle fed the factory's own pool."""
    import types
    import torch
    box_set = [(i, j, k) for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)]   # arguments.py:122-128
    args = types.SimpleNamespace(container_size=(10, 10, 10), enable_rotation=False, data_type="rs", box_size_set=box_set)
    envs = bpp.make_vec_envs("Bpp-v0", 1, 16, 1.0, None, "cuda:0", False, args=args, pool_size=64)
    ref = oracle.OracleEnv(envs.pool_host, (10, 10, 10), False, 16)
    obs = envs.reset()
    robs, rmask = ref.reset()
    rng = np.random.RandomState(0)
    episode_rewards = []
    for t in range(80):
        location_masks = [bpp.get_possible_position(o, args.container_size) for o in obs]     # main.py:163-169
        np.testing.assert_array_equal(np.array(location_masks, np.float32), rmask)
        action = torch.tensor([[int(rng.choice(np.flatnonzero(m)))] for m in location_masks])
        obs, reward, done, infos = envs.step(action)
        o = ref.step(action.numpy()[:, 0])
        for i in range(len(infos)):                                                            # main.py:159-162
            if "episode" in infos[i].keys():
                episode_rewards.append(infos[i]["episode"]["r"])
                assert infos[i]["episode"]["r"] == round(float(o["ep_ret"][i]), 6)
        np.testing.assert_array_equal(obs.cpu().numpy(), o["obs"])
        np.testing.assert_array_equal(reward.numpy()[:, 0], o["reward"])
        np.testing.assert_array_equal(done, o["done"].astype(bool))
        rmask = o["mask"]
    assert len(episode_rewards) > 20


@pytest.mark.parametrize("rot,fresh", [(False, True), (True, False)])
def test_gpu_row_helpers_hand_back_the_envs_own_mask_rows(bpp, oracle, rot, fresh):
    """masks.ROW_CACHE: a row of the observation the env has just returned gets the mask row the step kernel wrote for it (one
    fetch of the [E, M] mask per lock-step); the same rows through the kernel path (ROW_CACHE off), rows of an OLDER step, a
    CPU copy and a row the caller built all give the oracle's mask of that observation."""
    import torch
    from bpp_amd import masks
    size, E = (10, 10, 10), 24
    pool = bpp.sequences.cut2_pool(size, 64, seed=3)
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool, fresh_outputs=fresh)
    helper = bpp.get_rotation_mask if rot else (lambda o, s: np.array(bpp.get_possible_position(o, s), np.int32))
    obs = env.reset()
    older = None
    for t in range(12):
        want = oracle.mask_from_obs(obs.cpu().numpy(), size, rot).astype(np.int32)
        fetched_before = getattr(env, "_mask_rows_host", (None,))[0]
        rows = np.stack([helper(o, size) for o in obs])
        np.testing.assert_array_equal(rows, want)
        assert env._mask_rows_host[0] != fetched_before and env._mask_rows_host[0][0] == env._serial      # the cache answered, once
        masks.ROW_CACHE = False
        try:
            np.testing.assert_array_equal(np.stack([helper(o, size) for o in obs[:5]]), want[:5])          # the kernel path
        finally:
            masks.ROW_CACHE = True
        np.testing.assert_array_equal(helper(obs[3].cpu(), size), want[3])                               # a CPU row
        np.testing.assert_array_equal(helper(obs[4].clone(), size), want[4])                             # a row of the caller's own
        if older is not None and fresh:       # a row of the previous step's observation (its buffer is still ours: fresh outputs)
            np.testing.assert_array_equal(helper(older[0][2], size), older[1][2])
        older = (obs, want)
        a = env.sample_feasible(seed=5, step=t)
        obs, _, _, _ = env.step(a)


def test_gpu_example_policy_in_the_loop_runs():
    """examples/rollout_with_policy.py: CNN policy -> bpp_masked_act -> step_tensors -> EpisodeStats, end to end."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (["--rotation"], ["--stream"]):
        out = subprocess.run([sys.executable, os.path.join(root, "examples", "rollout_with_policy.py"), "--envs", "512",
                              "--steps", "12"] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "policy in the loop" in out.stdout and "episodes" in out.stdout


@pytest.mark.parametrize("epw,wpb", [(1, 1), (1, 16), (2, 4), (8, 2), (8, 8), (16, 4), (4, 16), (64, 1)])
def test_gpu_tuning_knobs_do_not_change_results(bpp, knobs, epw, wpb):
    """Every bins-per-wave / waves-per-workgroup setting (incl. workgroups above 64 KiB of LDS and a wave
    serving 64 bins) replays the reference's golden rollouts bit for bit, with and without XCD remapping."""
    knobs(bins_per_wave=epw, waves_per_group=wpb, xcd_remap=(epw + wpb) & 1)
    for case in ("rollout_cut2_10_rot", "rollout_cut2_20", "rollout_wide_8x12x9_rot"):
        g = load_golden(case)
        size = tuple(int(v) for v in g["size"])
        if epw * (size[0] * size[1] * (2 + int(g["rotation"])) + 48 + (size[0] + 1) * (size[1] + 1) * 16) * wpb > 150 * 1024:
            continue      # would not fit the 160 KiB of LDS per workgroup: the library refuses, nothing to compare
        check_rollout(lambda pool, sz, rot, E, rule: GpuEnv(bpp, pool, sz, rot, E, rule), g)


def test_gpu_env_checkpoint_resume(bpp):
    """state_dict()/load_state_dict(): the environment state is three tensors; resuming replays identically."""
    import torch
    size, E = (10, 10, 10), 700
    pool = bpp.sequences.cut2_pool(size, 32, seed=4)
    env = bpp.BppVecEnv(E, size, enable_rotation=True, pool=pool)
    env.reset()
    env.rollout_uniform(seed=3, step0=0, nsteps=9)
    ckpt = env.state_dict()
    first = env.rollout_uniform(seed=3, step0=9, nsteps=11)
    want = {k: getattr(first, k).clone() for k in ("obs", "mask", "counter", "ratio", "ep_ret")}
    other = bpp.BppVecEnv(E, size, enable_rotation=True, pool=pool)       # a fresh env object, never reset
    other.load_state_dict(ckpt)
    assert torch.equal(other.location_masks, ckpt["mask"])
    again = other.rollout_uniform(seed=3, step0=9, nsteps=11)
    for k, v in want.items():
        assert torch.equal(getattr(again, k), v), k


def test_gpu_mask_entry_points_clamp_oversized_items(bpp, oracle):
    """Items far wider than the bin (also beyond a byte) cannot fit anywhere: all-ones fallback, exactly like
    the reference's empty loop ranges (acktr/utils.py:54-60)."""
    size = (10, 10, 10)
    hm = np.zeros((3, 100), np.int32)
    items = np.array([[300, 2, 2], [2, 1000, 2], [3, 3, 3]], np.int32)
    want = oracle.mask_from_hmap(hm, items, size, True, 0)
    assert want[0].min() == 1 and want[1].min() == 1 and want[2].sum() == 128
    np.testing.assert_array_equal(bpp.batched_mask_from_hmap(hm, items, size, True, "utils").cpu().numpy(), want)


def _lockstep_all_bins(bpp, oracle, size, rot, E, base, total, P, steps=12, seed=17):
    """HIP vs oracle on EVERY bin of a full-size shard: observation, mask, reward, done, counter, ratio,
    Monitor sums every step; byte heightmaps and complete state records at the end.  Step `steps // 2` forces
    failures on a third of the bins (corner placement), so auto-reset and pool-row advance happen mid-grid."""
    pool = bpp.sequences.cut2_pool(size, P, seed=1)
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool, env_id_base=base, env_id_total=total)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=base, env_id_total=total)
    obs = env.reset()
    robs, rmask = ref.reset()
    np.testing.assert_array_equal(obs.cpu().numpy(), robs)
    np.testing.assert_array_equal(env.location_masks.cpu().numpy(), rmask)
    A = size[0] * size[1]
    finished = 0
    for t in range(steps):
        a = env.sample_feasible(seed=seed, step=t)
        if t == steps // 2:
            a[::3] = A - 1
        r = env.step_tensors(a)
        o = ref.step(a.cpu().numpy(), copy=False)
        for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(getattr(r, k).cpu().numpy(), o[k], err_msg="%s t=%d" % (k, t))
        np.testing.assert_array_equal(r.reward.cpu().numpy()[:, 0], o["reward"], err_msg="reward t=%d" % t)
        finished += int(o["done"].sum())
    np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
    st = env.state_numpy()
    for f in st.dtype.names:
        if f != "pad":
            np.testing.assert_array_equal(st[f], ref.state[f], err_msg=f)
    np.testing.assert_array_equal(env.ep_acc.cpu().numpy(), ref.ep_acc)
    got, want = env.episode_stats().cpu().numpy(), ref.episode_stats()
    np.testing.assert_array_equal(env.episode_stats(wide=False).cpu().numpy(), want)   # one-workgroup form: the same bits
    np.testing.assert_array_equal(got, want)
    assert finished > E // 4


@pytest.mark.parametrize("path,xcd", [("tile", 1), ("tile", 0), ("rt", 1), ("generic", 1)])
@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), False, 65536), ((10, 10, 10), True, 65536),
                                         ((20, 20, 20), False, 32768)])
def test_gpu_full_size_every_bin_matches_oracle(bpp, oracle, knobs, size, rot, E, path, xcd):
    """BASELINE.json configs 2-4 at full size, ALL bins compared (not slices): the XCD block remap, tail
    workgroups and middle-of-grid bins are covered on both kernel paths, remap on and off."""
    knobs(force_generic=int(path == "generic"), legacy_fast=int(path == "rt"), xcd_remap=xcd, bins_per_wave=0, waves_per_group=0)
    assert bpp._lib.launch_info(E, size, rot)["kernel"] == {"tile": 2, "rt": 1, "generic": 0}[path]
    _lockstep_all_bins(bpp, oracle, size, rot, E, 0, E, 64 if size[0] > 10 else 512)


@pytest.mark.parametrize("base,total,P", [(458752, 524288, 8192),      # BASELINE config 5, rank 7 of 8
                                           (458752, 524288, 8191),      # prime pool: base_mod and seq_stride both wrap
                                           (65536 * 3 + 5, 65536 * 4 + 77, 1000),
                                           (2 ** 31 - 70000, 2 ** 31 + 12345, 4099)])   # ids beyond int32
def test_gpu_shard_coordinates_of_multi_gpu_jobs(bpp, oracle, kernel_path, base, total, P):
    """A rank of a multi-GPU job is a shard with env_id_base > 0: every bin of a full 65 536-bin shard at the
    coordinates of BASELINE config 5's last rank (and at bases where base mod P / total mod P wrap) equals
    the oracle stepping the same global ids."""
    _lockstep_all_bins(bpp, oracle, (10, 10, 10), False, 65536, base, total, P, steps=10)
