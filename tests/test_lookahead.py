"""what the reference's lookahead consumers do with the env, batched.  Goldens recorded from the
unmodified reference (tests/golden/make_lookahead_golden.py):
  * lookahead_branch_10: play a prefix, copy.deepcopy(env) four times, step every copy with another action, two levels
    deep (acktr/reorder.py:245-262, MCTS/node.py:92-137) -- here: copy_bins + step_subset (BPP_ACTION_NOOP for the rest);
  * windows_20_to_10: 10x10 sliding windows over 20x20 pallets with get_possible_position per window
    (multi_bin/multi_bin.py:7-17,28-45) -- here: batched_window_masks.
The oracle and the emulated product kernels replay them on the CPU, the HIP path in the `-m gpu` tests."""
import numpy as np
import pytest

from conftest import load_golden

NOOP = -2 ** 63


def replay_branches(make_env, g):
    """make_env(pool, E) -> object with reset(), step(actions ndarray int64 [E]) -> dict of numpy, copy(src, dst)."""
    size = tuple(int(v) for v in g["size"])
    A = size[0] * size[1]
    n, B = int(g["n"]), 4
    E = n * (1 + B)                                         # per root: the root bin + B copies
    pool = g["pool"]
    # global bin e plays pool row e mod P: give root r the row r by placing the roots at ids 0..n-1 of a P = n pool
    env = make_env(pool, E)
    obs, mask = env.reset()
    maxlen = max(len(g["%d_prefix" % r]) for r in range(n))
    for t in range(maxlen):                                 # prefixes: roots step, everything else is left alone
        a = np.full(E, NOOP, np.int64)
        for r in range(n):
            pre = g["%d_prefix" % r]
            if t < len(pre):
                a[r] = pre[t]
        o = env.step(a)
        assert not o["done"].any() and (o["reward"][n:] == 0).all()
    for r in range(n):
        np.testing.assert_array_equal(o["obs"][r], g["%d_obs0" % r].astype(np.float32))
        np.testing.assert_array_equal(o["mask"][r], g["%d_mask0" % r].astype(np.float32))
    # branch: copies of root r live in bins n + r*B .. n + r*B + B-1   (copy.deepcopy(env), acktr/reorder.py:247)
    src = np.repeat(np.arange(n), B)
    dst = n + np.arange(n * B)
    o = env.copy(src, dst)
    for r in range(n):
        for b in range(B):                                  # the refreshed copies show the root's observation and mask
            np.testing.assert_array_equal(o["obs"][n + r * B + b], g["%d_obs0" % r].astype(np.float32))
            np.testing.assert_array_equal(o["mask"][n + r * B + b], g["%d_mask0" % r].astype(np.float32))
    a = np.full(E, NOOP, np.int64)
    for r in range(n):
        a[n + r * B:n + (r + 1) * B] = g["%d_level1" % r]
    o1 = env.step(a)
    for r in range(n):
        sl = slice(n + r * B, n + (r + 1) * B)
        done = g["%d_done1" % r]
        np.testing.assert_array_equal(o1["done"][sl].astype(bool), done)
        np.testing.assert_array_equal(o1["reward"][sl], g["%d_rew1" % r].astype(np.float32))
        np.testing.assert_array_equal(o1["counter"][sl], g["%d_counter1" % r])
        np.testing.assert_array_equal(o1["ratio"][sl], g["%d_ratio1" % r])
        for b in range(B):
            if not done[b]:                                 # (a finished copy shows its auto-reset observation instead)
                np.testing.assert_array_equal(o1["obs"][n + r * B + b], g["%d_obs1" % r][b].astype(np.float32))
                np.testing.assert_array_equal(o1["mask"][n + r * B + b], g["%d_mask1" % r][b].astype(np.float32))
        # the roots were not touched by the branch step
        np.testing.assert_array_equal(o1["obs"][r], g["%d_obs0" % r].astype(np.float32))
        assert o1["reward"][r] == 0 and o1["done"][r] == 0
    a = np.full(E, NOOP, np.int64)
    for r in range(n):
        for b in range(B):
            if not g["%d_done1" % r][b]:
                a[n + r * B + b] = g["%d_level2" % r][b]
    o2 = env.step(a)
    for r in range(n):
        for b in range(B):
            if not g["%d_done1" % r][b]:
                e = n + r * B + b
                assert bool(o2["done"][e]) == bool(g["%d_done2" % r][b])
                assert o2["reward"][e] == np.float32(g["%d_rew2" % r][b])
                if not g["%d_done2" % r][b]:
                    np.testing.assert_array_equal(o2["obs"][e], g["%d_obs2" % r][b].astype(np.float32))


class HostEnv(object):
    """numpy front-end (oracle library, or the emulated product) with the copy operation done on its arrays."""

    def __init__(self, mod, pool, E, size):
        self.env = mod.OracleEnv(pool, size, True, E)

    def reset(self):
        return self.env.reset()

    def step(self, a):
        return self.env.step(a)

    def copy(self, src, dst):
        self.env.hmap[dst] = self.env.hmap[src]
        self.env.state[dst] = self.env.state[src]
        return self.env.step(np.full(self.env.E, NOOP, np.int64))


def test_oracle_lookahead_branches_match_reference_deepcopy(oracle):
    g = load_golden("lookahead_branch_10")
    replay_branches(lambda pool, E: HostEnv(oracle, pool, E, tuple(int(v) for v in g["size"])), g)


@pytest.mark.parametrize("path", ["tile", "rt", "generic"])
def test_emulated_lookahead_branches_match_reference_deepcopy(emu, path):
    g = load_golden("lookahead_branch_10")
    emu.set_knobs(force_generic=int(path == "generic"), legacy_fast=int(path == "rt"))
    try:
        replay_branches(lambda pool, E: HostEnv(emu, pool, E, tuple(int(v) for v in g["size"])), g)
    finally:
        emu.set_knobs()


def window_obs(g):
    hm, items, offs = g["hmap"], g["items"], g["offsets"]
    n = hm.shape[0]
    wins = np.stack([hm[:, dx:dx + 10, dy:dy + 10].reshape(n, 100) for dx, dy in offs], 1)      # [n, 4, 100]
    return wins, np.repeat(items[:, None, :], len(offs), 1)


def test_oracle_and_emulated_window_masks_match_reference(oracle, emu):
    g = load_golden("windows_20_to_10")
    wins, its = window_obs(g)
    want = g["masks"].astype(np.float32).reshape(-1, 100)
    for mod in (oracle, emu):
        np.testing.assert_array_equal(mod.mask_from_hmap(wins.reshape(-1, 100), its.reshape(-1, 3), (10, 10, 10), False, 0), want)


def test_noop_actions_leave_bins_untouched(oracle, emu):
    """BPP_ACTION_NOOP on a random subset, mixed with real and failing actions, product (emulated) == oracle; the
    untouched bins keep heightmap and every state field."""
    from bpp_amd import sequences
    size, E = (10, 10, 10), 61
    pool = sequences.cut2_pool(size, 9, seed=4, native=False)
    rng = np.random.RandomState(1)
    for rot in (False, True):
        a_env = emu.OracleEnv(pool, size, rot, E)
        b_env = oracle.OracleEnv(pool, size, rot, E)
        a_env.reset()
        _, mask = b_env.reset()
        for t in range(30):
            a = oracle.sample_feasible(mask, 3, t)
            a[rng.rand(E) < 0.1] = -5
            skip = rng.rand(E) < 0.4
            a[skip] = NOOP
            before = (b_env.hmap.copy(), b_env.state.copy())
            ra, rb = a_env.step(a), b_env.step(a)
            for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(ra[k], rb[k], err_msg="%s t=%d" % (k, t))
            np.testing.assert_array_equal(a_env.hmap, b_env.hmap)
            np.testing.assert_array_equal(a_env.state, b_env.state)
            np.testing.assert_array_equal(b_env.hmap[skip], before[0][skip])
            np.testing.assert_array_equal(b_env.state[skip], before[1][skip])
            assert (rb["reward"][skip] == 0).all() and (rb["done"][skip] == 0).all()
            mask = rb["mask"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["tile", "rt", "generic"])
def test_gpu_lookahead_branches_match_reference_deepcopy(path):
    import torch
    import bpp_amd
    g = load_golden("lookahead_branch_10")
    size = tuple(int(v) for v in g["size"])

    class GpuEnv(object):
        def __init__(self, pool, E):
            self.env = bpp_amd.BppVecEnv(E, size, enable_rotation=True, pool=pool)

        def _out(self, r):
            out = {k: getattr(r, k).cpu().numpy() for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len")}
            out["reward"] = r.reward.cpu().numpy()[:, 0]
            return out

        def reset(self):
            obs = self.env.reset()
            return obs.cpu().numpy(), self.env.location_masks.cpu().numpy()

        def step(self, a):
            ids = np.flatnonzero(a != NOOP)
            return self._out(self.env.step_subset(ids, a[ids]))          # the API the searches use

        def copy(self, src, dst):
            return self._out(self.env.clone_into(src, dst))

    old = bpp_amd._lib.set_knobs(force_generic=int(path == "generic"), legacy_fast=int(path == "rt"))
    try:
        replay_branches(GpuEnv, g)
    finally:
        bpp_amd._lib.set_knobs(**old)


@pytest.mark.gpu
def test_gpu_window_masks_and_item_override():
    import torch
    import bpp_amd
    g = load_golden("windows_20_to_10")
    masks, offs = bpp_amd.batched_window_masks(torch.from_numpy(g["hmap"]), g["items"], (10, 10, 10), stride=10)
    np.testing.assert_array_equal(offs.numpy(), g["offsets"])
    np.testing.assert_array_equal(masks.cpu().numpy(), g["masks"].astype(np.float32))
    # set_current_items + observe: the mask of an edited bin equals the stand-alone mask kernel on (heightmap, new item)
    size, E = (10, 10, 10), 40
    env = bpp_amd.BppVecEnv(E, size, enable_rotation=True, pool=bpp_amd.sequences.cut2_pool(size, 8, seed=2))
    env.reset()
    env.rollout_uniform(seed=1, step0=0, nsteps=7)
    items = torch.tensor([[2, 3, 4], [5, 1, 2], [3, 3, 3]])
    env.set_current_items([4, 9, 30], items)
    r = env.observe()
    want = bpp_amd.batched_mask_from_hmap(env.heightmaps()[[4, 9, 30]].reshape(3, -1), items, size, True, "utils")
    assert torch.equal(r.mask[[4, 9, 30]], want)
    assert torch.equal(r.obs[[4, 9, 30]].view(3, 4, 100)[:, 1:, 0], items.float().to(r.obs.device))


@pytest.mark.parametrize("rot", [False, True])
def test_emulated_reordered_items_match_oracle(emu, oracle, rot):
    """set_current_items (the reorder search plays previewed items in another order, acktr/reorder.py:181-215)
    overwrites bpp_env_state.item_cur; both the kernels and -- now -- the oracle play THAT item, then continue the
    sequence as before.  Subset stepping (BPP_ACTION_NOOP) for the bins that are not being reordered."""
    from bpp_amd import sequences
    size, E = (10, 10, 10), 37
    pool = sequences.cut2_pool(size, 16, seed=9, native=False)
    envs = [m.OracleEnv(pool, size, rot, E, mask_rule=1) for m in (emu, oracle)]
    masks = [e.reset()[1] for e in envs]
    rng = np.random.RandomState(4)
    T = pool.shape[1]
    for t in range(14):
        ids = np.flatnonzero(rng.rand(E) < 0.4)
        for env in envs:                                 # swap in the item two places ahead (preview(3)[2]) for bins `ids`
            st = env.state
            for e in ids:
                it = pool[int(st["seq"][e]), min(int(st["cursor"][e]) + 2, T - 1)]
                st["item_cur"][e] = int(it[0]) | (int(it[1]) << 8) | (int(it[2]) << 16)
        outs = [env.step(np.full(E, NOOP, np.int64)) for env in envs]            # observe(): new observation / mask
        for k in ("obs", "mask"):
            np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg="%s observe t=%d" % (k, t))
        for e in ids:                                    # the overwritten item is what the observation shows
            st = envs[1].state
            assert int(outs[1]["obs"][e, 100]) == int(st["item_cur"][e]) & 255
        a = oracle.sample_feasible(outs[1]["mask"], 3, t)
        a[rng.rand(E) < 0.3] = NOOP                      # step a subset only
        outs = [env.step(a) for env in envs]
        for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg="%s t=%d" % (k, t))
        for f in ("cursor", "episode", "n_boxes", "vol_sum", "seq", "item_cur", "item_next", "item_reset"):
            np.testing.assert_array_equal(envs[0].state[f], envs[1].state[f], err_msg=f)
    np.testing.assert_array_equal(envs[0].hmap, envs[1].hmap)
    assert envs[1].state["n_boxes"].max() >= 3
