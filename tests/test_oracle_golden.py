"""Pins the oracle: oracle/bpp_oracle.c vs golden vectors recorded from the unmodified reference
(tests/golden/make_golden.py).  Bit-exact for everything, float values included (rewards are
float32(float64 arithmetic), ratios/returns are float64 sums in step order)."""
import numpy as np
import pytest

from conftest import DEEP_CASES, MASK_CASES, ROLLOUT_CASES, depth_profile, load_golden


def check_rollout(env_factory, g):
    """Shared by the oracle test and the GPU parity test: replay golden actions, compare everything."""
    size = tuple(int(v) for v in g["size"])
    rot = int(g["rotation"])
    E = g["actions"].shape[1]
    for rule, m0, mt in ((0, g["mask0"], g["mask"]), (1, g["smask0"], g["smask"])):
        if rule == 1 and rot:
            continue  # PackingGame.get_possible_position has no rotation variant
        env = env_factory(g["pool"], size, rot, E, rule)
        obs, mask = env.reset()
        assert obs.dtype == np.float32 and mask.dtype == np.float32
        np.testing.assert_array_equal(obs, g["obs0"].astype(np.float32))
        np.testing.assert_array_equal(mask, m0.astype(np.float32))
        for t in range(g["actions"].shape[0]):
            o = env.step(g["actions"][t])
            np.testing.assert_array_equal(o["obs"], g["obs"][t].astype(np.float32), err_msg="obs t=%d" % t)
            np.testing.assert_array_equal(o["mask"], mt[t].astype(np.float32), err_msg="mask t=%d" % t)
            np.testing.assert_array_equal(o["reward"], g["reward"][t], err_msg="reward t=%d" % t)
            np.testing.assert_array_equal(o["done"], g["done"][t], err_msg="done t=%d" % t)
            np.testing.assert_array_equal(o["counter"], g["counter"][t], err_msg="counter t=%d" % t)
            np.testing.assert_array_equal(o["ratio"], g["ratio"][t], err_msg="ratio t=%d" % t)
            d = g["done"][t].astype(bool)
            np.testing.assert_array_equal(o["ep_ret"][d], g["ep_r_raw"][t][d], err_msg="ep_ret t=%d" % t)
            np.testing.assert_array_equal(np.round(o["ep_ret"][d], 6), g["ep_r"][t][d])
            np.testing.assert_array_equal(o["ep_len"][d], g["ep_l"][t][d], err_msg="ep_len t=%d" % t)


def check_masks(mask_from_obs, mask_from_hmap, g):
    size = tuple(int(v) for v in g["size"])
    A = size[0] * size[1]
    hm, it = g["hmap"], g["items"]
    obs = np.concatenate([hm, np.repeat(it[:, 0:1], A, 1), np.repeat(it[:, 1:2], A, 1), np.repeat(it[:, 2:3], A, 1)],
                         axis=1).astype(np.float32)
    np.testing.assert_array_equal(mask_from_obs(obs, size, 0, 0), g["mask_utils"].astype(np.float32))
    np.testing.assert_array_equal(mask_from_obs(obs, size, 1, 0), g["mask_utils_rot"].astype(np.float32))
    np.testing.assert_array_equal(mask_from_obs(obs, size, 0, 1), g["mask_space"].astype(np.float32))
    np.testing.assert_array_equal(mask_from_hmap(hm, it, size, 0, 1), g["mask_space"].astype(np.float32))
    np.testing.assert_array_equal(mask_from_hmap(hm, it, size, 1, 0), g["mask_utils_rot"].astype(np.float32))


@pytest.mark.parametrize("case", ROLLOUT_CASES)
def test_oracle_rollout_matches_reference_golden(oracle, case):
    check_rollout(lambda pool, size, rot, E, rule: oracle.OracleEnv(pool, size, rot, E, mask_rule=rule),
                  load_golden(case))


@pytest.mark.parametrize("case", DEEP_CASES)
def test_deep_fixtures_hold_deep_states(case):
    """the round-6 fixtures must contain what the uniform-feasible recordings do not -- bins with >= 20
    boxes on >= 5 % of the recorded env-steps (10x10x10 heuristic cases; 20x20x20: most of them), completely packed bins,
    and, for the reference's own checkpoints, the utilisation the paper reports for them (~0.7 on CUT-2)."""
    g = load_golden(case)
    deep, full, ratio = depth_profile(g)
    if case.startswith("rollout_deep"):
        assert deep >= (0.5 if case.endswith("_20") else 0.05), deep
        assert full >= (0 if case.endswith("_20") else 1), full
        assert ratio > 0.55
    else:
        assert ratio > 0.7 and deep >= 0.04 and g["counter"].max() >= 24


@pytest.mark.parametrize("case", MASK_CASES)
def test_oracle_masks_match_reference_golden(oracle, case):
    check_masks(oracle.mask_from_obs, oracle.mask_from_hmap, load_golden(case))


def test_oracle_kat1(oracle):
    """SURVEY.md Appendix B KAT-1 (captured from the reference)."""
    pool = np.zeros((1, 5, 4), np.uint8)
    pool[0, :, :3] = [(4, 5, 2), (3, 3, 3), (2, 2, 5), (5, 5, 5), (10, 10, 10)]
    env = oracle.OracleEnv(pool, (10, 10, 10), 0, 1)
    obs, mask = env.reset()
    sums, rews, ratios = [int(mask.sum())], [], []
    for a in (0, 0, 40, 55):
        o = env.step([a])
        sums.append(int(o["mask"].sum()))
        rews.append(float(o["reward"][0]))
        ratios.append(float(o["ratio"][0]))
        assert not o["done"][0]
    assert sums[:4] == [42, 50, 68, 12]
    assert rews == [np.float32(0.4), np.float32(0.27), np.float32(0.2), np.float32(1.25)]
    assert ratios == [0.04, 0.067, 0.087, 0.212]
    rows = env.hmap.reshape(10, 10)
    assert rows[0].tolist() == [5, 5, 5, 2, 2, 0, 0, 0, 0, 0] and rows[5].tolist() == [5, 5, 0, 0, 0, 5, 5, 5, 5, 5]


def test_oracle_kat2_rotation_quirk(oracle):
    """KAT-2: mask[A]=1 but step(A) ends the episode (idx > area is strict, bin3D.py:102)."""
    pool = np.zeros((1, 3, 4), np.uint8)
    pool[0, :, :3] = [(2, 3, 2), (2, 3, 2), (10, 10, 10)]
    env = oracle.OracleEnv(pool, (10, 10, 10), 1, 1)
    obs, mask = env.reset()
    assert int(mask.sum()) == 144 and mask[0, 100] == 1.0
    o = env.step([100])
    assert o["done"][0] == 1 and o["reward"][0] == 0.0 and o["counter"][0] == 0
    o = env.step([101])
    assert o["done"][0] == 0 and o["reward"][0] == np.float32(0.12)
    assert (env.hmap.reshape(10, 10)[0:3, 1:3] == 2).all() and env.hmap.sum() == 12
