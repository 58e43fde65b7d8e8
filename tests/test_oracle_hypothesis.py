"""Property-based differential test (build container only): for arbitrary small bins, heightmaps and items
the oracle's masks equal the live reference's -- acktr.utils.get_possible_position / get_rotation_mask
(rule U) and PackingGame.get_possible_position (rule S) -- and one reference step equals one oracle step."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import ref_shims

pytestmark = pytest.mark.skipif(not ref_shims.available(), reason="reference tree not present")


@st.composite
def scenes(draw):
    W, L, H = draw(st.integers(2, 9)), draw(st.integers(2, 9)), draw(st.integers(2, 12))
    kind = draw(st.integers(0, 2))
    if kind == 0:
        cells = draw(st.lists(st.integers(0, H), min_size=W * L, max_size=W * L))
    elif kind == 1:   # two plateaus: feasible positions likely
        a, b, cut = draw(st.integers(0, H)), draw(st.integers(0, H)), draw(st.integers(0, W))
        cells = [a if i < cut else b for i in range(W) for _ in range(L)]
    else:             # flat with a few dents
        base = draw(st.integers(0, H))
        cells = [base] * (W * L)
        for _ in range(draw(st.integers(0, 3))):
            cells[draw(st.integers(0, W * L - 1))] = draw(st.integers(0, H))
    item = (draw(st.integers(1, W + 1)), draw(st.integers(1, L + 1)), draw(st.integers(1, H)))
    return (W, L, H), np.array(cells, np.int32), item


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenes())
def test_oracle_masks_equal_live_reference(oracle, scene):
    import torch
    ref_shims.install()
    from acktr.utils import get_possible_position, get_rotation_mask
    from envs.bpp0 import PackingGame
    size, hm, item = scene
    W, L, H = size
    A = W * L
    obs = np.concatenate([hm, np.full(A, item[0]), np.full(A, item[1]), np.full(A, item[2])]).astype(np.float32)
    assert oracle.mask_from_obs(obs, size, 0, 0)[0].tolist() == [float(v) for v in get_possible_position(obs, size)]
    np.testing.assert_array_equal(oracle.mask_from_obs(obs, size, 1, 0)[0], get_rotation_mask(torch.from_numpy(obs), size))
    env = PackingGame(box_creator=ref_shims.make_replay_creator([[item]], size), container_size=size)
    env.reset()
    np.testing.assert_array_equal(oracle.mask_from_hmap(hm, np.array(item, np.int32), size, 0, 1)[0],
                                  env.get_possible_position(plain=hm.reshape(W, L)).reshape(-1))


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenes(), st.integers(-2, 200), st.booleans())
def test_oracle_single_step_equals_live_reference(oracle, scene, action, rot):
    """Arbitrary (also unreachable) heightmap, arbitrary action: reward, done, new map, info."""
    ref_shims.install()
    from envs.bpp0 import PackingGame
    size, hm, item = scene
    W, L, H = size
    A = W * L
    action = min(action, A * (2 if rot else 1) + 1) if rot else min(action, A)
    pool = np.zeros((1, 3, 4), np.uint8)
    pool[0, :, :3] = [item, (1, 1, 1), size]
    env = PackingGame(box_creator=ref_shims.make_replay_creator([[item, (1, 1, 1)]], size), container_size=size,
                      enable_rotation=rot)
    env.reset()
    env.space.plain = hm.reshape(W, L).copy()
    o = oracle.OracleEnv(pool, size, rot, 1)
    o.reset()
    o.hmap[0] = hm.astype(np.uint8)
    obs, rew, done, info = env.step(action)
    r = o.step([action])
    assert bool(r["done"][0]) == bool(done) and r["reward"][0] == np.float32(rew)
    assert r["counter"][0] == info["counter"] and r["ratio"][0] == float(info["ratio"])
    if not done:
        np.testing.assert_array_equal(o.hmap[0], env.space.plain.reshape(-1))
        np.testing.assert_array_equal(r["obs"][0], obs.astype(np.float32))
