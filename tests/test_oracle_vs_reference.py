"""Build-container only: fresh seeded rollouts of the LIVE reference (unmodified PackingGame + Monitor +
DummyVecEnv + VecPyTorch + per-row acktr.utils masks) beside the C oracle.  Complements the committed
golden vectors with cases generated at test time."""
import numpy as np
import pytest

from oracle import ref_shims

pytestmark = pytest.mark.skipif(not ref_shims.available(), reason="reference tree not present")


@pytest.mark.parametrize("size,rot,seed", [((10, 10, 10), False, 1), ((10, 10, 10), True, 2), ((6, 9, 7), True, 3),
                                            ((12, 12, 12), False, 4)])
def test_oracle_tracks_live_reference(oracle, size, rot, seed):
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        import make_golden as mg
    import torch
    rng = np.random.RandomState(seed)
    hi = max(2, min(size) // 2)
    seqs = [[tuple(rng.randint(1, hi + 1, size=3)) for _ in range(rng.randint(5, 40))] for _ in range(9)]
    pool = mg.pad_pool(seqs, max(len(s) for s in seqs) + 1, size)
    E = 4
    dummy, venv = mg.make_stack(pool, size, rot, E)
    env = oracle.OracleEnv(pool, size, rot, E)
    obs = venv.reset()
    o_obs, o_mask = env.reset()
    np.testing.assert_array_equal(obs.numpy(), o_obs)
    mask = mg.loop_masks(obs, size, rot)
    np.testing.assert_array_equal(mask, o_mask)
    M = mask.shape[1]
    for t in range(60):
        a = np.array([rng.choice(np.flatnonzero(mask[e])) if rng.rand() > 0.1 else rng.randint(0, M) for e in range(E)])
        obs, rew, done, infos = venv.step(torch.from_numpy(a).unsqueeze(1))
        mask = mg.loop_masks(obs, size, rot)
        o = env.step(a)
        np.testing.assert_array_equal(obs.numpy(), o["obs"])
        np.testing.assert_array_equal(mask, o["mask"])
        np.testing.assert_array_equal(rew.numpy()[:, 0], o["reward"])
        np.testing.assert_array_equal(np.asarray(done), o["done"].astype(bool))
        for e, i in enumerate(infos):
            assert i["counter"] == o["counter"][e] and float(i["ratio"]) == o["ratio"][e]
            if done[e]:
                assert i["episode"]["l"] == o["ep_len"][e] and i["episode"]["r"] == round(float(o["ep_ret"][e]), 6)
