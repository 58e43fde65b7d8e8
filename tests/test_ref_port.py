"""oracle/ref_port.py (the pure-Python restatement bench.py times on the GPU box as `cpu_baseline.python_port`) gives the
same masks, rewards, episode ends, counters, ratios and heightmaps as the C oracle, which is pinned to the live reference
(tests/test_oracle_golden.py, tests/test_oracle_vs_reference.py)."""
import numpy as np
import pytest

from bpp_amd import sequences
from oracle import ref_port


@pytest.mark.parametrize("size,rot", [((10, 10, 10), False), ((10, 10, 10), True), ((8, 12, 9), True)])
def test_python_port_equals_oracle(oracle, size, rot):
    E, steps = 6, 70
    pool = sequences.cut2_pool(size, 16, seed=3)
    rows = [[tuple(int(v) for v in it[:3]) for it in row] for row in pool]
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=2, env_id_total=9)
    bins = [ref_port.PortBin(rows, size, rot, bin_id=2 + e, total=9) for e in range(E)]
    robs, rmask = ref.reset()
    obs = [b.observation() for b in bins]
    rng = np.random.RandomState(5)
    for t in range(steps):
        masks = np.stack([ref_port.location_mask(o.astype(np.float32), size, rot) for o in obs]).astype(np.float32)
        np.testing.assert_array_equal(masks, rmask, err_msg="mask t=%d" % t)
        np.testing.assert_array_equal(np.stack(obs).astype(np.float32), robs, err_msg="obs t=%d" % t)
        a = np.array([rng.choice(np.flatnonzero(m)) for m in masks])
        a[rng.rand(E) < 0.1] = size[0] * size[1] - 1                      # some placements that fail
        o = ref.step(a)
        res = [b.step(int(v)) for b, v in zip(bins, a)]
        obs = [r[0] for r in res]
        np.testing.assert_array_equal(np.array([r[1] for r in res], np.float64).astype(np.float32), o["reward"])
        np.testing.assert_array_equal(np.array([r[2] for r in res]), o["done"].astype(bool))
        np.testing.assert_array_equal(np.array([r[3]["counter"] for r in res]), o["counter"])
        np.testing.assert_array_equal(np.array([r[3]["ratio"] for r in res]), o["ratio"])
        robs, rmask = o["obs"], o["mask"]
    np.testing.assert_array_equal(np.stack([b.plain.reshape(-1) for b in bins]), ref.hmap)


def test_python_port_timing_helper_runs():
    pool = sequences.cut2_pool((10, 10, 10), 8, seed=0)
    rate, longest = ref_port.timed_all_cores(pool, (10, 10, 10), False, 0.3, 2)
    assert rate > 50 and longest < 10
