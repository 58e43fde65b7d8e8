"""CPU: the C oracle vs rollouts recorded at test time from oracle/_ref/ -- the copy of the reference that travels to
the GPU box (tests/live_reference.py).  The GPU twin of this file replays the same kind of recordings on the HIP path."""
import numpy as np
import pytest

import live_reference
from oracle import ref_shims
from test_oracle_golden import check_rollout

pytestmark = pytest.mark.skipif(not ref_shims.copy_available(), reason="oracle/_ref/ not made (python oracle/make_ref.py)")


@pytest.fixture(scope="module")
def recordings(tmp_path_factory):
    return live_reference.record_all(str(tmp_path_factory.mktemp("live_ref")))


class _ScatteredOracle(object):
    """The recorded bins are bins `ids` of a full-size oracle env (the GPU twin does the same with the 65 536-bin launch)."""

    def __init__(self, oracle, pool, size, rot, ids, total, rule):
        self.o, self.env, self.ids, self.t = oracle, oracle.OracleEnv(pool, size, rot, total, mask_rule=rule), np.asarray(ids), 0

    def reset(self):
        obs, mask = self.env.reset()
        self.mask = mask
        return obs[self.ids], mask[self.ids]

    def step(self, actions):
        a = self.o.sample_feasible(self.mask, 77, self.t)
        a[self.ids] = actions
        self.t += 1
        out = self.env.step(a)
        self.mask = out["mask"]
        return {k: v[self.ids] for k, v in out.items()}


@pytest.mark.parametrize("case", sorted(live_reference.CASES))
def test_oracle_replays_live_reference_recording(oracle, recordings, case):
    g = dict(np.load(recordings[case]))
    if "env_ids" in g:
        check_rollout(lambda pool, size, rot, E, rule: _ScatteredOracle(oracle, pool, size, rot, g["env_ids"], int(g["env_total"]), rule), g)
    else:
        check_rollout(lambda pool, size, rot, E, rule: oracle.OracleEnv(pool, size, rot, E, mask_rule=rule), g)
    assert g["done"].sum() > live_reference.min_episodes(case, g)     # a real number of episodes went through the recording
    live_reference.check_depth(case, g)                               # ... and the deep cases hold deep states


def test_ref_copy_is_byte_identical_to_the_reference_tree():
    """oracle/_ref/ against its manifest -- and against /root/reference where that exists (the build container)."""
    from oracle import make_ref
    assert make_ref.check() == []
