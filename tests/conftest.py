import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure), built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


# rollout_deep_* / rollout_pretrained_* (round 6): recorded from the reference under a COMPETENT policy -- the lowest-top
# heuristic of oracle/policies.py and the reference's own pretrained checkpoints played greedily (make_golden.py --deep)
DEEP_CASES = ["rollout_deep_cut2_10", "rollout_deep_cut2_10_rot", "rollout_deep_cut2_20",
              "rollout_pretrained_cut2_10", "rollout_pretrained_cut2_10_rot"]
ROLLOUT_CASES = ["rollout_cut2_10", "rollout_cut2_10_rot", "rollout_cut2_20", "rollout_rs_10",
                 "rollout_wide_8x12x9_rot", "rollout_short_5x4x6"] + DEEP_CASES
MASK_CASES = ["masks_10", "masks_20", "masks_7x13x8"]


def depth_profile(g):
    """(share of recorded env-steps on bins that already hold >= 20 boxes, number of COMPLETELY packed bins -- final ratio
    1.0 with the terminator as the current item --, mean final ratio) of a recording in the golden format."""
    import numpy as np
    W, L, H = (int(v) for v in g["size"])
    A = W * L
    d = g["done"].astype(bool)
    full = 0
    for t, e in zip(*np.nonzero(d & (g["ratio"] == 1.0))):
        prev = g["obs0"][e] if t == 0 else g["obs"][t - 1][e]
        assert (int(prev[A]), int(prev[2 * A]), int(prev[3 * A])) == (W, L, H)    # what failed was the terminator
        full += 1
    return float((g["counter"] >= 20).mean()), full, float(g["ratio"][d].mean())


@pytest.fixture(scope="session")
def emu():
    """The product kernels compiled for the host SIMT emulator (tests/emu/): numpy front-end with the
    oracle's API, bound to the emulated libbpp_hip."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu_binding
    return emu_binding.load()
