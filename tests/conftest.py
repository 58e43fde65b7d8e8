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


ROLLOUT_CASES = ["rollout_cut2_10", "rollout_cut2_10_rot", "rollout_cut2_20", "rollout_rs_10",
                 "rollout_wide_8x12x9_rot", "rollout_short_5x4x6"]
MASK_CASES = ["masks_10", "masks_20", "masks_7x13x8"]


@pytest.fixture(scope="session")
def emu():
    """The product kernels compiled for the host SIMT emulator (tests/emu/): numpy front-end with the
    oracle's API, bound to the emulated libbpp_hip."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu_binding
    return emu_binding.load()
