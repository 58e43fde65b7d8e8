"""Build-container only (needs /root/reference): the host-side item generators reproduce the
reference's creators exactly when fed the same Python RNG stream."""
import contextlib
import io
import random

import numpy as np
import pytest

from bpp_amd import sequences
from oracle import ref_shims

pytestmark = pytest.mark.skipif(not ref_shims.available(), reason="reference tree not present")


@pytest.mark.parametrize("size,n", [((10, 10, 10), 60), ((20, 20, 20), 8), ((20, 20, 10), 10)])
def test_cut2_generator_is_rng_exact(size, n):
    ref_shims.install()
    from envs.bpp0.mdCreator import MDlayerBoxCreator
    with contextlib.redirect_stdout(io.StringIO()):
        cr = MDlayerBoxCreator(size, [2, 5])
        for s in range(n):
            random.seed(1000 + s)
            cr.reset()
            ref = [tuple(b) for b in cr.box_set[:-1]]
            assert sequences.cut2_sequence(size, (2, 5), random.Random(1000 + s)) == ref
            native = sequences.cut2_pool(size, 1, seed=1000 + s, native=True)[0]      # C++ generator, own MT19937
            assert [tuple(int(v) for v in it[:3]) for it in native[:len(ref)]] == ref
            assert tuple(int(v) for v in native[len(ref), :3]) == tuple(size)


@pytest.mark.parametrize("size,rot,n", [((10, 10, 10), False, 40), ((10, 10, 10), True, 20), ((20, 20, 20), False, 4)])
def test_cut1_generator_is_rng_exact(size, rot, n):
    ref_shims.install()
    from envs.bpp0.cutCreator import CuttingBoxCreator
    rng = (2, 2, 2, 5, 5, 5)
    for s in range(n):
        random.seed(77 + s)
        np.random.seed(77 + s)
        cr = CuttingBoxCreator(size, list(rng), rot)     # __init__ cuts once ...
        random.seed(77 + s)
        np.random.seed(77 + s)
        cr.reset()                                        # ... reset() cuts again from the same stream
        ref = []
        while True:
            cr.generate_box_size()
            if tuple(cr.box_list[-1]) == tuple(size) and len(cr.candidates) == 0:
                break
            ref.append(tuple(int(v) for v in cr.box_list[-1]))
        mine = sequences.cut1_sequence(size, rng, random.Random(77 + s), rot, np.random.RandomState(77 + s))
        assert mine == ref
        assert sum(x * y * z for x, y, z in mine) == size[0] * size[1] * size[2]


@pytest.mark.parametrize("name,size", [("cut_2.pt", (10, 10, 10)), ("rs.pt", (10, 10, 10)), ("cut_1.pt", (10, 10, 10))])
def test_from_dataset_plays_what_loadboxcreator_plays(name, size):
    """sequences.from_dataset == the reference's LoadBoxCreator (binCreator.py:42-72): pre-incremented trajectory index,
    stored + appended [10,10,10], (10,10,10) for ever after; checked by draining the live creator for several resets."""
    import os
    ref_shims.install()
    from envs.bpp0.binCreator import LoadBoxCreator
    path = os.path.join(ref_shims.REFERENCE_ROOT, "dataset", name)
    pool = sequences.from_dataset(path, size)
    T = pool.shape[1]
    with contextlib.redirect_stdout(io.StringIO()):
        cr = LoadBoxCreator(path)
    assert pool.shape[0] == cr.traj_nums
    for episode in range(5):
        cr.reset()                                   # index += 1 first: episode 0 plays trajectory 1
        n = T + 7                                    # well past the end of the stored list
        got = [tuple(int(v) for v in b) for b in cr.preview(n)]
        want = [tuple(int(v) for v in pool[episode, min(c, T - 1), :3]) for c in range(n)]
        assert got == want, (name, episode)


def test_from_dataset_npz_fixture_equals_the_reference_file():
    import os
    ref_shims.install()
    a = sequences.from_dataset(os.path.join(ref_shims.REFERENCE_ROOT, "dataset", "cut_2.pt"))
    b = sequences.from_dataset(os.path.join(os.path.dirname(__file__), "golden", "cut2_dataset_10.npz"))
    np.testing.assert_array_equal(a, b)
    assert a.shape == (2100, 48, 4)
