"""the item-sequence generators.  Pinned against sequences the UNMODIFIED reference creators
produced under fixed seeds (tests/golden/sequences_reference.npz, written by tests/golden/make_sequences_golden.py
in the build container): the native generators of the product library (bpp_gen_cut2 / bpp_gen_cut1 / bpp_gen_rs:
exact re-implementations of CPython's random and numpy's legacy RandomState streams) and the Python restatements in
sequences.py must both reproduce them item for item.  Runs anywhere (no GPU work, no reference tree)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from bpp_amd import _lib, sequences


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(GOLDEN, "sequences_reference.npz")))


def rows(ref, name):
    pool, ln = ref[name + "_pool"], ref[name + "_len"]
    return [[tuple(int(v) for v in it) for it in pool[k, :ln[k]]] for k in range(pool.shape[0])]


def pool_rows(pool, size):
    out = []
    for row in pool:
        seq = [tuple(int(v) for v in it[:3]) for it in row]
        assert seq[-1] == tuple(size)                    # the last entry is always the terminator
        out.append(seq)
    return out


@pytest.mark.parametrize("name", ["cut2_10", "cut2_20", "cut2_20x20x10", "cut2_12_b36"])
@pytest.mark.parametrize("native", [True, False])
def test_cut2_generators_equal_the_reference_creator(ref, name, native):
    W, L, H, lo, hi, seed0 = (int(v) for v in ref[name + "_meta"])
    want = rows(ref, name)
    got = pool_rows(sequences.cut2_pool((W, L, H), len(want), seed=seed0, bound=(lo, hi), native=native, threads=3), (W, L, H))
    for k, w in enumerate(want):
        assert got[k][:len(w)] == w and all(it == (W, L, H) for it in got[k][len(w):]), (name, k)


@pytest.mark.parametrize("name", ["cut1_10", "cut1_10_rot", "cut1_20", "cut1_8x12x9_rot"])
@pytest.mark.parametrize("native", [True, False])
def test_cut1_generators_equal_the_reference_creator(ref, name, native):
    m = [int(v) for v in ref[name + "_meta"]]
    size, rg, rot, seed0 = tuple(m[:3]), tuple(m[3:9]), bool(m[9]), m[10]
    want = rows(ref, name)
    got = pool_rows(sequences.cut1_pool(size, len(want), seed=seed0, box_range=rg, rotation=rot, native=native, threads=2), size)
    for k, w in enumerate(want):
        assert got[k][:len(w)] == w and all(it == size for it in got[k][len(w):]), (name, k)
        assert sum(x * y * z for x, y, z in w) == size[0] * size[1] * size[2]


@pytest.mark.parametrize("name", ["rs_default", "rs_args"])
@pytest.mark.parametrize("native", [True, False])
def test_rs_generators_equal_the_reference_creator(ref, name, native):
    which, seed0 = (int(v) for v in ref[name + "_meta"])
    box_set = sequences.DEFAULT_BOX_SET if which == 1 else [(2 + i, 2 + j, 2 + k) for i in range(5) for j in range(5) for k in range(5)]
    want = rows(ref, name)
    got = pool_rows(sequences.rs_pool((10, 10, 10), len(want), len(want[0]), seed=seed0, box_set=box_set, native=native), (10, 10, 10))
    for k, w in enumerate(want):
        assert got[k][:len(w)] == w, (name, k)


def test_native_generators_equal_the_python_restatements_on_more_seeds():
    """Beyond the fixtures: many more seeds and shapes, native (C++ streams) vs Python (`random` / numpy themselves)."""
    for size, bound in (((10, 10, 10), (2, 5)), ((16, 12, 9), (2, 4)), ((20, 20, 20), (3, 7))):
        a = sequences.cut2_pool(size, 40, seed=12345, bound=bound, native=True, T=None)
        b = sequences.cut2_pool(size, 40, seed=12345, bound=bound, native=False, T=a.shape[1])
        np.testing.assert_array_equal(a, b)
    for size, rg, rot in (((10, 10, 10), (2, 2, 2, 5, 5, 5), True), ((12, 10, 8), (2, 2, 1, 6, 5, 3), False),
                          ((9, 9, 9), (1, 1, 1, 3, 3, 3), True)):
        a = sequences.cut1_pool(size, 30, seed=999, box_range=rg, rotation=rot, native=True)
        b = sequences.cut1_pool(size, 30, seed=999, box_range=rg, rotation=rot, native=False, T=a.shape[1])
        np.testing.assert_array_equal(a, b)
    for n_box in (1, 2, 3, 64, 125, 200):
        bs = [(1 + (i % 7), 1 + (i % 5), 1 + (i % 3)) for i in range(n_box)]
        np.testing.assert_array_equal(sequences.rs_pool((10, 10, 10), 20, 50, seed=7, box_set=bs, native=True),
                                      sequences.rs_pool((10, 10, 10), 20, 50, seed=7, box_set=bs, native=False))


def test_generator_argument_validation():
    """Arguments the reference itself cannot handle are refused instead of hanging or reading out of bounds."""
    with pytest.raises(RuntimeError):
        sequences.cut2_pool((5, 5, 5), 2, bound=(2, 5))            # bin already inside the bounds: random.choice([])
    with pytest.raises(RuntimeError):
        sequences.cut2_pool((10, 1, 10), 2, bound=(2, 5))          # a side below the lower bound
    with pytest.raises(RuntimeError):
        sequences.cut1_pool((10, 10, 10), 2, box_range=(3, 3, 3, 4, 4, 4))   # high < 2*low - 1
    with pytest.raises(ValueError):
        sequences.cut2_pool((10, 10, 10), 4, T=5)                  # rows too short
    assert _lib.lib().bpp_gen_rs(None, 1, 4, 10, 10, 10, None, 0, 0, 1) == -1


@pytest.mark.gpu
def test_gpu_box_native_generators_pinned_to_reference_fixtures(ref):
    """The same pin inside the driver-run `-m gpu` suite (the fixtures travel, the reference tree does not)."""
    for name in ("cut2_10", "cut2_20"):
        W, L, H, lo, hi, seed0 = (int(v) for v in ref[name + "_meta"])
        want = rows(ref, name)
        got = pool_rows(sequences.cut2_pool((W, L, H), len(want), seed=seed0, bound=(lo, hi)), (W, L, H))
        assert all(got[k][:len(w)] == w for k, w in enumerate(want))
    from conftest import load_golden
    g = load_golden("rollout_cut2_20")       # its pool = reference CUT-2 output for random.seed(100..111)
    mine = sequences.cut2_pool((20, 20, 20), g["pool"].shape[0], seed=100, T=g["pool"].shape[1])
    np.testing.assert_array_equal(mine, g["pool"])
    m = [int(v) for v in ref["cut1_10_rot_meta"]]
    want = rows(ref, "cut1_10_rot")
    got = pool_rows(sequences.cut1_pool(tuple(m[:3]), len(want), seed=m[10], box_range=tuple(m[3:9]), rotation=True), tuple(m[:3]))
    assert all(got[k][:len(w)] == w for k, w in enumerate(want))


def test_from_dataset_fixture_without_the_reference_tree():
    """sequences.from_dataset on the committed repack of the reference's dataset/cut_2.pt (no /root/reference needed):
    LoadBoxCreator's pre-incremented index -> pool row r is trajectory r + 1 (wrapping), rows end in the terminator,
    every trajectory fills the 10^3 bin exactly (a CUT-2 property)."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "cut2_dataset_10.npz")
    raw = np.load(path)["pool"]
    pool = sequences.from_dataset(path, (10, 10, 10))
    assert pool.shape == raw.shape == (2100, 48, 4)
    np.testing.assert_array_equal(pool[0], raw[1])
    np.testing.assert_array_equal(pool[-1], raw[0])
    np.testing.assert_array_equal(sequences.from_dataset(path, first_index=0), raw)
    term = (pool[:, :, 0] == 10) & (pool[:, :, 1] == 10) & (pool[:, :, 2] == 10)
    assert term[:, -1].all()
    vol = (pool[:, :, 0].astype(int) * pool[:, :, 1] * pool[:, :, 2] * ~term).sum(1)
    assert (vol == 1000).all()
    with pytest.raises(ValueError):
        sequences.from_dataset(path, terminator=(9, 9, 9))
