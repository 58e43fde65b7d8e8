"""Host-side logic that needs no GPU: sequence pools, spaces, lazy infos, threshold rewrite."""
import random

import numpy as np
import pytest

import bpp_amd
from bpp_amd import sequences
from bpp_amd.vec_env import LazyInfos
from conftest import load_golden


def test_threshold_integer_rewrite_is_exact():
    """The kernels test 20*ma > 19*area etc. instead of ma/area > 0.95 (float64): identical for every
    window the kernels accept (area <= 1024)."""
    ma, area = np.meshgrid(np.arange(1, 1025), np.arange(1, 1025))
    keep = ma <= area
    ma, area = ma[keep].astype(np.int64), area[keep].astype(np.int64)
    r = ma / area
    assert np.array_equal(r > 0.95, 20 * ma > 19 * area)
    assert np.array_equal(r > 0.85, 20 * ma > 17 * area)
    assert np.array_equal(r > 0.50, 2 * ma > area)


def test_cut2_sequences_are_exact_partitions():
    for size in ((10, 10, 10), (20, 20, 20), (8, 12, 10)):
        for s in range(5):
            seq = sequences.cut2_sequence(size, (2, 5), random.Random(s))
            assert sum(x * y * z for x, y, z in seq) == size[0] * size[1] * size[2]
            assert all(2 <= v <= 5 for it in seq for v in it)


def test_cut2_generator_reproduces_reference_dataset_statistics():
    """Same length/size distribution family as the reference's dataset/cut_2.pt (2100 sequences)."""
    ds = load_golden("cut2_dataset_10")["pool"]
    n_ds = (ds[:, :, 0] != 10).sum(1)
    mine = sequences.cut2_pool((10, 10, 10), 400, seed=123)
    n_me = (mine[:, :, 0] != 10).sum(1)
    assert abs(n_ds.mean() - n_me.mean()) < 1.5 and n_me.min() >= 8 and n_me.max() <= 60
    vols = (mine[:, :, :3].astype(np.int64).prod(2) * (mine[:, :, 0] != 10)).sum(1)
    assert (vols == 1000).all()


def test_pool_format():
    pool = sequences.pad_pool([[(2, 3, 4)], [(1, 1, 1), (2, 2, 2)]], (10, 10, 10))
    assert pool.shape == (2, 3, 4) and pool.dtype == np.uint8
    assert pool[0, 1].tolist() == [10, 10, 10, 0] and pool[1, 2].tolist() == [10, 10, 10, 0]
    with pytest.raises(ValueError):
        sequences.check_pool(np.zeros((2, 3, 4), np.uint8), (10, 10, 10))
    rs = sequences.rs_pool((10, 10, 10), 8, 130, seed=1)
    assert rs.shape == (8, 131, 4) and rs[:, :130, :3].min() >= 2 and rs[:, :130, :3].max() <= 5


def test_spaces_look_like_gym():
    d = bpp_amd.Discrete(200)
    assert d.__class__.__name__ == "Discrete" and d.n == 200 and d.shape == ()
    b = bpp_amd.Box(0.0, 10, (400,))
    assert b.shape == (400,) and b.dtype == np.float32 and b.high.max() == 10


class _FakeTensor(object):
    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def numpy(self):
        return self.a


def test_lazy_infos_match_reference_dict_shape():
    class Env:
        num_envs, act_len, _tstart = 3, 100, 0.0

    class Res:
        done = _FakeTensor(np.array([0, 1, 0], np.uint8))
        counter = _FakeTensor(np.array([3, 11, 0], np.int32))
        ratio = _FakeTensor(np.array([0.1, 0.476, 0.0]))
        ep_ret = _FakeTensor(np.array([1.0, 4.7600000000000001, 0.0]))
        ep_len = _FakeTensor(np.array([3, 12, 1], np.int32))

    infos = LazyInfos(Env(), Res(), 5.0)
    assert len(infos) == 3 and sorted(infos[0].keys()) == ["counter", "ratio"]
    t = infos[1]
    assert sorted(t.keys()) == ["counter", "episode", "mask", "ratio"]
    assert t["episode"] == {"r": 4.76, "l": 12, "t": 5.0} and t["mask"].shape == (100,) and t["counter"] == 11
    assert ["episode" in i for i in infos] == [False, True, False]
    assert infos.done_indices().tolist() == [1]
    assert infos[-1]["counter"] == 0


@pytest.mark.parametrize("size,bound,seed", [((10, 10, 10), (2, 5), 0), ((20, 20, 20), (2, 5), 7), ((8, 12, 10), (2, 4), 123),
                                              ((10, 10, 10), (2, 5), (1 << 32) + 5), ((10, 10, 10), (1, 3), 9)])
def test_native_cut2_generator_equals_python_generator(size, bound, seed):
    """bpp_gen_cut2 (C++, multithreaded, own MT19937) == the pure-Python restatement, which in turn equals the
    reference's MDlayerBoxCreator under random.seed (tests/test_sequences_vs_reference.py): same items, same
    order, for every sequence; seeds above 2^32 exercise the two-word init_by_array key."""
    n = 40
    a = sequences.cut2_pool(size, n, seed=seed, bound=bound, native=True, threads=3)
    b = sequences.cut2_pool(size, n, seed=seed, bound=bound, native=False)
    assert a.shape == b.shape and np.array_equal(a, b)
    # the oracle library exports the same entry point (single-threaded build of the same source)
    import ctypes
    from oracle import oracle as orc
    pool = np.zeros_like(a)
    lengths = np.zeros(n, np.int32)
    assert orc.lib().bpp_gen_cut2(pool.ctypes.data, lengths.ctypes.data, n, a.shape[1], *size, bound[0], bound[1], seed, 1) == 0
    assert np.array_equal(pool, a) and lengths.max() == a.shape[1] - 1
    with pytest.raises(ValueError):
        sequences.cut2_pool(size, 4, seed=seed, bound=bound, T=5)


def test_threshold_magic_division_is_exact():
    """bpp_tile_kernel derives floor(k * area / 20) as (k * area * 0xCCCD) >> 20 in 32-bit arithmetic (k = 17, 19;
    area <= 1024, the largest window a supported bin admits): exact and overflow-free."""
    for area in range(0, 1025):
        for k in (17, 19):
            n = k * area
            assert n * 0xCCCD < 2 ** 32 and (n * 0xCCCD) >> 20 == n // 20


def test_step_tensors_over_one_flat_allocation_makes_views_lazily():
    """BppVecEnv's output sets are ONE allocation; StepTensors creates each view on first use (a step that only looks
    at `obs` pays for one view) and the views alias the allocation."""
    import torch
    from bpp_amd.vec_env import StepTensors
    E, A, M = 5, 8, 8
    regions = {"obs": (0, torch.float32, (E, 4 * A), E * 4 * A * 4), "mask": (1024, torch.float32, (E, M), E * M * 4),
               "reward": (2048, torch.float32, (E, 1), E * 4), "done": (2048 + 24, torch.uint8, (E,), E),
               "_small": (2048, torch.uint8, (64,), 64)}
    flat = torch.zeros(4096, dtype=torch.uint8)
    r = StepTensors(_flat=flat, _layout=regions, _offs={"reward": 0, "done": 24}, _hot=32)
    made = lambda: [k for k in StepTensors.FIELDS if k in [n for n in StepTensors.__slots__ if _has(r, n)]]

    def _has(obj, name):
        try:
            object.__getattribute__(obj, name)
            return True
        except AttributeError:
            return False

    assert made() == []
    assert tuple(r.obs.shape) == (E, 4 * A) and r.obs.dtype == torch.float32 and made() == ["obs"]
    r.obs[2, 3] = 7.0
    assert flat[(2 * 4 * A + 3) * 4:(2 * 4 * A + 3) * 4 + 4].view(torch.float32).item() == 7.0      # aliases the allocation
    assert r["mask"] is r.mask and tuple(r.reward.shape) == (E, 1) and r.done.dtype == torch.uint8
    assert r.counter is None and r.ratio is None                 # not part of this layout
    assert float(r.masks.sum()) == E and tuple(r.bad_masks.shape) == (E, 1)
    with pytest.raises(AttributeError):
        r.nonsense
    # built from ready tensors (tests, emulator front-end): plain attributes
    t = StepTensors(obs=torch.ones(2, 4), done=torch.zeros(2, dtype=torch.uint8))
    assert t.mask is None and float(t.obs.sum()) == 8.0 and t._flat is None
