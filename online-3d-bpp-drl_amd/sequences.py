"""Item-sequence pools fed to the device env (host side; the step kernels only index the pool).

Pool format (include/bpp_abi.h): uint8 [P][T][4] = (x, y, z, 0); every row is padded with a terminator
item and its LAST entry is always a terminator, which is what a cursor >= T keeps returning.

Generators restate the reference's item creators.  They draw from a private `random.Random(seed)`,
consuming it in exactly the reference's order, so under `random.seed(seed)` the reference creator
produces the identical sequence (tests/test_sequences_vs_reference.py checks this in the build
container).
"""
import random

import numpy as np


def pad_pool(seqs, terminator, T=None):
    """List of item lists -> uint8 [P][T][4] padded with `terminator`."""
    if T is None:
        T = max(len(s) for s in seqs) + 1
    term = tuple(int(v) for v in terminator)
    pool = np.zeros((len(seqs), T, 4), np.uint8)
    pool[:, :, 0], pool[:, :, 1], pool[:, :, 2] = term
    for p, s in enumerate(seqs):
        if len(s) > T - 1:
            raise ValueError("sequence %d has %d items, pool rows hold %d + terminator" % (p, len(s), T - 1))
        if len(s):
            pool[p, :len(s), :3] = np.asarray(s, dtype=np.int64).reshape(-1, 3)
    return pool


def check_pool(pool, container_size):
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    if pool.ndim != 3 or pool.shape[2] != 4 or pool.shape[0] < 1 or pool.shape[1] < 1:
        raise ValueError("pool must be uint8 [P][T][4]")
    if (pool[:, :, :3] == 0).any():
        raise ValueError("pool holds a zero-sized item")
    return pool


def from_dataset(path, container_size=(10, 10, 10), terminator=(10, 10, 10), first_index=1):
    """Pool from a stored trajectory set -- what the reference's LoadBoxCreator plays (envs/bpp0/binCreator.py:42-72,
    `--load-dataset --data-name cut_2.pt`): `path` is one of the reference's dataset/*.pt files (a torch-saved list of
    trajectories, each a list of [x, y, z]) or an .npz holding the same trajectories already packed as uint8 [n][T][4]
    `pool` in file order (tests/golden/cut2_dataset_10.npz = dataset/cut_2.pt).

    LoadBoxCreator semantics kept: `reset()` PRE-increments its index (:54-55), so the first episode plays
    trajectory 1, not 0 -> pool row r = trajectory (first_index + r) mod n, and a single bin (env_id_total = 1) plays
    the trajectories in the creator's order; the creator appends [10, 10, 10] to the stored list (:62; cut_2.pt
    already stores one at the end of every trajectory) and keeps yielding (10, 10, 10) once the list is used up
    (:69-72) -> rows are padded with `terminator`, literally (10, 10, 10) as in the reference whatever the bin (pass
    the bin size for a terminator that can never be placed).  Beyond the reference: it raises IndexError after the
    last trajectory, the pool wraps around."""
    term = tuple(int(v) for v in terminator)
    if str(path).endswith(".npz"):
        pool = check_pool(np.load(path)["pool"], container_size)
        tail = pool[:, -1, :3]
        if not (tail == np.asarray(term, dtype=np.uint8)).all():
            raise ValueError("%s: the last entry of every row must be the terminator %r" % (path, term))
        return np.ascontiguousarray(np.roll(pool, -int(first_index), axis=0))
    import torch
    trajs = torch.load(path, weights_only=True)      # a list of lists of ints: no pickled code is ever needed or run
    n = len(trajs)
    seqs = [[tuple(int(v) for v in it) for it in trajs[(int(first_index) + r) % n]] for r in range(n)]
    for q in seqs:                   # a stored trailing terminator is the same thing as the padding
        while q and q[-1] == term:
            q.pop()
    return check_pool(pad_pool(seqs, term), container_size)


class CounterRandom(object):
    """The counter-based generator of BPP_STREAM_RNG_COUNTER (normative definition: include/bpp_abi.h) with the two
    methods the cutting algorithm calls on `random.Random`: episode `episode` of stream id `sid` under `seed0` is
    cut2_sequence(size, bound, CounterRandom(seed0, sid, episode)).  Third statement of that generator, beside the oracle
    library's C and the device kernels."""
    M = 0xFFFFFFFF

    @classmethod
    def fmix32(cls, x):
        x &= cls.M
        x ^= x >> 16
        x = (x * 0x85ebca6b) & cls.M
        x ^= x >> 13
        x = (x * 0xc2b2ae35) & cls.M
        x ^= x >> 16
        return x

    def __init__(self, seed0, sid, episode):
        f, M = self.fmix32, self.M
        h = f((seed0 & M) + 0x9E3779B9)
        h = f(h ^ ((seed0 >> 32) & M))
        h = f(h ^ (sid & M))
        h = f(h ^ ((sid >> 32) & M))
        self.klo = f(h ^ (episode & M))
        self.khi = f(self.klo + 0x7F4A7C15 + (episode & M))
        self.n = 0

    def _word(self, a):
        M = self.M
        return self.fmix32(((self.klo + self.n * 0x9E3779B9) & M) ^ ((self.khi + a * 0x85EBCA77) & M))

    def below(self, lim):
        a = 0
        m = self._word(a) * lim
        if (m & self.M) < lim:
            t = ((1 << 32) - lim) % lim
            while (m & self.M) < t:
                a += 1
                m = self._word(a) * lim
        self.n += 1
        return m >> 32

    def choice(self, seq):
        return seq[self.below(len(seq))]

    def randint(self, a, b):
        return a + self.below(b - a + 1)


class _Cut(object):
    """A cuboid being cut (identity semantics, like the reference's Box objects)."""
    __slots__ = ("x", "y", "z", "low", "high")

    def __init__(self, x, y, z, low, high):
        self.x, self.y, self.z, self.low, self.high = x, y, z, low, high


def cut2_sequence(container_size, bound, rng):
    """One CUT-2 item sequence: cut the bin into boxes with every side in [bound[0], bound[1]], then
    sort by the height of the box's base (stable).  Restates envs/bpp0/mdCreator.py:59-100
    (Box.benchmark_split), :117-135 (bin.gen_benchmark, including its iterate-while-mutating list walk)
    and :137-138 (depart_box).  Returns [(x, y, z), ...] without any terminator."""
    lo, hi = bound
    W, L, H = container_size
    valid = []
    invalid = [_Cut(W, L, H, 0, H)]
    while True:
        i = 0
        while i < len(invalid):          # `for box in invalid_box` with remove/append inside
            b = invalid[i]
            i += 1
            flags = []                   # mdCreator.py:60-66
            if b.x > hi:
                flags.append(0)
            if b.y > hi:
                flags.append(1)
            if b.z > hi:
                flags.append(2)
            f = rng.choice(flags)        # :68
            if f == 0:                   # :70-79
                if b.x <= lo:
                    continue
                r = rng.randint(1, b.x)
                if r < lo or b.x - r < lo:
                    continue
                subs = (_Cut(r, b.y, b.z, b.low, b.high), _Cut(b.x - r, b.y, b.z, b.low, b.high))
            elif f == 1:                 # :80-89
                if b.y < lo:
                    continue
                r = rng.randint(1, b.y)
                if r < lo or b.y - r < lo:
                    continue
                subs = (_Cut(b.x, r, b.z, b.low, b.high), _Cut(b.x, b.y - r, b.z, b.low, b.high))
            else:                        # :90-99
                if b.z < lo:
                    continue
                r = rng.randint(1, b.z)
                if r < lo or b.z - r < lo:
                    continue
                subs = (_Cut(b.x, b.y, b.z - r, b.low, b.high - r), _Cut(b.x, b.y, r, b.high - r, b.high))
            # :124-130: the split box leaves the list (shifting the rest left under the iterator)
            del invalid[i - 1]
            for s in subs:
                if lo <= s.x <= hi and lo <= s.y <= hi and lo <= s.z <= hi:
                    valid.append(s)
                else:
                    invalid.append(s)
        if not invalid:
            break
    valid.sort(key=lambda c: c.low)      # :137-138
    return [(c.x, c.y, c.z) for c in valid]


def cut2_pool(container_size, n, seed=0, bound=(2, 5), T=None, native=True, threads=0):
    """P = n CUT-2 sequences, sequence k drawn from random.Random(seed + k).  `native=True` runs the
    multithreaded C++ generator of the library (bpp_gen_cut2, bit-identical output: it re-implements CPython's
    MT19937 stream); `native=False` is the pure-Python restatement above."""
    if native:                        # no silent fallback: a missing/broken library raises
        return _cut2_pool_native(container_size, n, seed, bound, T, threads)
    seqs = [cut2_sequence(container_size, bound, random.Random(seed + k)) for k in range(n)]
    return pad_pool(seqs, container_size, T)


def _cut2_pool_native(container_size, n, seed, bound, T, threads):
    import ctypes
    from . import _lib
    W, L, H = (int(v) for v in container_size)
    lo, hi = (int(v) for v in bound)
    lib = _lib.lib()
    cap = T if T is not None else W * L * H // (lo ** 3) + 2          # upper bound on items + terminator
    pool = np.zeros((n, cap, 4), np.uint8)
    lengths = np.zeros(n, np.int32)
    rc = lib.bpp_gen_cut2(pool.ctypes.data, lengths.ctypes.data, n, cap, W, L, H, lo, hi, int(seed), int(threads))
    if rc == -2 and T is not None:
        raise ValueError("a sequence has %d items, pool rows hold %d + terminator" % (int(lengths.max()), T - 1))
    _lib.check(rc)
    if T is None:
        pool = np.ascontiguousarray(pool[:, :int(lengths.max()) + 1])
    return pool


class _Meta(object):
    __slots__ = ("x", "y", "z", "lx", "ly", "lz")

    def __init__(self, x, y, z, lx, ly, lz):
        self.x, self.y, self.z, self.lx, self.ly, self.lz = x, y, z, lx, ly, lz


def cut1_sequence(container_size, box_range, rng, rotation=False, np_rng=None):
    """One CUT-1 item sequence.  Restates envs/bpp0/cutCreator.py:32-128: guillotine-cut the bin until
    every piece is inside `box_range` = (low_x, low_y, low_z, high_x, high_y, high_z) (`_cut_box`,
    :78-95, `_choose_pos` :58-76), then repeatedly draw a random piece whose support is complete
    (`_add_candidate` :97-106, `generate_box_size` :111-128).  The draw order does not depend on where the
    agent puts the items, so the whole sequence can be generated up front.  With `rotation` the reference
    swaps x/y of an item when np.random.rand() >= 0.5 (:119-125); pass `np_rng` (numpy RandomState)."""
    low_x, low_y, low_z, high_x, high_y, high_z = box_range
    W, L, H = container_size
    meta = [_Meta(W, L, H, 0, 0, 0)]
    again = True
    while again:                                     # cutCreator.py:78-95
        again = False
        new = []
        for b in meta:
            check = ((b.x < low_x or b.x > high_x) * 1 + (b.y < low_y or b.y > high_y) * 2 +
                     (b.z < low_z or b.z > high_z) * 4)
            if check == 0:
                new.append(b)
                continue
            df_list = [d for d, bit in ((0, 1), (1, 2), (2, 4)) if check & bit]
            df = rng.choice(df_list)                 # :66
            if df == 0:
                lo, hi = low_x, b.x - low_x
            elif df == 1:
                lo, hi = low_y, b.y - low_y
            else:
                lo, hi = low_z, b.z - low_z
            assert lo <= hi
            pos = rng.randint(lo, hi)                # :75
            if df == 0:
                new += [_Meta(pos, b.y, b.z, b.lx, b.ly, b.lz), _Meta(b.x - pos, b.y, b.z, b.lx + pos, b.ly, b.lz)]
            elif df == 1:
                new += [_Meta(b.x, pos, b.z, b.lx, b.ly, b.lz), _Meta(b.x, b.y - pos, b.z, b.lx, b.ly + pos, b.lz)]
            else:
                new += [_Meta(b.x, b.y, pos, b.lx, b.ly, b.lz), _Meta(b.x, b.y, b.z - pos, b.lx, b.ly, b.lz + pos)]
            again = True
        meta = new
    plain = np.zeros((W, L), np.int32)
    cands = []

    def add_candidates():                            # :97-106
        nonlocal meta
        rest = []
        for mb in meta:
            if (plain[mb.lx:mb.lx + mb.x, mb.ly:mb.ly + mb.y] == mb.lz).sum() == mb.x * mb.y:
                cands.append(mb)
            else:
                rest.append(mb)
        meta = rest

    add_candidates()
    seq = []
    while cands:                                     # :111-128
        b = cands.pop(rng.randint(0, len(cands) - 1))
        if rotation and np_rng.rand() >= 0.5:
            seq.append((b.y, b.x, b.z))
        else:
            seq.append((b.x, b.y, b.z))
        plain[b.lx:b.lx + b.x, b.ly:b.ly + b.y] += b.z
        add_candidates()
    return seq


def cut1_pool(container_size, n, seed=0, box_range=(2, 2, 2, 5, 5, 5), rotation=False, T=None, native=True, threads=0):
    """P = n CUT-1 sequences; sequence k consumes random.Random(seed + k) and -- for the rotation coin -- numpy's
    legacy RandomState(seed + k), i.e. what the reference's CuttingBoxCreator yields after random.seed(s);
    np.random.seed(s); reset().  `native=True`: the library's multithreaded C++ generator (bpp_gen_cut1, exact
    re-implementations of both streams); `native=False`: the Python restatement above."""
    if native:
        return _pool_native("cut1", container_size, n, seed, T, threads, box_range=box_range, rotation=rotation)
    seqs = [cut1_sequence(container_size, box_range, random.Random(seed + k), rotation,
                          np.random.RandomState(seed + k) if rotation else None) for k in range(n)]
    return pad_pool(seqs, container_size, T)


DEFAULT_BOX_SET = [(i, j, k) for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)]   # acktr/arguments.py:122-128


def rs_pool(container_size, n, length, seed=0, box_set=None, native=True, threads=0):
    """RS sequences (envs/bpp0/binCreator.py:24-40): row k = the first `length` draws
    box_set[np.random.randint(0, len(box_set))] after np.random.seed(seed + k), then the terminator.  `native=True`:
    the library's generator (bpp_gen_rs, numpy's legacy MT19937 stream re-implemented); `native=False`: numpy itself."""
    if box_set is None:
        box_set = DEFAULT_BOX_SET
    if native:
        return _pool_native("rs", container_size, n, seed, length + 1, threads, box_set=box_set)
    bs = np.asarray(box_set, dtype=np.int64)
    pool = np.zeros((n, length + 1, 4), np.uint8)
    for k in range(n):
        rng = np.random.RandomState(seed + k)
        pool[k, :length, :3] = bs[[rng.randint(0, len(bs)) for _ in range(length)]]
    pool[:, length, :3] = container_size
    return pool


def _pool_native(kind, container_size, n, seed, T, threads, box_range=None, rotation=False, box_set=None):
    import ctypes
    from . import _lib
    W, L, H = (int(v) for v in container_size)
    lib = _lib.lib()
    if kind == "rs":
        bs = np.ascontiguousarray(np.asarray(box_set, dtype=np.int32).reshape(-1, 3))
        pool = np.zeros((n, T, 4), np.uint8)
        _lib.check(lib.bpp_gen_rs(pool.ctypes.data, n, T, W, L, H, bs.ctypes.data, bs.shape[0], int(seed), int(threads)))
        return pool
    rg = np.ascontiguousarray(np.asarray(box_range, dtype=np.int32).reshape(6))
    cap = T if T is not None else W * L * H // max(1, int(rg[0]) * int(rg[1]) * int(rg[2])) + 2
    pool = np.zeros((n, cap, 4), np.uint8)
    lengths = np.zeros(n, np.int32)
    rc = lib.bpp_gen_cut1(pool.ctypes.data, lengths.ctypes.data, n, cap, W, L, H, rg.ctypes.data, int(bool(rotation)), int(seed),
                          int(threads))
    if rc == -2 and T is not None:
        raise ValueError("a sequence has %d items, pool rows hold %d + terminator" % (int(lengths.max()), T - 1))
    _lib.check(rc)
    if T is None:
        pool = np.ascontiguousarray(pool[:, :int(lengths.max()) + 1])
    return pool
