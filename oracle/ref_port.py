"""TEST INFRASTRUCTURE, NOT THE PRODUCT -- a pure-Python / numpy restatement of what ONE reference worker and the parent's
mask loop do per lock-step, written to cost what the reference costs: the same numpy calls per candidate position (a
window slice, np.max, np.sum(rec == max_h), four corner reads, float64 ratio tests), the same copies per step (heightmap
deep copy on every placement, a fresh stacked observation, a float32 round trip before the mask), Python-level loops
where the reference has them.  It exists because /root/reference cannot travel to the GPU box: bench.py times THIS
(`cpu_baseline.python_port`, all host cores, forked workers) beside the GPU number, and tests/test_ref_port.py pins it --
outputs against the C oracle (itself pinned to the live reference), speed against the live reference in the build
container (oracle/time_reference.py prints both; the port runs within a few percent of it).

Follows envs/bpp0/space.py:36-46,111-144,164-181 (Space.update_height_graph / check_box / drop_box),
envs/bpp0/bin3D.py:49-66,95-127 (observation, step), acktr/utils.py:8-94 (mask rule and the two mask builders),
baselines/common/vec_env/shmem_vec_env.py:126-130 (reset on done), main.py:163-169 (mask per observation)."""
import copy
import time

import numpy as np


def _window(plain, x, y, lx, ly):
    """(max height, cells at it, the four corner heights) of plain[lx:lx+x, ly:ly+y] -- the numpy work both rules share."""
    rec = plain[lx:lx + x, ly:ly + y]
    top = np.max(rec)
    return top, np.sum(rec == top), (rec[0, 0], rec[x - 1, 0], rec[0, y - 1], rec[x - 1, y - 1])


def mask_rule(plain, x, y, lx, ly, z, size):
    """acktr/utils.py:8-35: resting height of an x*y*z box at (lx, ly) under the MASK rule, -1 if not allowed."""
    if lx + x > size[0] or ly + y > size[1] or lx < 0 or ly < 0:
        return -1
    top, cells, corners = _window(plain, x, y, lx, ly)
    if top + z > size[2]:
        return -1
    share = cells / (x * y)
    level = int(corners[0] == top) + int(corners[1] == top) + int(corners[2] == top) + int(corners[3] == top)
    if share > 0.95 or (level == 3 and share > 0.85) or (level == 4 and share > 0.50):
        return top
    return -1


def place_rule(plain, x, y, lx, ly, z, size):
    """envs/bpp0/space.py:111-144: the PLACEMENT rule (three of the four corners must share the highest corner)."""
    if lx + x > size[0] or ly + y > size[1] or lx < 0 or ly < 0:
        return -1
    rec = plain[lx:lx + x, ly:ly + y]
    corners = (rec[0, 0], rec[x - 1, 0], rec[0, y - 1], rec[x - 1, y - 1])
    high = max(corners)
    support = int(corners[0] == high) + int(corners[1] == high) + int(corners[2] == high) + int(corners[3] == high)
    if support < 3:
        return -1
    top = np.max(rec)
    share = np.sum(rec == top) / (x * y)
    if top + z > size[2]:
        return -1
    if share > 0.95 or (high == top and support == 3 and share > 0.85) or (high == top and support == 4 and share > 0.50):
        return top
    return -1


def location_mask(observation, size, rotation):
    """get_possible_position / get_rotation_mask (acktr/utils.py:37-94) of one float32 observation row."""
    info = np.asarray(observation).reshape((4, -1))
    x, y, z = int(info[1][0]), int(info[2][0]), int(info[3][0])
    plain = info[0].reshape((size[0], size[1]))
    halves = []
    for a, b in ((x, y), (y, x)) if rotation else ((x, y),):
        m = np.zeros((size[0], size[1]), np.int32)
        for i in range(size[0] - a + 1):
            for j in range(size[1] - b + 1):
                if mask_rule(plain, a, b, i, j, z, size) >= 0:
                    m[i, j] = 1
        halves.append(m.reshape((-1,)))
    mask = np.hstack(halves) if rotation else halves[0]
    if mask.sum() == 0:
        mask[:] = 1
    return mask


class PortBin(object):
    """One PackingGame behind one vec-env worker, item sequences from a pool row per episode (the deterministic
    replacement for the creator's RNG that the GPU path and the oracle use as well)."""

    def __init__(self, pool, size, rotation, bin_id=0, total=1):
        self.pool, self.size, self.rotation = pool, tuple(int(v) for v in size), bool(rotation)
        self.area = self.size[0] * self.size[1]
        self.bin_id, self.total, self.episode = int(bin_id), int(total), 0
        self._start()

    def _start(self):
        self.plain = np.zeros(self.size[:2], np.int32)
        self.row = self.pool[(self.bin_id + self.episode * self.total) % len(self.pool)]
        self.cursor, self.boxes, self.volume = 0, 0, 0

    @property
    def item(self):
        x, y, z = self.row[min(self.cursor, len(self.row) - 1)][:3]
        return int(x), int(y), int(z)

    def observation(self):
        planes = [np.ones(self.size[:2], np.int32) * v for v in self.item]
        return np.reshape(np.stack((self.plain, *planes)), (-1,))

    def step(self, action):
        """(observation after the worker's auto-reset, reward, done, info) -- bin3D.py:95-127 + shmem_vec_env.py:126-130."""
        idx, turned = int(action), False
        if idx > self.area:
            idx, turned = idx - self.area, True
        lx, ly = idx // self.size[1], idx % self.size[1]
        bx, by, bz = self.item
        x, y = (by, bx) if turned else (bx, by)
        rest = place_rule(self.plain, x, y, lx, ly, bz, self.size)
        binvol = self.size[0] * self.size[1] * self.size[2]
        if rest == -1:
            info = {"counter": self.boxes, "ratio": self.volume / binvol}
            self.episode += 1
            self._start()
            return self.observation(), 0.0, True, info
        plain = copy.deepcopy(self.plain)                      # update_height_graph
        plain[lx:lx + x, ly:ly + y] = max(np.max(plain[lx:lx + x, ly:ly + y]), rest + bz)
        self.plain = plain
        self.boxes += 1
        self.volume += bx * by * bz
        reward = (bx * by * bz) / binvol * 10
        self.cursor += 1
        return self.observation(), reward, False, {"counter": self.boxes, "ratio": self.volume / binvol}


def rollout(pool, size, rotation, seconds, seed=0, bin_id=0, total=1):
    """Uniform-feasible policy on one bin for about `seconds`: lock-steps done and the time they took."""
    env = PortBin(pool, size, rotation, bin_id, total)
    rng = np.random.RandomState(seed)
    obs = env.observation()
    steps, t0 = 0, time.perf_counter()
    while True:
        for _ in range(20):
            mask = location_mask(obs.astype(np.float32), size, rotation)
            obs, _, _, _ = env.step(int(rng.choice(np.flatnonzero(mask))))
        steps += 20
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return steps, dt


def _worker(args):
    pool, size, rotation, seconds, k, n = args
    return rollout(pool, size, rotation, seconds, seed=k, bin_id=k, total=n)


def timed_all_cores(pool, size, rotation, seconds, cores):
    """`cores` forked workers, one bin each (what ShmemVecEnv's workers plus a perfectly parallel mask loop would do)."""
    import multiprocessing as mp
    pool = [[tuple(int(v) for v in it[:3]) for it in row] for row in np.asarray(pool)]
    with mp.get_context("fork").Pool(cores) as p:
        res = p.map(_worker, [(pool, tuple(size), bool(rotation), seconds, k, cores) for k in range(cores)])
    return sum(s / dt for s, dt in res), max(dt for _, dt in res)
