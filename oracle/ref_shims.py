"""TEST INFRASTRUCTURE ONLY -- import shims that let the *unmodified* reference run in
the build container (no `gym`, no `transforms3d` installed there).

Only `tests/`, `tests/golden/make_golden.py`, `oracle/ref_baseline.py` (bench.py's cpu_baseline leg) and
`oracle/time_reference.py` may import this module.  The product package never does.  Nothing here restates reference logic: the shims are the
minimal surface of two third-party packages the reference imports:

* `gym`            -- Env / Wrapper / ObservationWrapper / spaces.{Discrete,Box,Dict,Tuple} /
                      make / envs.registration.register   (reference uses: envs/bpp0/bin3D.py:4,9,40-41,
                      acktr/envs.py:3,6,38-41,95-99, baselines/bench/monitor.py:3,12,
                      baselines/common/vec_env/util.py)
* `transforms3d.euler` -- quat2euler / quat2mat (decorative in envs/bpp0/mdCreator.py:19,33)

`/root/reference` exists only in the build container.  On the GPU box the reference is `oracle/_ref/`: byte-for-byte
copies of the files the env side loads, made by the committed recipe `oracle/make_ref.py` (git-ignored, travels with the
working tree).  `available()`: is either of them there?  `install(root=...)` picks one explicitly; bench.py and the
`-m gpu` tests always pass `REF_COPY` (they must not read /root/reference at run time).
"""
import os
import sys
import types

import numpy as np

REF_COPY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _is_tree(root):
    return bool(root) and os.path.isfile(os.path.join(root, "envs", "bpp0", "bin3D.py"))


def _default_root():
    env = os.environ.get("BPP_REFERENCE_ROOT")
    if env:
        return env
    return "/root/reference" if _is_tree("/root/reference") or not _is_tree(REF_COPY) else REF_COPY


REFERENCE_ROOT = _default_root()


def available():
    return _is_tree(REFERENCE_ROOT)


def copy_available():
    """oracle/_ref/ (made by oracle/make_ref.py) is there."""
    return _is_tree(REF_COPY) and os.path.isfile(os.path.join(REF_COPY, "MANIFEST.json"))


class _Space(object):
    shape = None
    dtype = None


class _Discrete(_Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n


class _Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)


class _DictSpace(_Space):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {})


class _TupleSpace(_Space):
    def __init__(self, spaces=()):
        self.spaces = tuple(spaces)


class _Env(object):
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def seed(self, seed=None):
        return []

    def close(self):
        pass

    def render(self, mode="human"):
        return None

    @property
    def unwrapped(self):
        return self


class _Wrapper(_Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.reward_range = getattr(env, "reward_range", None)
        self.metadata = getattr(env, "metadata", {})

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        return self.env.close()


class _ObservationWrapper(_Wrapper):
    def reset(self, **kwargs):
        return self.observation(self.env.reset(**kwargs))

    def step(self, action):
        ob, rew, done, info = self.env.step(action)
        return self.observation(ob), rew, done, info


_REGISTRY = {}


def _register(id, entry_point=None, **kwargs):
    _REGISTRY[id] = entry_point


def _make(id, **kwargs):
    import importlib
    mod, cls = _REGISTRY[id].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)


def install(root=None):
    """Register the fake modules and put the reference root on sys.path (idempotent; one root per process)."""
    global REFERENCE_ROOT
    if root is not None:
        root = os.path.abspath(root)
        loaded = getattr(sys.modules.get("envs.bpp0"), "__file__", None)
        if loaded and not os.path.abspath(loaded).startswith(root + os.sep):
            raise RuntimeError("reference already imported from %s, cannot switch to %s" % (loaded, root))
        REFERENCE_ROOT = root
    if not available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        spaces = types.ModuleType("gym.spaces")
        box = types.ModuleType("gym.spaces.box")
        core = types.ModuleType("gym.core")
        envs = types.ModuleType("gym.envs")
        reg = types.ModuleType("gym.envs.registration")
        spaces.Discrete, spaces.Box, spaces.Dict, spaces.Tuple = _Discrete, _Box, _DictSpace, _TupleSpace
        spaces.box = box
        box.Box = _Box
        core.Wrapper, core.Env = _Wrapper, _Env
        reg.register = _register
        envs.registration = reg
        gym.Env, gym.Wrapper, gym.ObservationWrapper = _Env, _Wrapper, _ObservationWrapper
        gym.spaces, gym.core, gym.envs, gym.make = spaces, core, envs, _make
        sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.spaces.box": box, "gym.core": core,
                            "gym.envs": envs, "gym.envs.registration": reg})
    if "transforms3d" not in sys.modules:
        t3d = types.ModuleType("transforms3d")
        eu = types.ModuleType("transforms3d.euler")
        eu.quat2euler = lambda q, *a, **k: (0.0, 0.0, 0.0)
        eu.quat2mat = lambda q, *a, **k: np.eye(3)
        t3d.euler = eu
        sys.modules.update({"transforms3d": t3d, "transforms3d.euler": eu})
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _register("Bpp-v0", entry_point="envs.bpp0:PackingGame")


def make_replay_creator(pool, terminator, env_id=0, env_total=1):
    """A `BoxCreator` (reference base class, envs/bpp0/binCreator.py:5-22) that replays explicit
    item sequences: the k-th `reset()` (k = 0, 1, ...) of bin `env_id` selects
    `pool[(env_id + k * env_total) % len(pool)]` -- the same assignment include/bpp_abi.h documents;
    `generate_box_size` appends the next item, or `terminator` past the end.  Injected through
    `PackingGame(box_creator=...)` (envs/bpp0/bin3D.py:10-13) so the reference and the device env
    consume identical items."""
    install()
    from envs.bpp0.binCreator import BoxCreator

    class ReplayBoxCreator(BoxCreator):
        def __init__(self):
            super().__init__()
            self._pool = [[tuple(int(v) for v in it) for it in s] for s in pool]
            self._term = tuple(int(v) for v in terminator)
            self._episode = -1
            self._cursor = 0

        def reset(self):
            self.box_list.clear()
            self._episode += 1
            self._cursor = 0

        def generate_box_size(self, **kwargs):
            seq = self._pool[(env_id + self._episode * env_total) % len(self._pool)]
            self.box_list.append(seq[self._cursor] if self._cursor < len(seq) else self._term)
            self._cursor += 1

    return ReplayBoxCreator()
