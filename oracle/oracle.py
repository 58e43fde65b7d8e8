"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/bpp_oracle.c (the CPU restatement of the
reference environment step).  Imported by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg, never by the product package.  numpy in, numpy out; no torch.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "bpp_oracle.c")
LIB = os.path.join(HERE, "libbpp_oracle.so")
HDR = os.path.join(os.path.dirname(HERE), "include", "bpp_abi.h")

RULE_UTILS, RULE_SPACE = 0, 1
RESET_INIT, RESET_ADVANCE = 0, 1

STATE_DTYPE = np.dtype([("cursor", "<i4"), ("episode", "<i4"), ("n_boxes", "<i4"), ("vol_sum", "<i4"),
                        ("ep_ret", "<f8"), ("ep_len", "<i4"), ("seq", "<i4"), ("item_cur", "<u4"), ("item_next", "<u4"),
                        ("item_reset", "<u4"), ("hmax", "<u4")])
assert STATE_DTYPE.itemsize == 48


class Batch(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int32), ("W", ctypes.c_int32), ("L", ctypes.c_int32), ("H", ctypes.c_int32),
                ("rotation", ctypes.c_int32), ("mask_rule", ctypes.c_int32), ("pool_size", ctypes.c_int32),
                ("pool_len", ctypes.c_int32), ("env_id_base", ctypes.c_int64), ("env_id_total", ctypes.c_int64),
                ("seq_pool", ctypes.c_void_p), ("hmap", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("ep_acc", ctypes.c_void_p), ("pool_mode", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                ("seq_cache", ctypes.c_void_p)]


class Stream(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int32), ("depth", ctypes.c_int32), ("pool_len", ctypes.c_int32), ("W", ctypes.c_int32),
                ("L", ctypes.c_int32), ("H", ctypes.c_int32), ("bound_lo", ctypes.c_int32), ("bound_hi", ctypes.c_int32),
                ("env_id_base", ctypes.c_int64), ("seed0", ctypes.c_uint64), ("ring", ctypes.c_void_p), ("mt", ctypes.c_void_p),
                ("work", ctypes.c_void_p), ("gen_next", ctypes.c_void_p), ("state", ctypes.c_void_p), ("overflow", ctypes.c_void_p),
                ("rng", ctypes.c_int32), ("reserved1", ctypes.c_int32)]


class StepOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len",
                                               "next_action")] + [("sample_seed", ctypes.c_uint64),
                                                                  ("sample_step", ctypes.c_uint64),
                                                                  ("host_reward", ctypes.c_void_p), ("host_done", ctypes.c_void_p)]


class Knobs(ctypes.Structure):
    _fields_ = [("bins_per_wave", ctypes.c_int32), ("waves_per_group", ctypes.c_int32), ("xcd_remap", ctypes.c_int32),
                ("force_generic", ctypes.c_int32), ("ablate", ctypes.c_int32), ("legacy_fast", ctypes.c_int32),
                ("tile_groups", ctypes.c_int32), ("stream_legacy", ctypes.c_int32), ("stream_overlap", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 3)]


def set_knobs(bins_per_wave=0, waves_per_group=0, xcd_remap=1, force_generic=0, legacy_fast=0, tile_groups=0, stream_legacy=0, stream_overlap=1):
    """bpp_set_knobs of whatever library this front-end is bound to (meaningful for the emulated product)."""
    k = Knobs(int(bins_per_wave), int(waves_per_group), int(xcd_remap), int(force_generic), 0, int(legacy_fast), int(tile_groups), int(stream_legacy), int(stream_overlap))
    _check(lib().bpp_set_knobs(ctypes.byref(k)))


def launch_info(E, size, rotation=False):
    out = (ctypes.c_int32 * 6)()
    _check(lib().bpp_launch_info(int(E), int(size[0]), int(size[1]), int(size[2]), int(bool(rotation)), out))
    return [int(v) for v in out]


def build(force=False):
    """gcc the restatement into oracle/libbpp_oracle.so (no-op when up to date)."""
    if (not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC)
            and os.path.getmtime(LIB) >= os.path.getmtime(HDR)):
        return LIB
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-Wall", "-Wextra", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        L.bpp_last_error.restype = ctypes.c_char_p
        L.bpp_reset.argtypes = [ctypes.POINTER(Batch), ctypes.c_int32, ctypes.POINTER(StepOut), ctypes.c_void_p]
        L.bpp_step.argtypes = [ctypes.POINTER(Batch), ctypes.c_void_p, ctypes.POINTER(StepOut), ctypes.c_void_p]
        L.bpp_mask_from_obs.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 6 + [ctypes.c_void_p]
        L.bpp_mask_from_hmap.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 6 + [ctypes.c_void_p]
        L.bpp_sample_feasible.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.bpp_rollout_uniform.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(StepOut), ctypes.c_void_p, ctypes.c_uint64,
                                          ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_rollout_uniform_sets.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(StepOut), ctypes.c_int32, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_void_p]
        L.bpp_epsilon_override.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64,
                                           ctypes.c_uint32, ctypes.c_void_p]
        L.bpp_masked_act.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64,
                                     ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_masked_act_counter.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_int32, ctypes.c_void_p]
        L.bpp_masked_evaluate.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_masked_evaluate_backward.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_gen_cut2.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 7 + [ctypes.c_uint64, ctypes.c_int32]
        L.bpp_gen_cut1.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 5 + [ctypes.c_void_p, ctypes.c_int32,
                                                                                     ctypes.c_uint64, ctypes.c_int32]
        L.bpp_gen_rs.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 5 + [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64,
                                                                          ctypes.c_int32]
        L.bpp_stream_sizes.argtypes = [ctypes.POINTER(Stream), ctypes.POINTER(ctypes.c_int64)]
        L.bpp_stream_init.argtypes = [ctypes.POINTER(Stream), ctypes.c_void_p]
        L.bpp_stream_refill.argtypes = [ctypes.POINTER(Stream), ctypes.c_void_p]
        L.bpp_rollout_uniform_stream.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(StepOut), ctypes.c_void_p, ctypes.c_uint64,
                                                 ctypes.c_uint64, ctypes.c_int32, ctypes.POINTER(Stream), ctypes.c_int32,
                                                 ctypes.c_void_p, ctypes.c_void_p]
        L.bpp_side_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        L.bpp_side_destroy.argtypes = [ctypes.c_void_p]
        L.bpp_episode_stats.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.bpp_episode_acc_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle error %d: %s" % (rc, lib().bpp_last_error().decode()))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _aligned_zeros(nbytes, align=64):
    raw = np.zeros(nbytes + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes]


def _stream_cache(stream, depth):
    """Same default as BppVecEnv: a row cache wherever the refill schedule leaves the extra row of look-ahead."""
    if stream.get("cache") is not None:
        return bool(stream["cache"])
    return depth >= 5 and depth - int(stream.get("refill_every", 1)) >= 4


class OracleEnv(object):
    """E bins stepped in lock-step on the host by the C restatement."""

    def __init__(self, pool, size, rotation, num_envs, env_id_base=0, env_id_total=None, mask_rule=RULE_UTILS, stream=None):
        """stream=dict(bound=(lo, hi), seed=s, depth=D[, pool_len=T]): endless CUT-2 supply through a ring pool
        (include/bpp_abi.h: bpp_stream) instead of `pool` (pass pool=None)."""
        self.W, self.L, self.H = (int(v) for v in size)
        self.stream = None
        if stream is not None:
            lo, hi = (int(v) for v in stream.get("bound", (2, 5)))
            D = int(stream.get("depth", 8))
            T = int(stream.get("pool_len", self.W * self.L * self.H // lo ** 3 + 3))   # 2 look-ahead entries + items + terminator
            pool = np.zeros((D * int(num_envs), T, 4), np.uint8)
            # bpp_batch.seq_cache (include/bpp_abi.h): the row cache of the device library; the restatement ignores it, the
            # emulated product (tests/emu loads this class over its own library) runs its cache-keeping kernels with it
            self.seq_cache = _aligned_zeros(int(num_envs) * (2 * 128 + 8 + 8), 128) if _stream_cache(stream, D) else None
        self.pool = np.ascontiguousarray(pool, dtype=np.uint8)
        assert self.pool.ndim == 3 and self.pool.shape[2] == 4
        self.A = self.W * self.L
        self.rotation = int(bool(rotation))
        self.M = self.A * (1 + self.rotation)
        self.E = int(num_envs)
        self.hmap = np.zeros((self.E, self.A), np.uint8)
        self.state = np.zeros(self.E, STATE_DTYPE)
        self.out = dict(obs=np.zeros((self.E, 4 * self.A), np.float32), mask=np.zeros((self.E, self.M), np.float32),
                        reward=np.zeros(self.E, np.float32), done=np.zeros(self.E, np.uint8),
                        counter=np.zeros(self.E, np.int32), ratio=np.zeros(self.E, np.float64),
                        ep_ret=np.zeros(self.E, np.float64), ep_len=np.zeros(self.E, np.int32))
        self.ep_acc = _aligned_zeros(self.E * 32).view(np.float64).reshape(self.E, 4)   # per-bin episode accumulators
        self._b = Batch(self.E, self.W, self.L, self.H, self.rotation, int(mask_rule), self.pool.shape[0],
                        self.pool.shape[1], int(env_id_base),
                        int(env_id_total if env_id_total is not None else env_id_base + self.E),
                        _p(self.pool).value, _p(self.hmap).value, _p(self.state).value, _p(self.ep_acc).value,
                        1 if stream is not None else 0, 0,
                        _p(self.seq_cache).value if stream is not None and self.seq_cache is not None else None)
        if stream is not None:
            E = self.E
            self.gen_next = np.zeros(E, np.int32)
            self.overflow = np.zeros(1, np.int32)
            self.stream = Stream(E, D, T, self.W, self.L, self.H, lo, hi, int(env_id_base), int(stream.get("seed", 0)),
                                 _p(self.pool).value, None, None, _p(self.gen_next).value,
                                 _p(self.state).value, _p(self.overflow).value,
                                 {"mt19937": 0, "counter": 1}[stream.get("rng", "mt19937")], 0)
            sizes = (ctypes.c_int64 * 2)()          # the two opaque buffers are sized by the library in use
            _check(lib().bpp_stream_sizes(ctypes.byref(self.stream), sizes))
            self._mt = _aligned_zeros(int(sizes[0]) * 4)
            self._work = _aligned_zeros(int(sizes[1]))
            self.stream.mt, self.stream.work = _p(self._mt).value, _p(self._work).value
            self.refill_every = int(stream.get("refill_every", max(1, D - (4 if self.seq_cache is not None else 3))))
            self._since_refill = 0
            _check(lib().bpp_stream_init(ctypes.byref(self.stream), None))
            self.refill()
        self._o = StepOut(*[_p(self.out[k]).value for k in ("obs", "mask", "reward", "done", "counter", "ratio",
                                                              "ep_ret", "ep_len")])
        self._first = True

    def reset_seq_cache(self):
        """`state` or the ring were written behind the library's back: zero the row cache (include/bpp_abi.h)."""
        self.seq_cache[:] = 0

    def refill(self):
        _check(lib().bpp_stream_refill(ctypes.byref(self.stream), None))
        self._since_refill = 0

    def _after_steps(self, n):
        if self.stream is not None:
            self._since_refill += n
            if self._since_refill >= self.refill_every:
                self.refill()

    def reset(self):
        if self.stream is not None and not self._first:
            self.refill()                      # a VecEnv.reset() advances every bin by one episode
        _check(lib().bpp_reset(ctypes.byref(self._b), RESET_INIT if self._first else RESET_ADVANCE,
                               ctypes.byref(self._o), None))
        self._first = False
        self._after_steps(1)
        return self.out["obs"].copy(), self.out["mask"].copy()

    def episode_stats(self, reset=False, wide=False):
        """float64 [4]: the per-bin accumulators summed in the ABI's fixed order (bpp_episode_acc_reduce).  wide: hand the
        library a scratch buffer (the oracle ignores it; the emulated product then runs its many-workgroup form)."""
        acc = np.zeros(4, np.float64)
        scratch = np.zeros(1024 * 4 + 8, np.float64) if wide else None
        _check(lib().bpp_episode_acc_reduce(_p(self.ep_acc), self.E, _p(acc), int(bool(reset)),
                                            _p(scratch) if wide else None, None))
        if wide:
            assert scratch[1024 * 4:].view(np.uint32)[0] == 0, "the arrival counter must be left at zero"
        return acc

    def step(self, actions, copy=True):
        a = np.ascontiguousarray(np.asarray(actions).reshape(-1), dtype=np.int64)
        assert a.shape[0] == self.E
        _check(lib().bpp_step(ctypes.byref(self._b), _p(a), ctypes.byref(self._o), None))
        self._after_steps(1)
        return {k: v.copy() for k, v in self.out.items()} if copy else self.out


def rollout_uniform(env, seed, step0, nsteps):
    """nsteps lock-steps of OracleEnv `env` under the uniform-feasible policy; returns env.out (views)."""
    a = np.zeros(env.E, np.int64)
    if env.stream is not None:
        env.refill()
        side = getattr(env, "_side", None)          # bpp_side_create once per env: the overlapped schedule's stream + events
        if side is None:
            side = env._side = ctypes.c_void_p()
            _check(lib().bpp_side_create(ctypes.byref(side)))
        _check(lib().bpp_rollout_uniform_stream(ctypes.byref(env._b), ctypes.byref(env._o), _p(a), int(seed), int(step0),
                                                int(nsteps), ctypes.byref(env.stream), env.refill_every, side, None))
        return env.out, a
    _check(lib().bpp_rollout_uniform(ctypes.byref(env._b), ctypes.byref(env._o), _p(a), int(seed), int(step0),
                                     int(nsteps), None))
    return env.out, a


def epsilon_override(actions, M, seed, step, eps, env_id_base=0):
    """bpp_epsilon_override on a host int64 array, in place."""
    a = np.ascontiguousarray(actions, dtype=np.int64)
    _check(lib().bpp_epsilon_override(_p(a), a.shape[0], int(M), int(env_id_base), int(seed), int(step), int(round(eps * (1 << 24))), None))
    return a


def rollout_uniform_sets(env, seed, step0, nsteps, nsets, resume=False, actions=None, first_mask=None, eps=0.0):
    """bpp_rollout_uniform_sets on OracleEnv `env`: lock-step t writes output set t mod nsets; returns (list of the
    sets as dicts of arrays, actions = the draw for lock-step step0 + nsteps)."""
    E, A, M = env.E, env.A, env.M
    sets = [dict(obs=np.zeros((E, 4 * A), np.float32), mask=np.zeros((E, M), np.float32), reward=np.zeros(E, np.float32),
                 done=np.zeros(E, np.uint8), counter=np.zeros(E, np.int32), ratio=np.zeros(E, np.float64),
                 ep_ret=np.zeros(E, np.float64), ep_len=np.zeros(E, np.int32)) for _ in range(nsets)]
    outs = (StepOut * nsets)(*[StepOut(*[_p(d[k]).value for k in ("obs", "mask", "reward", "done", "counter", "ratio",
                                                                    "ep_ret", "ep_len")]) for d in sets])
    a = np.zeros(E, np.int64) if actions is None else actions
    fm = env.out["mask"] if first_mask is None else first_mask
    _check(lib().bpp_rollout_uniform_sets(ctypes.byref(env._b), outs, nsets, _p(fm), _p(a), int(seed), int(step0), int(nsteps),
                                          ctypes.c_int32(((1 if resume else 0) | (min(int(round(eps * (1 << 24))), (1 << 24) - 1) << 8)) & 0xFFFFFFFF).value, None))
    return sets, a


def mask_from_obs(obs, size, rotation, rule=RULE_UTILS):
    W, L, H = (int(v) for v in size)
    obs = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1, 4 * W * L)
    mask = np.zeros((obs.shape[0], W * L * (1 + int(bool(rotation)))), np.float32)
    _check(lib().bpp_mask_from_obs(_p(obs), _p(mask), obs.shape[0], W, L, H, int(bool(rotation)), int(rule), None))
    return mask


def mask_from_hmap(hmap, items, size, rotation, rule=RULE_SPACE):
    W, L, H = (int(v) for v in size)
    hmap = np.ascontiguousarray(hmap, dtype=np.int32).reshape(-1, W * L)
    items = np.ascontiguousarray(items, dtype=np.int32).reshape(-1, 3)
    assert items.shape[0] == hmap.shape[0]
    mask = np.zeros((hmap.shape[0], W * L * (1 + int(bool(rotation)))), np.float32)
    _check(lib().bpp_mask_from_hmap(_p(hmap), _p(items), _p(mask), hmap.shape[0], W, L, H, int(bool(rotation)),
                                    int(rule), None))
    return mask


def sample_feasible(mask, seed, step, env_id_base=0):
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    actions = np.zeros(mask.shape[0], np.int64)
    _check(lib().bpp_sample_feasible(_p(mask), _p(actions), mask.shape[0], mask.shape[1], int(env_id_base),
                                     int(seed), int(step), None))
    return actions


def episode_stats(done, ep_ret, ratio, ep_len, acc=None):
    done = np.ascontiguousarray(done, dtype=np.uint8)
    ep_ret = np.ascontiguousarray(ep_ret, dtype=np.float64)
    ratio = np.ascontiguousarray(ratio, dtype=np.float64)
    ep_len = np.ascontiguousarray(ep_len, dtype=np.int32)
    if acc is None:
        acc = np.zeros(4, np.float64)
    _check(lib().bpp_episode_stats(_p(done), _p(ep_ret), _p(ratio), _p(ep_len), done.shape[0], _p(acc), None))
    return acc


def masked_evaluate(logits, mask, action):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    action = np.ascontiguousarray(action, dtype=np.int64).reshape(-1)
    E, M = logits.shape
    lp, ent, bad = (np.zeros(E, np.float32) for _ in range(3))
    _check(lib().bpp_masked_evaluate(_p(logits), _p(mask), _p(action), _p(lp), _p(ent), _p(bad), E, M, None))
    return lp, ent, bad


def masked_evaluate_backward(logits, mask, action, g_lp, g_ent, g_bad):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    action = np.ascontiguousarray(action, dtype=np.int64).reshape(-1)
    E, M = logits.shape
    g = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32).reshape(-1), (E,))) for v in (g_lp, g_ent, g_bad)]
    grad = np.zeros((E, M), np.float32)
    _check(lib().bpp_masked_evaluate_backward(_p(logits), _p(mask), _p(action), _p(g[0]), _p(g[1]), _p(g[2]), _p(grad), E, M, None))
    return grad


def masked_act(logits, mask, seed, step, deterministic=False, env_id_base=0, counter=False):
    """counter=True: through bpp_masked_act_counter, (seed, step) handed over in memory"""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    a = np.zeros(logits.shape[0], np.int64)
    lp = np.zeros(logits.shape[0], np.float32)
    if counter:
        ss = np.array([int(seed), int(step)], np.uint64)
        _check(lib().bpp_masked_act_counter(_p(logits), _p(mask), _p(a), _p(lp), logits.shape[0], logits.shape[1], int(env_id_base), _p(ss),
                                            int(bool(deterministic)), None))
        return a, lp
    _check(lib().bpp_masked_act(_p(logits), _p(mask), _p(a), _p(lp), logits.shape[0], logits.shape[1], int(env_id_base),
                                int(seed), int(step), int(bool(deterministic)), None))
    return a, lp
