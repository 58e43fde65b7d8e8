"""BppVecEnv -- the reference's VecEnv boundary re-presented over the HIP step kernels.

Replaces `VecPyTorch(VecNormalize(ShmemVecEnv([PackingGame x N])))` as built by
`acktr.envs.make_vec_envs` (acktr/envs.py:77-118): same attributes (`num_envs`, `observation_space`,
`action_space`), same methods (`reset`, `step_async`, `step_wait`, `step`, `close`), same outputs
(obs float32 [E,4A] on the device; reward float32 [E,1] CPU tensor; done numpy bool [E]; infos: one
dict per bin with 'counter', 'ratio' and on terminal steps 'mask' and 'episode', bin3D.py:111,123-125,
baselines/bench/monitor.py:64-75).  All bins live on ONE GPU and are stepped by one fused kernel launch.

Tensor-native fast path (no host sync, everything stays on the device): `step_tensors(actions)`.
"""
import ctypes
import time

import numpy as np
import torch

from . import _lib
from .sequences import check_pool
from .spaces import Box, Discrete


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
STATE_FORMAT = 1      # state_dict()["format"]: 1 = 48-byte records whose last word is hmax, per-bin ep_acc rows
STREAM_LAYOUT = 2     # state_dict()["stream_layout"]: layout of the ring rows and generator records of a streaming env.
#                       2 = ring rows start with two look-ahead entries, item i at entry 2 + i; byte-output MT19937 records /
#                       16-byte counter records (ABI v12+).  A checkpoint without the key (layout 1 or older) is refused.


class StepTensors(object):
    """Device-resident result of one lock-step (views of the env's output buffers unless the env was built with
    fresh_outputs=True).  Either built from ready tensors (keyword arguments), or -- what the env does -- over ONE flat
    device allocation with a layout {name: (byte offset, dtype, shape)}: the views are then made on first use, so a
    step that nobody inspects beyond `obs` costs one allocation and one view."""
    FIELDS = ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len", "_small")
    __slots__ = FIELDS + ("_offs", "_stage", "_hot", "_flat", "_layout")

    def __init__(self, _flat=None, _layout=None, **kw):
        self._flat, self._layout = _flat, _layout
        self._offs, self._stage, self._hot = kw.pop("_offs", None), kw.pop("_stage", None), kw.pop("_hot", None)
        if _flat is None:
            for k in self.FIELDS:
                setattr(self, k, kw.get(k))

    def __getattr__(self, name):            # only reached for a slot that has not been filled yet: make the view
        lay = object.__getattribute__(self, "_layout") if name in StepTensors.FIELDS else None
        if lay is None:
            raise AttributeError(name)
        ent = lay.get(name)
        v = None
        if ent is not None:
            off, dtype, shape, nbytes = ent
            v = self._flat[off:off + nbytes].view(dtype).view(shape)
        setattr(self, name, v)
        return v

    def __getitem__(self, name):            # bufs["obs"] of older call sites
        return getattr(self, name)

    @property
    def masks(self):
        """float32 [E,1]: 0 where the episode just ended, else 1 -- `masks` of main.py:172 (rollouts.insert)."""
        return 1.0 - self.done.to(torch.float32).reshape(-1, 1)

    @property
    def bad_masks(self):
        """float32 [E,1] of ones: main.py:173 sets 0 only for 'bad_transition' infos (time-limit truncations), which this
        environment -- like the reference's PackingGame -- never produces."""
        return torch.ones((self.done.numel(), 1), dtype=torch.float32, device=self.done.device)

    def _to_host(self, nbytes, stream=None):
        """The first `nbytes` of the per-bin scalar block as a numpy byte array in page-locked host memory (a pageable
        destination runs at a fraction of the link's speed).  One native call: hipMemcpyAsync behind the step + stream
        synchronise (bpp_fetch_to_host).  The array is a view of a staging buffer that the env hands out again only
        once nobody holds a view of it any more, so whatever is built on it stays valid for as long as it is referenced."""
        if self._stage is not None and self._flat is not None:
            host = self._stage()
            if stream is None:
                stream = ctypes.c_void_p(torch.cuda.current_stream(self._flat.device).cuda_stream)
            rc = _lib.lib().bpp_fetch_to_host(self._flat.data_ptr() + self._layout["_small"][0], host.ctypes.data, int(nbytes), stream)
            if rc:
                _lib.check(rc)
            return host[:nbytes]
        return self._small[:nbytes].cpu().numpy()

    def host_reward_done(self, stream=None):
        """(reward float32 [E], done uint8 [E]) with ONE small device->host copy (5 bytes per bin): all the
        reference-shaped step() needs before anybody looks at `infos`.  `stream`: the stream the step was launched on
        (c_void_p; default: the device's current stream)."""
        if self._flat is None:
            E = self.done.numel()
            return self.reward.cpu().numpy().reshape(E), self.done.cpu().numpy()
        o = self._offs
        E = self._layout["done"][3]
        h = self._to_host(self._hot, stream)
        return h[o["reward"]:o["reward"] + 4 * E].view("<f4"), h[o["done"]:o["done"] + E]

    def host_scalars(self):
        """reward, done, counter, ratio, ep_ret, ep_len as numpy arrays (owned by the caller) with ONE device->host copy."""
        E = self.done.numel()
        if self._small is None:
            return dict(reward=self.reward.cpu().numpy().reshape(E), done=self.done.cpu().numpy(),
                        counter=self.counter.cpu().numpy(), ratio=self.ratio.cpu().numpy(),
                        ep_ret=self.ep_ret.cpu().numpy(), ep_len=self.ep_len.cpu().numpy())
        h = self._to_host(self._small.numel())
        o = self._offs
        return dict(reward=h[o["reward"]:o["reward"] + 4 * E].view("<f4"), done=h[o["done"]:o["done"] + E],
                    counter=h[o["counter"]:o["counter"] + 4 * E].view("<i4"),
                    ratio=h[o["ratio"]:o["ratio"] + 8 * E].view("<f8"),
                    ep_ret=h[o["ep_ret"]:o["ep_ret"] + 8 * E].view("<f8"),
                    ep_len=h[o["ep_len"]:o["ep_len"] + 4 * E].view("<i4"))


class LazyInfos(object):
    """`infos` of a VecEnv step: behaves like the reference's tuple of E dicts, but nothing is fetched or built until
    somebody looks.  The ACKTR loop only reads `'episode' in infos[i]`, `infos[i]['episode']['r']`,
    `infos[i]['ratio']` and `'bad_transition' in info` (main.py:159-162,173).

    What moves when: `step()` itself copies reward + done (5 bytes per bin).  The first access to the info of a
    FINISHED bin gathers (ep_ret, ep_len, ratio, counter) of the finished bins on the device and copies those few
    rows; the first access to a bin that is still running copies the whole counter / ratio arrays (12 bytes per bin).
    The device tensors behind this are the step's own outputs: with fresh_outputs=True (make_vec_envs' setting, the
    reference's semantics) they stay valid for as long as the infos object lives; with shared output buffers
    (fresh_outputs=False) they are overwritten by the next step, and a late first access raises instead of returning
    another step's numbers."""

    def __init__(self, env, res, t_now, done=None, serial=None, fin=None):
        self._env = env
        self._res = res
        self._t = t_now
        self._done = None if done is None else (done if isinstance(done, np.ndarray) and done.dtype == np.bool_ else np.asarray(done).astype(bool))
        self._serial = serial
        self._fin = fin        # (bins, [ep_ret, ratio], [ep_len, counter]) of the finished bins; handed in by an env with eager_infos
        self._live = None      # (counter, ratio) arrays of all bins
        self._dicts = None

    def _check_fresh(self):
        env = self._env
        if self._serial is not None and not getattr(env, "fresh_outputs", True) and getattr(env, "_serial", self._serial) != self._serial:
            raise RuntimeError("infos of an earlier step read after the env stepped again: its output buffers are shared "
                               "(fresh_outputs=False); read infos before the next step or build the env with fresh_outputs=True")

    def _done_mask(self):
        if self._done is None:
            self._check_fresh()
            self._done = self._res.done.cpu().numpy().astype(bool)
        return self._done

    def _finished(self):
        if self._fin is None:
            self._check_fresh()
            r = self._res
            env = self._env
            if getattr(r, "_flat", None) is not None and hasattr(env, "_gather_finished"):
                # ONE native call (bpp_gather_finished): a launch compacts the finished bins' (r, ratio, l, counter, bin) rows on
                # the device in bin order, one copy brings exactly those rows over
                n = int(np.count_nonzero(self._done_mask()))
                if n:
                    bins, ret, ratio, ln, cnt = env._gather_finished(r, n)
                else:
                    bins, ret, ratio, ln, cnt = (np.zeros(0, "<i4"), np.zeros(0), np.zeros(0), np.zeros(0, "<i4"), np.zeros(0, "<i4"))
                self._fin = (bins, (ret, ratio), (ln, cnt))
                return self._fin
            idx = np.flatnonzero(self._done_mask())
            if idx.size and torch.is_tensor(r.ep_ret):
                it = torch.from_numpy(idx).to(r.ep_ret.device)
                f64 = torch.stack([r.ep_ret[it], r.ratio[it]]).cpu().numpy()
                i32 = torch.stack([r.ep_len[it], r.counter[it]]).cpu().numpy()
            else:       # plain arrays (host tests)
                f64 = np.stack([r.ep_ret.cpu().numpy()[idx], r.ratio.cpu().numpy()[idx]])
                i32 = np.stack([r.ep_len.cpu().numpy()[idx], r.counter.cpu().numpy()[idx]])
            self._fin = (idx, f64, i32)
        return self._fin

    def episodes(self):
        """The episodes that finished in this step as arrays -- what main.py:159-162 collects one dict at a time:
        {'bins': int [n], 'r': float64 [n] (rounded like Monitor's), 'l': int32 [n], 'ratio': float64 [n], 'counter': int32 [n]}."""
        idx, f64, i32 = self._finished()
        return {"bins": idx, "r": np.round(f64[0], 6), "l": i32[0], "ratio": f64[1], "counter": i32[1]}

    def _running(self):
        if self._live is None:
            self._check_fresh()
            r = self._res
            if getattr(r, "_flat", None) is not None and hasattr(r, "host_scalars"):
                h = r.host_scalars()          # ONE device-to-host copy of the 29-byte-per-bin scalar block + one synchronisation
                self._live = (h["counter"].copy(), h["ratio"].copy())
            else:
                self._live = (r.counter.cpu().numpy().copy(), r.ratio.cpu().numpy().copy())
        return self._live

    def __len__(self):
        return self._env.num_envs

    def _make(self, i):
        if self._done_mask()[i]:
            idx, f64, i32 = self._finished()
            k = int(np.searchsorted(idx, i))
            return {"counter": int(i32[1][k]), "ratio": np.float64(f64[1][k]),
                    "mask": np.ones(shape=self._env.act_len),                  # bin3D.py:111
                    "episode": {"r": round(float(f64[0][k]), 6), "l": int(i32[0][k]),
                                "t": round(self._t - self._env._tstart, 6)}}   # bench/monitor.py:64
        counter, ratio = self._running()
        return {"counter": int(counter[i]), "ratio": np.float64(ratio[i])}

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if self._dicts is None:
            self._dicts = [None] * len(self)
        if self._dicts[i] is None:
            self._dicts[i] = self._make(i)
        return self._dicts[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def done_indices(self):
        """Local numbers of the bins that finished in this step, ascending, as a fresh intp array (whatever was accessed before)."""
        if self._fin is not None:          # (eager_infos: the compaction's bin list is the answer, ascending)
            return self._fin[0].astype(np.intp)
        return np.flatnonzero(self._done_mask())


class MonitorCsv(object):
    """`<log_dir>/<rank>.monitor.csv` in the format of baselines/bench/monitor.py:99-119 (ResultsWriter): a `# {json}` header
    line with `t_start` and `env_id`, the csv header `r,l,t` + extra keys, one row per finished episode with Monitor's
    values (`r` = round(sum of rewards, 6), `l` = steps incl. the failing one, `t` = seconds since t_start, :58-66).  The
    reference writes one file PER WORKER (acktr/envs.py:54-58: os.path.join(log_dir, str(rank))); here all bins of a shard
    live in one process, so there is ONE file per shard -- named after the shard's rank -- with the extra column `bin`
    (global bin id; Monitor's own extra-key mechanism, info_keywords), and the reference's `load_results(log_dir)` reads it.
    Rows of one lock-step are written in bin order."""
    EXT = "monitor.csv"

    def __init__(self, log_dir, rank=0, env_id="Bpp-v0", env_id_base=0, t_start=None):
        import json
        import os
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "%d.%s" % (int(rank), self.EXT))
        self.t_start = time.time() if t_start is None else float(t_start)
        self.env_id_base = int(env_id_base)
        self.rows = 0
        self.f = open(self.path, "wt")
        self.f.write("# %s \n" % json.dumps({"t_start": self.t_start, "env_id": env_id}))     # monitor.py:109-111
        self.f.write("r,l,t,bin\r\n")                                                        # csv.DictWriter.writeheader
        self.f.flush()

    def write(self, episodes, t_now=None):
        """episodes: LazyInfos.episodes() of one lock-step (arrays bins / r / l)."""
        n = len(episodes["bins"])
        if n and self.f is not None:
            t = repr(round((time.time() if t_now is None else t_now) - self.t_start, 6))
            base = self.env_id_base
            self.f.write("".join("%r,%d,%s,%d\r\n" % (r, l, t, base + b)
                                 for r, l, b in zip(episodes["r"].tolist(), episodes["l"].tolist(), episodes["bins"].tolist())))
            self.f.flush()
            self.rows += n

    def close(self):
        if self.f is not None:
            self.f.close()
            self.f = None


def copy_bin_records(hmap, state, src, dst, ring=None, mt=None, gen_next=None, depth=None):
    """Bins `dst` become copies of bins `src` (int64 index tensors on the tensors' device): byte heightmap [E][A] and
    the 48-byte record as int32 [E][12].  Streaming supply (ring [depth * E][T][4], generator records mt [E][*], gen_next
    [E]): the copy continues the SOURCE's item stream, so its ring column, generator record and progress are copied
    too -- and, because a ring row number encodes the bin (row = (episode mod depth) * E + bin, advanced by += E in the
    step kernels), the copied `seq` is rebased from the source's column to the destination's; without that the copy
    would go on reading the source's column, which the source's refills overwrite."""
    hmap[dst] = hmap[src]
    state[dst] = state[src]
    if ring is not None:
        E = state.shape[0]
        cols = ring.view(int(depth), E, -1)
        cols[:, dst] = cols[:, src]
        mt[dst] = mt[src]
        gen_next[dst] = gen_next[src]
        state[dst, 7] = state[src, 7] - src.to(state.dtype) + dst.to(state.dtype)     # bpp_env_state.seq


class BppVecEnv(object):
    """E independent bins stepped in lock-step on one MI355X.

    pool:           uint8 [P][T][4] item sequences (sequences.py); episode k of global bin g plays row
                    (g + k * env_id_total) mod P.
    env_id_base/total: this shard's first global bin id / bins in the whole job (multi-GPU sharding).
    mask_rule:      'utils' (acktr/utils.py check_box -- what the training loop consumes) or 'space'
                    (Space.check_box, i.e. PackingGame.get_possible_position).
    compute_mask:   also produce the feasibility mask of every returned observation (`location_masks`).
    eager_infos:    step() also enqueues the compaction of the finished bins' infos behind the step kernel (one more launch,
                    ~10 us of device time at 65 536 bins), so that reading `infos.episodes()` / a finished bin's dict costs no
                    launch and no second synchronisation -- for loops that look at the finished episodes EVERY step, as
                    main.py:159-162 does (make_vec_envs sets it); False: the gather happens when somebody looks.
    fresh_outputs:  allocate new output tensors every step (reference semantics: results of earlier
                    steps stay valid); False = reuse one set of buffers, results are valid until the
                    next step/reset call.
    stream:         instead of `pool`: dict(bound=(lo, hi), seed=s, depth=D, refill_every=R, rng="mt19937") -- an endless
                    CUT-2 supply generated on the device (include/bpp_abi.h: bpp_stream).  Every bin owns an exact
                    random.Random(seed + global bin id); its k-th episode plays the k-th sequence that stream
                    yields through the reference's MDlayerBoxCreator, so no sequence is ever replayed.
                    rng="counter": the same cutting algorithm on a counter-based generator (distribution parity,
                    SURVEY 8f2's bar; no per-bin state, no regeneration kernel -- the fast supply).  The ring
                    is refilled every R <= D - 3 lock-steps, R <= D - 4 with the row cache (default D = 8: R = 4 for
                    the 10x10 / 20x20 bins, whose tile step kernel keeps a row cache by default -- cache=True / False
                    overrides --, R = 5 for every other geometry); `rollout_uniform` runs the refills beside the
                    lock-steps when D >= 2 R + 3 (+ 4 with the cache; e.g. D = 32, R = 14).  Costs 6.6 KB of
                    generator state per bin (16 bytes with rng="counter") plus D rows of W*L*H / lo^3 + 3 entries.
    """

    def __init__(self, num_envs, container_size=(10, 10, 10), enable_rotation=False, pool=None, device="cuda",
                 env_id_base=0, env_id_total=None, mask_rule="utils", compute_mask=True, fresh_outputs=False, stream=None, eager_infos=False):
        if not torch.cuda.is_available():
            raise RuntimeError("BppVecEnv needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BppVecEnv runs on a HIP device only, got %r" % (device,))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.lib()
        self.num_envs = self.E = int(num_envs)
        self.bin_size = tuple(int(v) for v in container_size)
        self.W, self.L, self.H = self.bin_size
        self.area = self.A = self.W * self.L
        self.can_rotate = bool(enable_rotation)
        self.act_len = self.M = self.A * (1 + int(self.can_rotate))
        self.obs_len = 4 * self.A
        max_area, max_dim = _lib.limits()
        if self.A > max_area or max(self.bin_size) > max_dim:
            raise ValueError("bin %r exceeds kernel limits (W*L <= %d, sides <= %d)" % (self.bin_size, max_area, max_dim))
        # bin3D.py:38-41
        self.action_space = Discrete(self.act_len)
        self.observation_space = Box(low=0.0, high=self.H, shape=(self.obs_len,))
        if (pool is None) == (stream is None):
            raise ValueError("give either an item-sequence pool (sequences.cut2_pool / cut1_pool / rs_pool) or stream=dict(...)")
        self.pool_host = check_pool(pool, self.bin_size) if pool is not None else None
        self.mask_rule = {"utils": _lib.RULE_UTILS, "space": _lib.RULE_SPACE}[mask_rule]
        self.compute_mask = bool(compute_mask)
        self.fresh_outputs = bool(fresh_outputs)
        self.eager_infos = bool(eager_infos)
        self.env_id_base = int(env_id_base)
        self.env_id_total = int(env_id_total) if env_id_total is not None else self.env_id_base + self.E
        dev = self.device
        self.stream_spec = None
        with torch.cuda.device(dev):
            if stream is None:
                self.pool = torch.from_numpy(self.pool_host).to(dev)
                pool_rows, pool_len, pool_mode = self.pool_host.shape[0], self.pool_host.shape[1], _lib.POOL_STATIC
            else:
                lo, hi = (int(v) for v in stream.get("bound", (2, 5)))
                depth = int(stream.get("depth", 8))
                # bpp_batch.seq_cache: the row cache (include/bpp_abi.h) -- no step workgroup waits for a read of the ring;
                # its lines refer to the row after next, which costs one row of look-ahead.  Default: wherever the refill
                # schedule leaves that row
                if stream.get("cache") is not None:
                    cache = bool(stream["cache"])
                else:       # only the tile step kernel keeps the cache: any other geometry would pay a row of look-ahead for nothing
                    cache = depth >= 5 and depth - int(stream.get("refill_every", 1)) >= 4 and \
                        int(stream.get("pool_len", self.W * self.L * self.H // lo ** 3 + 3)) < 8192 and \
                        _lib.launch_info(self.E, self.bin_size, self.can_rotate)["kernel"] == 2      # BPP_KERNEL_TILE
                behind = 4 if cache else 3
                if depth < behind + 1:
                    raise ValueError("stream depth must be >= %d" % (behind + 1))
                # entries per ring row: two look-ahead entries (include/bpp_abi.h), the longest possible sequence, the terminator
                pool_len = int(stream.get("pool_len", self.W * self.L * self.H // lo ** 3 + 3))
                self.refill_every = int(stream.get("refill_every", depth - behind))
                if not 1 <= self.refill_every <= depth - behind:
                    raise ValueError("refill_every must be in 1 .. depth - %d" % behind)
                self.pool = torch.zeros((depth * self.E, pool_len, 4), dtype=torch.uint8, device=dev)   # the ring
                sizes = (ctypes.c_int64 * 2)()      # the two opaque buffers of a bpp_stream: generator records, scratch
                rng = {"mt19937": _lib.STREAM_RNG_MT19937, "counter": _lib.STREAM_RNG_COUNTER}[stream.get("rng", "mt19937")]
                probe = _lib.Stream(self.E, depth, pool_len, self.W, self.L, self.H, lo, hi, 0, 0, None, None, None, None, None, None, rng, 0)
                _lib.check(self.lib.bpp_stream_sizes(ctypes.byref(probe), sizes))
                self._mt = torch.zeros((self.E, int(sizes[0]) // self.E), dtype=torch.int32, device=dev)
                self._work = torch.zeros(((int(sizes[1]) + 15) // 16, 4), dtype=torch.int32, device=dev)
                self.gen_next = torch.zeros((self.E,), dtype=torch.int32, device=dev)
                self.stream_overflow = torch.zeros((1,), dtype=torch.int32, device=dev)
                self._seq_cache = torch.zeros((self.E * _lib.SEQ_CACHE_BYTES_PER_BIN,), dtype=torch.uint8, device=dev) if cache else None
                pool_rows, pool_mode = depth * self.E, _lib.POOL_RING
                self.stream_spec = dict(bound=(lo, hi), seed=int(stream.get("seed", 0)), depth=depth, pool_len=pool_len,
                                        refill_every=self.refill_every, rng=stream.get("rng", "mt19937"), cache=cache)
            self.hmap = torch.zeros((self.E, self.A), dtype=torch.uint8, device=dev)  # Space.plain as bytes
            self.state = torch.zeros((self.E, 12), dtype=torch.int32, device=dev)  # bpp_env_state[E], 48 B each
            # episode statistics kept inside the step kernel, one row per bin: [return sum, final-ratio sum, length
            # sum, episodes] -- plain read-modify-write by the bin's own lane, reduced in a fixed order on demand
            self.ep_acc = torch.zeros((self.E, 4), dtype=torch.float64, device=dev)
        self._batch = _lib.Batch(self.E, self.W, self.L, self.H, int(self.can_rotate), self.mask_rule,
                                 pool_rows, pool_len, self.env_id_base, self.env_id_total,
                                 self.pool.data_ptr(), self.hmap.data_ptr(), self.state.data_ptr(),
                                 self.ep_acc.data_ptr(), pool_mode, 0,
                                 self._seq_cache.data_ptr() if self.stream_spec is not None and self._seq_cache is not None else None)
        self._batch_ref = ctypes.byref(self._batch)
        self._stream = None
        self._since_refill = 0
        if self.stream_spec is not None:
            sp = self.stream_spec
            self._stream = _lib.Stream(self.E, sp["depth"], sp["pool_len"], self.W, self.L, self.H, sp["bound"][0], sp["bound"][1],
                                       self.env_id_base, sp["seed"], self.pool.data_ptr(), self._mt.data_ptr(),
                                       self._work.data_ptr(), self.gen_next.data_ptr(), self.state.data_ptr(),
                                       self.stream_overflow.data_ptr(),
                                       {"mt19937": _lib.STREAM_RNG_MT19937, "counter": _lib.STREAM_RNG_COUNTER}[sp["rng"]], 0)
            with torch.cuda.device(dev):
                _lib.check(self.lib.bpp_stream_init(ctypes.byref(self._stream), self._stream_ptr()))
            self.refill()
        self._bufs = None
        self._out = None
        self._res = None
        self._first_reset = True
        self._out_pool = []        # fresh_outputs: output sets that may be handed out again once unreferenced
        self._last_stream = None
        self._serial = 0           # lock-steps issued (LazyInfos: which step the shared output buffers belong to)
        self._pending = None
        self._mark = 0             # serial number of the last bpp_mark (never 0: a cleared completion word)
        self._tstart = time.time()
        self.monitor = None        # MonitorCsv (make_vec_envs with a log_dir): step_wait() appends the finished episodes' rows
        self._side = None          # bpp_side (rollout_uniform in streaming mode): created on first use
        self.closed = False
        from . import masks as _masks
        _masks.register_env(self)      # the per-row mask helpers may hand back this env's own mask rows (masks.ROW_CACHE)

    MAX_STAGING = 16     # page-locked host buffers (29 B per bin each) handed out at the same time, at most
    spin_wait = True     # step_wait() spins on the step's completion word (bpp_mark / bpp_wait_mark); False: hipStreamSynchronize

    # ------------------------------------------------------------------ buffers
    def _layout(self):
        """Byte layout of one set of output buffers inside a single allocation (every region 256-byte aligned): obs,
        mask, then the six small per-bin outputs as one block whose first part -- reward and done, 5 bytes per bin -- is
        all the reference-shaped step_wait() copies to the host."""
        lay = getattr(self, "_lay", None)
        if lay is None:
            E, total, regions = self.E, 0, {}

            def add(name, dtype, shape, width):
                nonlocal total
                n = 1
                for d in shape:
                    n *= d
                regions[name] = (total, dtype, shape, n * width)
                return n * width

            total += (add("obs", torch.float32, (E, self.obs_len), 4) + 255) // 256 * 256
            if self.compute_mask:
                total += (add("mask", torch.float32, (E, self.M), 4) + 255) // 256 * 256
            small0, offs, hot = total, {}, 0
            for name, dtype, shape, width in (("reward", torch.float32, (E, 1), 4), ("done", torch.uint8, (E,), 1),
                                              ("ratio", torch.float64, (E,), 8), ("ep_ret", torch.float64, (E,), 8),
                                              ("counter", torch.int32, (E,), 4), ("ep_len", torch.int32, (E,), 4)):
                offs[name] = total - small0
                total += (add(name, dtype, shape, width) + 7) // 8 * 8
                if name == "done":
                    hot = total - small0
            regions["_small"] = (small0, torch.uint8, (total - small0,), total - small0)
            lay = self._lay = (regions, total, offs, hot)
        return lay

    def _alloc(self):
        """One set of output buffers: (StepTensors over one flat allocation, bpp_step_out with its pointers).  With
        fresh_outputs=True every step gets its own set; sets whose storage nobody references any more (the StepTensors
        AND every view made from it are gone -- torch's storage use count says so) are handed out again instead of going
        through the allocator: same lifetime rule as the caching allocator's (the next writer is a kernel on the caller's
        stream, ordered behind whatever was enqueued there before)."""
        regions, total, offs, hot = self._layout()
        use_count, idle = self._storage_use_count() if self.fresh_outputs else (None, 0)
        flat = out = None
        if use_count is not None:
            pool = self._out_pool
            for k, (f, o) in enumerate(pool):
                if use_count(f.untyped_storage()._cdata) <= idle:   # the pool's tensor + the wrapper just made for the query
                    flat, out = f, o
                    pool.append(pool.pop(k))
                    break
        if flat is None:
            flat = torch.empty((total,), dtype=torch.uint8, device=self.device)
            base = flat.data_ptr()
            ptr_offs = getattr(self, "_ptr_offs", None)
            if ptr_offs is None:
                ptr_offs = self._ptr_offs = [(regions[k][0] if k in regions else None)
                                             for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len")]
            out = _lib.StepOut(*[(base + o if o is not None else None) for o in ptr_offs])
            if use_count is not None and len(self._out_pool) < 8:
                self._out_pool.append((flat, out))
        if use_count is not None:
            flat = flat.detach()        # the result's own handle on the storage: the pool's stays the only one when it is gone
        res = StepTensors(_flat=flat, _layout=regions, _offs=offs, _stage=self._staging, _hot=hot)
        return res, out

    def _storage_use_count(self):
        """(torch's storage use-count query, its reading for a storage only ONE tensor refers to) -- or (None, 0) when
        this torch build has no such query or it does not behave as expected.  Calibrated once per env: the reading of a
        fresh tensor is the idle value, and it must rise by one when a second handle (`detach()`) exists; otherwise
        output sets are never handed out again (plain allocation every step).  Holders of raw pointers or of a bare
        `untyped_storage()` are NOT counted by torch: they do not keep an output set alive."""
        cal = getattr(self, "_use_count_cal", None)
        if cal is None:
            fn, idle = getattr(torch._C, "_storage_Use_Count", None), 0
            try:
                t = torch.empty((16,), dtype=torch.uint8, device=self.device)
                idle = fn(t.untyped_storage()._cdata)
                d = t.detach()
                if fn(t.untyped_storage()._cdata) != idle + 1:
                    fn = None
                del d
                if fn is not None and fn(t.untyped_storage()._cdata) != idle:
                    fn = None
            except Exception:  # noqa: BLE001
                fn = None
            cal = self._use_count_cal = (fn, idle)
        return cal

    def _staging(self, mapped=False):
        """A page-locked host buffer (numpy uint8 view) for the per-bin scalars of a step that nobody else references:
        buffers handed out earlier come back into use only when every view of them (the CPU reward tensor, `done`) has
        been dropped -- checked by reference count --, otherwise a new one is pinned.  The reference loop copies reward
        and done into its rollout storage at once, so two or three buffers circulate."""
        import sys
        pool = getattr(self, "_stage_pool", None)
        n = self._stage_bytes()
        if pool is None or (pool and pool[0][1].size != n):
            pool = self._stage_pool = []
        for k, (t, a) in enumerate(pool):
            if sys.getrefcount(a) <= 3:          # the pool's tuple, the loop variable, getrefcount's argument
                pool.append(pool.pop(k))
                return a
        if len(pool) >= self.MAX_STAGING:       # a caller keeps every step's reward / done views alive: do not pin without bound
            if not getattr(self, "_staging_warned", False):
                import warnings
                warnings.warn("%d page-locked step buffers are all still referenced; further steps use pageable host memory "
                              "(slower) -- copy reward / done out of the step's views instead of keeping them" % len(pool), RuntimeWarning)
                self._staging_warned = True
            return None if mapped else np.empty((n,), dtype=np.uint8)    # (a kernel cannot write into pageable memory)
        t = torch.empty((n,), dtype=torch.uint8).pin_memory()
        a = t.numpy()
        a[n - 8:] = 0                           # the completion word (bpp_mark): no step's serial number yet
        pool.append((t, a))
        return a

    def _fin_offset(self):
        """Byte offset of the eager gather's area (bpp_gather_finished, BPP_GATHER_ENQUEUE_ONLY: 32-byte header + five arrays laid
        out for E entries) inside a staging buffer: behind the per-bin scalar block, 8-byte aligned."""
        return (self._layout()[0]["_small"][3] + 7) // 8 * 8

    def _stage_bytes(self):
        """A staging buffer = the per-bin scalar block (29 B per bin; the kernel mirrors its first 5: reward, done) and, with
        eager_infos, the compacted records of the finished bins (28 B per bin of room; a step fills the first ~11 % of each array)."""
        return self._mark_offset() + 8

    def _mark_offset(self):
        """Byte offset of the step's completion word (bpp_mark / bpp_wait_mark) in a staging buffer: behind everything else."""
        return self._fin_offset() + ((32 + 28 * self.E + 4 + 7) // 8 * 8 if self.eager_infos else 0)

    def _gather_finished(self, res, n):
        """(bins int32 [n], ep_ret f64 [n], ratio f64 [n], ep_len int32 [n], counter int32 [n]) of the `n` finished bins of
        step result `res`, ascending bin order, through bpp_gather_finished: one compaction launch, ONE device-to-host copy
        of 32 + 28 n bytes laid out as those arrays, stream synchronise.  The device and page-locked staging areas are
        allocated on first use; what is returned are views of ONE private copy of the transferred bytes."""
        st = getattr(self, "_fin_stage", None)
        if st is None:
            nb = (32 + 28 * self.E + 4 + 7) // 8 * 8
            t = torch.empty((nb,), dtype=torch.uint8).pin_memory()
            st = self._fin_stage = (t, t.numpy())
        _, host = st
        base, lay = res._flat.data_ptr(), res._layout
        self._on_device()
        _lib.check(self.lib.bpp_gather_finished(base + lay["done"][0], base + lay["ep_ret"][0], base + lay["ratio"][0],
                                                base + lay["ep_len"][0], base + lay["counter"][0], self.E, None,
                                                host.ctypes.data, int(n), self._stream_ptr()))
        buf = host[32:32 + 28 * n].copy()
        return (buf[24 * n:28 * n].view("<i4"), buf[:8 * n].view("<f8"), buf[8 * n:16 * n].view("<f8"),
                buf[16 * n:20 * n].view("<i4"), buf[20 * n:24 * n].view("<i4"))

    def _buffers(self):
        if self.fresh_outputs or self._bufs is None:
            self._bufs, self._out = self._alloc()
        return self._bufs, self._out

    @staticmethod
    def _plain(out):
        """Clear the per-call fields of a bpp_step_out (step_tensors sets them in place): no fused draw, no host mirrors."""
        out.next_action = out.host_reward = out.host_done = None
        return out

    def _stream_ptr(self):
        """The calling thread's current HIP stream on the env's device as a c_void_p (raw handle straight from torch's
        stream table -- a third of the cost of building a torch.cuda.Stream object per step)."""
        raw = _RAW_STREAM
        if raw is not None:
            return ctypes.c_void_p(raw(self.device.index))
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def refill(self):
        """Streaming supply: cut new sequences for the episodes the bins have consumed (four kernels, no host sync)."""
        if self._stream is None:
            raise RuntimeError("refill() needs a streaming env (stream=dict(...))")
        self._on_device()
        _lib.check(self.lib.bpp_stream_refill(ctypes.byref(self._stream), self._stream_ptr()))
        self._since_refill = 0

    def _reset_seq_cache(self):
        """`state` or the ring were written behind the library's back: zero the row cache (the kernels then read the ring
        until their refresh requests have been served again)."""
        if self._stream is not None and self._seq_cache is not None:
            self._seq_cache.zero_()

    def _stepped(self, n=1):
        if self._stream is not None:
            self._since_refill += n
            if self._since_refill >= self.refill_every:
                self.refill()

    def _on_device(self):
        """HIP launches target the calling thread's current device: make sure it is ours (cheap check,
        the context switch only happens when it is not)."""
        if torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)

    @property
    def location_masks(self):
        """float32 [E][M] feasibility mask of the current observations (None before the first reset / without compute_mask)."""
        return self._res.mask if self._res is not None else None

    # ------------------------------------------------------------------ VecEnv interface
    def reset(self):
        """All bins start a fresh episode (next sequence of their stride); returns obs [E,4A] float32."""
        self._on_device()
        bufs, out = self._buffers()
        self._res = bufs
        mode = _lib.RESET_INIT if self._first_reset else _lib.RESET_ADVANCE
        if self._stream is not None and not self._first_reset:
            self.refill()              # RESET_ADVANCE moves every bin on by one episode
        _lib.check(self.lib.bpp_reset(self._batch_ref, mode, ctypes.byref(self._plain(out)), self._stream_ptr()))
        self._serial += 1
        self._stepped()
        self._first_reset = False
        self._tstart = time.time()
        return bufs["obs"]

    def step_tensors(self, actions, sample=None, _host=None, _dropin=None):
        """Enqueue one lock-step; returns device tensors, never synchronises.  actions: int64 [E] or [E,1].
        sample=(seed, step, out): additionally draw, inside the step kernel, the uniform-feasible action
        for the NEW observation into int64 tensor `out` [E] (== sample_feasible(seed, step) on the new mask;
        `out` may be the action tensor itself).  (_host: page-locked numpy byte buffer the kernel mirrors reward and
        done into; _dropin = (fin_host pointer or None, completion-word pointer or None, value): bpp_step_dropin instead of
        bpp_step -- step_async's business.)"""
        if self._first_reset:
            raise RuntimeError("call reset() before step()")
        a = actions
        if not torch.is_tensor(a):
            a = torch.as_tensor(np.asarray(a))
        if a.numel() != self.E:
            raise ValueError("expected %d actions, got %d" % (self.E, a.numel()))
        # ([E] and [E,1] contiguous tensors share their memory layout: no reshape needed)
        if a.device != self.device or a.dtype != torch.int64 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.int64).contiguous()
        self._on_device()
        if self.fresh_outputs or self._bufs is None:
            self._bufs, self._out = self._alloc()
            self._res = self._bufs
        out = self._out          # this output set's own bpp_step_out: the per-call fields are set in place (no struct copy)
        if sample is not None:
            seed, step, nxt = sample
            if nxt.device != self.device or nxt.dtype != torch.int64 or nxt.numel() != self.E or not nxt.is_contiguous():
                raise ValueError("sample out tensor must be a contiguous int64 [E] tensor on the env's device")
            out.next_action, out.sample_seed, out.sample_step = nxt.data_ptr(), int(seed), int(step)
        else:
            out.next_action = None
        if _host is not None:
            offs, base = self._layout()[2], _host.ctypes.data
            out.host_reward, out.host_done = base + offs["reward"], base + offs["done"]
        else:
            out.host_reward = out.host_done = None
        self._last_stream = sp = self._stream_ptr()
        if _dropin is None:
            rc = self.lib.bpp_step(self._batch_ref, a.data_ptr(), ctypes.byref(out), sp)
        else:       # step + eager gather of the finished bins + completion word: ONE trip through ctypes (ABI v16)
            rc = self.lib.bpp_step_dropin(self._batch_ref, a.data_ptr(), ctypes.byref(out), _dropin[0], _dropin[1], _dropin[2], sp)
        if rc:
            _lib.check(rc)
        self._serial += 1
        if self._stream is not None:
            self._stepped()
        return self._res

    def rollout_uniform(self, seed, step0, nsteps, actions=None):
        """`nsteps` lock-steps under the uniform-random-feasible policy, enqueued by ONE native call
        (bpp_rollout_uniform): no per-step Python.  Returns the StepTensors of the last step."""
        if self._first_reset:
            raise RuntimeError("call reset() before rollout_uniform()")
        if not self.compute_mask:
            raise RuntimeError("rollout_uniform needs compute_mask=True")
        if self.fresh_outputs:
            raise RuntimeError("rollout_uniform reuses one set of output buffers (fresh_outputs=False)")
        if actions is None:
            actions = torch.empty((self.E,), dtype=torch.int64, device=self.device)
        self._on_device()
        self._serial += int(nsteps)
        if self._stream is not None:
            self.refill()
            if self._side is None:      # the overlapped schedule's side stream + events: this env's own (bpp_side_create), released in close()
                self._side = ctypes.c_void_p()
                _lib.check(self.lib.bpp_side_create(ctypes.byref(self._side)))
            _lib.check(self.lib.bpp_rollout_uniform_stream(self._batch_ref, ctypes.byref(self._plain(self._out)), actions.data_ptr(), int(seed),
                                                           int(step0), int(nsteps), ctypes.byref(self._stream),
                                                           self.refill_every, self._side, self._stream_ptr()))
            return self._res
        _lib.check(self.lib.bpp_rollout_uniform(self._batch_ref, ctypes.byref(self._plain(self._out)), actions.data_ptr(), int(seed),
                                                int(step0), int(nsteps), self._stream_ptr()))
        return self._res

    def output_sets(self, n):
        """`n` complete sets of output buffers for rollout_uniform_sets (lock-step t writes set t mod n)."""
        return [self._alloc() for _ in range(int(n))]

    def rollout_uniform_sets(self, seed, step0, nsteps, actions, sets=None, resume=False, eps=0.0):
        """bpp_rollout_uniform_sets: like rollout_uniform, but lock-step t writes its outputs into sets[t mod n]
        (output_sets(n); default: the env's own single set) and the LAST lock-step also draws the next action, so that
        a following call with resume=True enqueues nothing but its `nsteps` step-kernel launches.  Finite pools only.
        eps > 0: SURVEY 8d's failure-path variant -- every draw is followed by bpp_epsilon_override (with probability eps
        the action becomes a uniform draw over ALL entries; one more tiny launch per lock-step).
        Returns the StepTensors of the last lock-step."""
        if self._first_reset:
            raise RuntimeError("call reset() before rollout_uniform_sets()")
        if self._stream is not None:
            raise RuntimeError("rollout_uniform_sets drives finite pools; streaming envs use rollout_uniform")
        if not self.compute_mask or self.fresh_outputs:
            raise RuntimeError("rollout_uniform_sets needs compute_mask=True and fresh_outputs=False")
        if actions.device != self.device or actions.dtype != torch.int64 or actions.numel() != self.E or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous int64 [E] tensor on the env's device")
        if sets is None:
            sets = [(self._bufs, self._out)]
        n = len(sets)
        outs = (_lib.StepOut * n)(*[self._plain(o) for _, o in sets])
        first = self.location_masks
        self._on_device()
        self._serial += int(nsteps)
        _lib.check(self.lib.bpp_rollout_uniform_sets(self._batch_ref, outs, n, first.data_ptr() if first is not None else None,
                                                     actions.data_ptr(), int(seed), int(step0), int(nsteps),
                                                     (_lib.ROLLOUT_CONTINUE if resume else 0) | (_lib.rollout_eps_flags(eps) if eps else 0),
                                                     self._stream_ptr()))
        if nsteps > 0:
            self._bufs, self._out = sets[(int(nsteps) - 1) % n]
            self._res = self._bufs
        return self._res

    def _stage_offsets(self):
        """(reward offset, done offset, eager gather's offset or None, completion word's offset) inside a staging buffer."""
        key = (self.E, bool(self.eager_infos))
        so = getattr(self, "_stage_offs", None)
        if so is None or so[0] != key:
            offs = self._layout()[2]
            so = self._stage_offs = (key, offs["reward"], offs["done"], self._fin_offset() if self.eager_infos else None, self._mark_offset())
        return so

    def step_async(self, actions, sample=None):
        """The reference-shaped path: the step kernel also writes reward and done (5 bytes per bin) straight into a
        page-locked host buffer (bpp_step_out.host_reward / host_done), so step_wait() copies nothing.  ONE native call
        (bpp_step_dropin) enqueues the step, with eager_infos the compaction of the finished bins' (r, ratio, l, counter, bin)
        into the same buffer, and the step's completion word behind both.
        sample=(seed, step, out): as in step_tensors -- also draw a uniform-feasible action for the new observation."""
        host = self._staging(mapped=True)      # None: every page-locked buffer is still referenced -> step_wait() copies
        if host is None:
            res = self.step_tensors(actions, sample=sample)
            self._pending = (res, None, self._last_stream, None)
            return
        _, _, _, fin_off, mark_off = self._stage_offsets()
        base = host.ctypes.data
        mark = None
        if self.spin_wait:   # the step's serial number lands in the buffer when everything is complete: step_wait() spins on it
            mark = self._mark = (self._mark + 1) & 0xffffffff or 1       # (never a value the buffer's word still holds from an earlier step)
        res = self.step_tensors(actions, sample=sample, _host=host,
                                _dropin=(base + fin_off if fin_off is not None else None, base + mark_off if mark is not None else None, mark or 0))
        self._pending = (res, host, self._last_stream, mark)      # the stream THIS step went to (observe() etc. may overwrite _last_stream)

    def step_wait(self):
        """(obs, reward, done, infos) with the reference's types (acktr/envs.py:189-193)."""
        if self._pending is None:
            raise RuntimeError("step_wait() without step_async()")
        (r, host, stream, mark), self._pending = self._pending, None
        fin = None
        E = self.E
        if host is None:
            rew, done = r.host_reward_done(stream)          # one 5-byte-per-bin copy + stream synchronise
            done = done.view(np.bool_)
            reward = torch.from_numpy(rew).unsqueeze(1)
        else:
            _, ro, do, fin_off, mark_off = self._stage_offsets()
            # the kernel wrote reward / done into `host` itself: wait for the step's completion word (or synchronise the stream)
            rc = self.lib.bpp_wait_mark(host.ctypes.data + mark_off, mark, stream) if mark is not None else self.lib.bpp_wait(stream)
            if rc:
                _lib.check(rc)
            reward = torch.from_numpy(host[ro:ro + 4 * E].view("<f4").reshape(E, 1))       # CPU [E,1], acktr/envs.py:192
            done = host[do:do + E].view(np.bool_)       # the kernels write exactly 0 / 1
            if fin_off is not None:     # the gather enqueued with the step is complete too: slice the five arrays (copies: ~28 B per finished bin)
                fo, E8 = fin_off + 32, 8 * E
                n = int(host[fo - 32:fo - 28].view("<i4")[0])
                fin = (host[fo + 3 * E8:fo + 3 * E8 + 4 * n].view("<i4").copy(),
                       (host[fo:fo + 8 * n].view("<f8").copy(), host[fo + E8:fo + E8 + 8 * n].view("<f8").copy()),
                       (host[fo + 2 * E8:fo + 2 * E8 + 4 * n].view("<i4").copy(), host[fo + 2 * E8 + 4 * E:fo + 2 * E8 + 4 * E + 4 * n].view("<i4").copy()))
        t_now = time.time()
        infos = LazyInfos(self, r, t_now, done=done, serial=self._serial, fin=fin)
        if self.monitor is not None and done.any():         # bench/monitor.py:58-72: a row per finished episode
            self.monitor.write(infos.episodes(), t_now)
        return r.obs, reward, done, infos

    def step(self, actions, sample=None):
        self.step_async(actions, sample=sample)
        return self.step_wait()

    def close(self):
        if self.monitor is not None:
            self.monitor.close()
        self._release_side()
        self.closed = True

    def _release_side(self):
        side, self._side = getattr(self, "_side", None), None
        if side is not None and side.value:
            try:
                with torch.cuda.device(self.device):
                    self.lib.bpp_side_destroy(side)
            except Exception:  # noqa: BLE001 -- interpreter shutdown: the runtime may be gone already
                pass

    def __del__(self):
        self._release_side()

    def render(self, mode="human"):
        raise NotImplementedError("BppVecEnv has no renderer (the reference's PackingGame.render is a no-op too)")

    def get_images(self):
        raise NotImplementedError("BppVecEnv has no renderer")

    @property
    def unwrapped(self):
        return self

    # ------------------------------------------------------------------ extras
    def sample_feasible(self, seed, step, mask=None, out=None):
        """Uniform random feasible action per bin (bench/test action source), int64 [E] on the device."""
        m = self.location_masks if mask is None else mask
        if m is None:
            raise RuntimeError("no mask available (compute_mask=False?)")
        if out is None:
            out = torch.empty((self.E,), dtype=torch.int64, device=self.device)
        self._on_device()
        rc = self.lib.bpp_sample_feasible(m.data_ptr(), out.data_ptr(), self.E, m.shape[1], self.env_id_base, int(seed),
                                          int(step), self._stream_ptr())
        if rc:
            _lib.check(rc)
        return out

    def epsilon_override(self, actions, seed, step, eps):
        """bpp_epsilon_override on an int64 [E] device tensor of actions, in place: with probability `eps` a bin's action
        becomes a uniform draw over all act_len entries (the failure-path variant of the benchmark policy)."""
        if actions.device != self.device or actions.dtype != torch.int64 or actions.numel() != self.E or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous int64 [E] tensor on the env's device")
        self._on_device()
        _lib.check(self.lib.bpp_epsilon_override(actions.data_ptr(), self.E, self.M, self.env_id_base, int(seed), int(step),
                                                 _lib.eps_q24(eps), self._stream_ptr()))
        return actions

    def episode_stats(self, reset=False, wide=True, out=None):
        """float64 [4] device tensor: sum of episode returns, sum of final ratios, sum of episode lengths,
        number of episodes finished since the last reset of the accumulators (main.py:159-162).  The per-bin rows
        (`ep_acc` [E,4]) are summed in the fixed order of include/bpp_abi.h: the result is bit-reproducible.
        wide=True: the 1 024 partial sums are computed by 64 workgroups through a scratch buffer (same additions, same
        order, same bits; ~10x shorter than the one-workgroup form, which wide=False selects).  out: float64 [4] device
        tensor the sums are ADDED to (no temporary, no extra launch)."""
        if out is not None:
            if out.device != self.device or out.dtype != torch.float64 or out.numel() != 4 or not out.is_contiguous():
                raise ValueError("out must be a contiguous float64 [4] tensor on the env's device")
            acc = out
        else:
            acc = torch.zeros((4,), dtype=torch.float64, device=self.device)
        self._on_device()
        scratch = None
        if wide:
            scratch = getattr(self, "_reduce_scratch", None)
            if scratch is None:     # zero-filled once; every call leaves its arrival counter at zero again
                scratch = self._reduce_scratch = torch.zeros((_lib.REDUCE_SCRATCH_BYTES // 8,), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.bpp_episode_acc_reduce(self.ep_acc.data_ptr(), self.E, acc.data_ptr(), int(bool(reset)),
                                                   scratch.data_ptr() if scratch is not None else None, self._stream_ptr()))
        return acc

    # ------------------------------------------------------------------ lookahead support
    NOOP = -2 ** 63      # BPP_ACTION_NOOP: this bin is not stepped (state untouched, reward 0, done 0, obs/mask re-emitted)

    def step_subset(self, ids, actions, sample=None):
        """Step only bins `ids` with `actions`; every other bin is left alone (BPP_ACTION_NOOP).  The lookahead
        searches of the reference step one deep-copied env at a time (acktr/reorder.py:245-262, MCTS/node.py:92-137);
        here the copies are bins of the same batch and one launch steps the chosen ones.  Returns StepTensors for
        the whole batch (untouched bins: reward 0, done 0, their current observation and mask)."""
        ids = torch.as_tensor(ids, dtype=torch.int64, device=self.device).reshape(-1)
        a = torch.as_tensor(actions, device=self.device).reshape(-1).to(torch.int64)
        if ids.numel() != a.numel():
            raise ValueError("ids and actions must have the same length")
        full = torch.full((self.E,), self.NOOP, dtype=torch.int64, device=self.device)
        full[ids] = a
        return self.step_tensors(full, sample=sample)

    def observe(self):
        """Re-emit every bin's current observation and mask without stepping (all bins BPP_ACTION_NOOP): what a
        search calls after editing bins (copy_bins / set_current_items)."""
        return self.step_tensors(torch.full((self.E,), self.NOOP, dtype=torch.int64, device=self.device))

    def clone_into(self, src, dst, refresh=True):
        """`copy.deepcopy(env)` of the reference's searches, batched: bins `dst` become exact copies of bins `src`
        (heightmap, counters, Monitor sums, sequence position -- and the item stream in streaming mode); with
        refresh=True the observations and masks of ALL bins are re-emitted so the copies can be read / sampled from
        right away."""
        self.copy_bins(src, dst)
        return self.observe() if refresh else None

    def set_current_items(self, ids, items):
        """Overwrite the item bins `ids` are about to place (int [n,3]); the reorder search plays the previewed items
        in a different order (acktr/reorder.py:181-215).  Only the current item changes; the sequence continues as
        before afterwards.  Call observe() to see the new observation / mask."""
        ids = torch.as_tensor(ids, dtype=torch.int64, device=self.device).reshape(-1)
        it = torch.as_tensor(items, device=self.device).reshape(-1, 3).to(torch.int32)
        if it.shape[0] != ids.numel() or bool((it < 1).any()) or bool((it > 255).any()):
            raise ValueError("items must be [n,3] with sides in 1..255")
        self.state[ids, 8] = it[:, 0] | (it[:, 1] << 8) | (it[:, 2] << 16)     # bpp_env_state.item_cur

    def copy_bins(self, src, dst):
        """Overwrite bins `dst` with the complete state of bins `src` (heightmap + scalar record): the
        device-side replacement of `copy.deepcopy(env)` in the reference's lookahead searches
        (acktr/reorder.py:181,236, MCTS/node.py:92-137).  The copies keep `src`'s sequence position, so
        stepping a copy plays the same upcoming items."""
        src = torch.as_tensor(src, dtype=torch.int64, device=self.device).reshape(-1)
        dst = torch.as_tensor(dst, dtype=torch.int64, device=self.device).reshape(-1)
        if src.numel() != dst.numel():
            raise ValueError("src and dst must have the same length")
        if self._stream is not None:
            copy_bin_records(self.hmap, self.state, src, dst, ring=self.pool, mt=self._mt, gen_next=self.gen_next,
                             depth=self.stream_spec["depth"])
            self._reset_seq_cache()
        else:
            copy_bin_records(self.hmap, self.state, src, dst)

    def preview(self, k):
        """The next `k` items of every bin, int32 [E, k, 3] -- `box_creator.preview(k)`
        (envs/bpp0/binCreator.py:15-18) for all bins at once (the terminator repeats past the end)."""
        st = self.state
        cursor, seq = st[:, 0].long(), st[:, 7].long()
        T = self.pool.shape[1]
        hdr = 2 if self._stream is not None else 0       # ring rows start with two look-ahead entries (include/bpp_abi.h)
        idx = hdr + torch.clamp(cursor.unsqueeze(1) + torch.arange(int(k), device=self.device).unsqueeze(0), max=T - 1 - hdr)
        return self.pool[seq.unsqueeze(1), idx][:, :, :3].to(torch.int32)

    def heightmaps(self):
        """`Space.plain` of every bin as int32 [E, W, L] (the state itself is kept as bytes)."""
        return self.hmap.view(self.E, self.W, self.L).to(torch.int32)

    def state_dict(self):
        """Env checkpoint (the reference never checkpoints env state; a handful of tensors here): byte
        heightmaps, per-bin records, per-bin episode accumulators and -- so that a loop can resume mid-rollout -- the
        last observation and its mask."""
        sd = {"format": STATE_FORMAT, "hmap": self.hmap.clone(), "state": self.state.clone(), "ep_acc": self.ep_acc.clone(),
              "first_reset": self._first_reset}
        if self._stream is not None:   # streaming supply: the ring, every bin's generator and its progress
            sd.update(stream_ring=self.pool.clone(), stream_mt=self._mt.clone(), stream_gen_next=self.gen_next.clone(),
                      stream_since_refill=self._since_refill, stream_spec=self._stream_identity(), stream_layout=STREAM_LAYOUT)
        if self._bufs is not None:
            sd["obs"] = self._bufs["obs"].clone()
            if self._bufs["mask"] is not None:
                sd["mask"] = self._bufs["mask"].clone()
        return sd

    def _stream_identity(self):
        """What must agree for a streaming checkpoint to continue the same item streams."""
        sp = self.stream_spec
        return dict(bound=tuple(sp["bound"]), seed=sp["seed"], depth=sp["depth"], pool_len=sp["pool_len"],
                    num_envs=self.E, env_id_base=self.env_id_base, bin_size=self.bin_size, rng=sp["rng"])

    def load_state_dict(self, sd):
        if self._stream is None and "stream_ring" in sd:
            raise ValueError("checkpoint of a streaming env loaded into a pool-based env (its ring and generators would be dropped)")
        if self._stream is not None:
            if "stream_ring" not in sd:
                raise ValueError("checkpoint of a pool-based env loaded into a streaming env")
            layout = sd.get("stream_layout")
            if layout is None:
                # checkpoints written before the key existed: layout 1 (ABI <= 11) kept raw 32-bit MT19937 outputs -- another
                # record width than layout 2's byte outputs (ABI 12 / 13 wrote layout 2 without saying so) -- and rows without the
                # two look-ahead entries: the buffer shapes tell them apart
                same = (tuple(sd["stream_ring"].shape) == tuple(self.pool.shape) and tuple(sd["stream_mt"].shape) == tuple(self._mt.shape))
                layout = STREAM_LAYOUT if same else 1
            if int(layout) != STREAM_LAYOUT:
                raise ValueError("streaming checkpoint with ring / generator layout %d, this build reads layout %d: its rows would be "
                                 "played as other items" % (int(layout), STREAM_LAYOUT))
            want, got = self._stream_identity(), sd.get("stream_spec")
            if got is not None:
                got = dict(got)
                got.setdefault("rng", "mt19937")      # checkpoints older than the counter generator
            if got is not None and dict(got) != want:
                raise ValueError("checkpoint stream_spec %r does not match this env's %r" % (dict(got), want))
            if tuple(sd["stream_ring"].shape) != tuple(self.pool.shape) or tuple(sd["stream_mt"].shape) != tuple(self._mt.shape):
                raise ValueError("checkpoint stream_spec: ring / generator buffers have another shape than this env's")
        if tuple(sd["hmap"].shape) != tuple(self.hmap.shape):
            raise ValueError("checkpoint holds %r heightmaps, this env %r" % (tuple(sd["hmap"].shape), tuple(self.hmap.shape)))
        fmt = int(sd.get("format", 0))
        if fmt > STATE_FORMAT:
            raise ValueError("checkpoint format %d is newer than this build's %d" % (fmt, STATE_FORMAT))
        if "stats" in sd and "ep_acc" not in sd:
            import warnings
            warnings.warn("legacy checkpoint: its `stats` (slotted episode sums of ABI < 10) cannot be mapped onto the per-bin "
                          "accumulators; episode statistics restart from zero", RuntimeWarning)
        self.hmap.copy_(sd["hmap"])
        self.state.copy_(sd["state"])
        # bpp_env_state.hmax (word 11; `pad` before format 1) is derived data the 20x20 kernel trusts: always recompute
        # it from the heightmaps instead of believing the checkpoint
        self.state[:, 11] = self.hmap.max(dim=1).values.to(torch.int32)
        if "ep_acc" in sd:
            self.ep_acc.copy_(sd["ep_acc"])
        else:
            self.ep_acc.zero_()
        self._first_reset = bool(sd["first_reset"])
        if self._stream is not None:
            self.pool.copy_(sd["stream_ring"])
            self._mt.copy_(sd["stream_mt"])
            self.gen_next.copy_(sd["stream_gen_next"])
            self._since_refill = int(sd["stream_since_refill"])
            self._reset_seq_cache()
            if self._since_refill >= self.refill_every:   # (written by an env with a longer refill period)
                self.refill()
        if "obs" in sd:
            bufs, _ = self._buffers()
            if self._res is None or self.fresh_outputs:
                self._res = bufs
            self._serial += 1          # (the output buffers change under whoever cached rows of them: masks.ROW_CACHE, LazyInfos)
            bufs["obs"].copy_(sd["obs"])
            if "mask" in sd and bufs["mask"] is not None:
                bufs["mask"].copy_(sd["mask"])
        
    def state_numpy(self):
        """bpp_env_state[E] as a structured numpy array (tests)."""
        dt = np.dtype([("cursor", "<i4"), ("episode", "<i4"), ("n_boxes", "<i4"), ("vol_sum", "<i4"),
                       ("ep_ret", "<f8"), ("ep_len", "<i4"), ("seq", "<i4"), ("item_cur", "<u4"), ("item_next", "<u4"),
                       ("item_reset", "<u4"), ("hmax", "<u4")])
        return self.state.cpu().numpy().view(dt).reshape(-1)
