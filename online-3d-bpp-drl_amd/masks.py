"""Drop-in replacements for the mask helpers the ACKTR loop calls per observation row
(acktr/utils.py:37-94), plus batched device versions.  All of them run the HIP kernel
`bpp_mask_from_obs` / `bpp_mask_from_hmap`; there is no host fallback."""
import ctypes

import numpy as np
import torch

from . import _lib


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("mask kernels need a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def batched_mask_from_obs(obs, container_size, enable_rotation=False, rule="utils", out=None):
    """obs: [E, 4*W*L] (torch tensor or numpy; moved to the GPU if needed) -> float32 [E, M] device tensor.
    Row e equals torch.FloatTensor(get_possible_position(obs[e], size)) (or get_rotation_mask)."""
    W, L, H = (int(v) for v in container_size)
    if not torch.is_tensor(obs):
        obs = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32))
    dev = obs.device if obs.device.type == "cuda" else _dev()
    o = obs.reshape(-1, 4 * W * L).to(device=dev, dtype=torch.float32).contiguous()
    E, M = o.shape[0], W * L * (1 + int(bool(enable_rotation)))
    if out is None:
        out = torch.empty((E, M), dtype=torch.float32, device=dev)
    r = {"utils": _lib.RULE_UTILS, "space": _lib.RULE_SPACE}[rule]
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().bpp_mask_from_obs(o.data_ptr(), out.data_ptr(), E, W, L, H, int(bool(enable_rotation)), r,
                                                _stream(dev)))
    return out


def batched_mask_from_hmap(hmap, items, container_size, enable_rotation=False, rule="space", out=None):
    """hmap int32 [E, W*L], items int32 [E,3] -> float32 [E, M]; rule='space', rotation off is
    PackingGame.get_possible_position (envs/bpp0/bin3D.py:72-93) for every bin."""
    W, L, H = (int(v) for v in container_size)
    dev = hmap.device if torch.is_tensor(hmap) and hmap.device.type == "cuda" else _dev()
    h = torch.as_tensor(hmap).reshape(-1, W * L).to(device=dev, dtype=torch.int32).contiguous()
    it = torch.as_tensor(items).reshape(-1, 3).to(device=dev, dtype=torch.int32).contiguous()
    if it.shape[0] != h.shape[0]:
        raise ValueError("hmap and items disagree on the number of bins")
    E, M = h.shape[0], W * L * (1 + int(bool(enable_rotation)))
    if out is None:
        out = torch.empty((E, M), dtype=torch.float32, device=dev)
    r = {"utils": _lib.RULE_UTILS, "space": _lib.RULE_SPACE}[rule]
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().bpp_mask_from_hmap(h.data_ptr(), it.data_ptr(), out.data_ptr(), E, W, L, H,
                                                 int(bool(enable_rotation)), r, _stream(dev)))
    return out


def batched_window_masks(hmap, items, window_size, stride=10, enable_rotation=False, rule="utils"):
    """Sliding-window masks of a larger pallet, batched (multi_bin/multi_bin.py:7-17 `slipingWindow` + :28-45: every
    window of `window_size` = (w, l, H) over the big heightmap becomes its own bin and gets
    get_possible_position(window_obs, window_size)).  hmap: int [E, Wb, Lb]; items: int [E, 3].
    Returns (masks float32 [E, n_windows, M], offsets int64 [n_windows, 2]) with windows in the reference's order
    (dx outer, dy inner, both stepping by `stride`)."""
    w, l, H = (int(v) for v in window_size)
    dev = hmap.device if torch.is_tensor(hmap) and hmap.device.type == "cuda" else _dev()
    h = torch.as_tensor(hmap).to(device=dev, dtype=torch.int32)
    if h.dim() != 3:
        raise ValueError("hmap must be [E, Wb, Lb]")
    E, Wb, Lb = h.shape
    it = torch.as_tensor(items).reshape(-1, 3).to(device=dev, dtype=torch.int32)
    if it.shape[0] != E:
        raise ValueError("hmap and items disagree on the number of bins")
    xs = list(range(0, Wb - w + 1, int(stride)))
    ys = list(range(0, Lb - l + 1, int(stride)))
    if not xs or not ys:
        raise ValueError("window larger than the pallet")
    wins = torch.stack([h[:, dx:dx + w, dy:dy + l].reshape(E, w * l) for dx in xs for dy in ys], dim=1)   # [E, n, w*l]
    n = wins.shape[1]
    masks = batched_mask_from_hmap(wins.reshape(E * n, w * l), it.repeat_interleave(n, dim=0), (w, l, H), enable_rotation, rule)
    offsets = torch.tensor([(dx, dy) for dx in xs for dy in ys], dtype=torch.int64)
    return masks.view(E, n, -1), offsets


class _RowHelper(object):
    """Host side of the per-ROW mask helpers (main.py:163-169 calls them once per observation row): one page-locked buffer
    per device that the mask kernel writes its float32 row into directly, a completion word behind it (bpp_mark) and a spin
    on that word (bpp_wait_mark) -- a launch and a few microseconds, instead of a launch, a cast kernel, a device-to-host
    copy and a stream synchronisation per row (37 -> ~15 us per call; the literal 3-line swap of INTEGRATION 4.1 is bound by
    exactly this)."""
    _per_device = {}

    def __init__(self, dev):
        self.dev = dev
        self.cap = 2048                                     # largest action space the kernels support: 2 * 1024
        self.pinned = torch.empty((4 * self.cap + 8,), dtype=torch.uint8).pin_memory()
        self.host = self.pinned.numpy()
        self.host[4 * self.cap:] = 0
        self.base = self.host.ctypes.data
        self.serial = 0

    @classmethod
    def of(cls, dev):
        h = cls._per_device.get(dev.index)
        if h is None:
            h = cls._per_device[dev.index] = cls(dev)
        return h

    def row(self, observation, W, L, H, rotation):
        A = W * L
        M = A * (2 if rotation else 1)
        lib = _lib.lib()
        if torch.cuda.current_device() != self.dev.index:
            torch.cuda.set_device(self.dev)
        sp = _stream(self.dev)
        self.serial = (self.serial + 1) & 0xffffffff or 1
        flag = self.base + 4 * self.cap
        _lib.check(lib.bpp_mask_from_obs(observation.data_ptr(), self.base, 1, W, L, H, int(bool(rotation)), _lib.RULE_UTILS, sp))
        _lib.check(lib.bpp_mark(flag, self.serial, sp))
        _lib.check(lib.bpp_wait_mark(flag, self.serial, sp))
        return self.host[:4 * M].view("<f4")


# Envs whose CURRENT observation buffer the per-row helpers may be handed rows of (weak references; BppVecEnv registers itself).
# main.py:163-169 calls the helper once per row of the observation the env has just returned -- and the fused step kernel has
# already written the mask of exactly those rows (rule U, acktr/utils.py:8-35: the env's default `mask_rule`).  For such a row
# the helper hands back the env's own mask row: the whole [E, M] mask is fetched ONCE per lock-step (first helper call after a
# step), every further row of that step costs a slice.  Anything else -- a row of an older step, a CPU tensor, an array, a tensor
# the caller built -- takes the kernel path below.  The observation must not have been modified in place (the reference loop never
# does); ROW_CACHE = False switches the shortcut off.
ROW_CACHE = True
ROW_CACHE_MAX_BINS = 4096          # beyond that one fetch of the whole mask costs more than the rows a caller can reasonably ask for
_ENVS = []


def register_env(env):
    """Called by BppVecEnv's constructor."""
    import weakref
    _ENVS[:] = [r for r in _ENVS if r() is not None]
    _ENVS.append(weakref.ref(env))


def _env_mask_row(observation, W, L, H, rotation):
    """int32 [M] row of the mask the env computed for this very observation row, or None."""
    ptr = observation.data_ptr()
    for ref in _ENVS:
        env = ref()
        res = getattr(env, "_res", None) if env is not None else None
        if res is None or env.E > ROW_CACHE_MAX_BINS or getattr(env, "closed", False):
            continue
        if (env.W, env.L, env.H) != (W, L, H) or bool(env.can_rotate) != bool(rotation) or not env.compute_mask or env.mask_rule != _lib.RULE_UTILS:
            continue
        obs = res.obs
        row_bytes = 16 * W * L
        off = ptr - obs.data_ptr()
        if off < 0 or off >= env.E * row_bytes or off % row_bytes or obs.device != observation.device:
            continue
        key = (env._serial, obs.data_ptr())
        cached = getattr(env, "_mask_rows_host", None)
        if cached is None or cached[0] != key:
            cached = env._mask_rows_host = (key, res.mask.to(torch.int32).cpu().numpy())      # ONE fetch per lock-step (synchronises)
        return cached[1][off // row_bytes]
    return None


def _row_mask(observation, container_size, rotation):
    """int32 ndarray [M]: the mask of ONE observation row, whatever it is handed (device row, CPU tensor, ndarray)."""
    W, L, H = (int(v) for v in container_size)
    if (torch.is_tensor(observation) and observation.device.type == "cuda" and observation.dtype == torch.float32
            and observation.is_contiguous() and observation.numel() == 4 * W * L and observation.data_ptr() % 16 == 0):
        if ROW_CACHE and _ENVS:
            row = _env_mask_row(observation, W, L, H, rotation)
            if row is not None:
                return row.copy()
        return _RowHelper.of(observation.device).row(observation, W, L, H, rotation).astype(np.int32)
    m = batched_mask_from_obs(observation, container_size, rotation)
    return m[0].to(torch.int32).cpu().numpy().reshape(-1)


def get_possible_position(observation, container_size):
    """Same signature and return type as acktr.utils.get_possible_position (acktr/utils.py:37-62):
    one observation row -> python list of W*L ints."""
    return _row_mask(observation, container_size, False).tolist()


def get_rotation_mask(observation, container_size):
    """Same as acktr.utils.get_rotation_mask (acktr/utils.py:64-94): int32 ndarray [2*W*L]."""
    return _row_mask(observation, container_size, True)


def masked_act(logits, location_masks, seed=0, step=0, deterministic=False, env_id_base=0, counter=None, out=None):
    """Fused replacement of the inference half of `Categorical.forward` + `dist.sample()/mode()` +
    `dist.log_probs(action)` in `Policy.act` (acktr/model.py:56-68, acktr/distributions.py:71-84):
    logits [E,M] (output of the policy's linear layer) and location_masks [E,M] float32 device tensors ->
    (action int64 [E,1], action_log_probs float32 [E,1]).  Sampling uses a counter-based stream keyed by
    (seed, global bin id, step) instead of torch.multinomial.
    counter: int64 / uint64 device tensor [2] = (seed, step) read BY THE KERNEL (bpp_masked_act_counter) -- for loops captured
    in a HIP graph, where `step` passed by value would be frozen into the captured launch: advance `counter[1]` on the
    device (`counter[1:].add_(1)`) inside the captured region.  out = (action, log_prob) tensors to write into (static outputs for a captured region)."""
    dev = logits.device if logits.device.type == "cuda" else _dev()
    x = logits.to(device=dev, dtype=torch.float32).contiguous()
    m = location_masks.to(device=dev, dtype=torch.float32).contiguous()
    if x.shape != m.shape or x.dim() != 2:
        raise ValueError("logits and location_masks must both be [E, M]")
    E, M = x.shape
    if out is not None:
        action, logp = out
        if (action.dtype != torch.int64 or logp.dtype != torch.float32 or action.numel() != E or logp.numel() != E
                or action.device != dev or logp.device != dev or not action.is_contiguous() or not logp.is_contiguous()):
            raise ValueError("out must be (int64 [E,1], float32 [E,1]) contiguous tensors on the logits' device")
    else:
        action = torch.empty((E, 1), dtype=torch.int64, device=dev)
        logp = torch.empty((E, 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if counter is not None:
            if counter.device != dev or counter.numel() != 2 or counter.dtype not in (torch.int64, torch.uint64) or not counter.is_contiguous():
                raise ValueError("counter must be a contiguous int64 / uint64 [2] tensor (seed, step) on the logits' device")
            _lib.check(_lib.lib().bpp_masked_act_counter(x.data_ptr(), m.data_ptr(), action.data_ptr(), logp.data_ptr(), E, M,
                                                         int(env_id_base), counter.data_ptr(), int(bool(deterministic)), _stream(dev)))
        else:
            _lib.check(_lib.lib().bpp_masked_act(x.data_ptr(), m.data_ptr(), action.data_ptr(), logp.data_ptr(), E, M,
                                                 int(env_id_base), int(seed), int(step), int(bool(deterministic)), _stream(dev)))
    return action, logp


class _MaskedEvaluate(torch.autograd.Function):
    """bpp_masked_evaluate / bpp_masked_evaluate_backward as one differentiable op (no saved probabilities:
    the backward kernel recomputes the row statistics from the logits)."""

    @staticmethod
    def forward(ctx, logits, location_masks, action):
        E, M = logits.shape
        dev = logits.device
        logp = torch.empty(E, dtype=torch.float32, device=dev)
        ent = torch.empty(E, dtype=torch.float32, device=dev)
        bad = torch.empty(E, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().bpp_masked_evaluate(logits.data_ptr(), location_masks.data_ptr(), action.data_ptr(),
                                                      logp.data_ptr(), ent.data_ptr(), bad.data_ptr(), E, M, _stream(dev)))
        ctx.save_for_backward(logits, location_masks, action)
        return logp, ent, bad

    @staticmethod
    def backward(ctx, g_logp, g_ent, g_bad):
        logits, location_masks, action = ctx.saved_tensors
        E, M = logits.shape
        dev = logits.device
        g = [torch.zeros(E, dtype=torch.float32, device=dev) if v is None else v.to(torch.float32).contiguous()
             for v in (g_logp, g_ent, g_bad)]
        grad = torch.empty_like(logits)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().bpp_masked_evaluate_backward(logits.data_ptr(), location_masks.data_ptr(), action.data_ptr(),
                                                               g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                                               grad.data_ptr(), E, M, _stream(dev)))
        return grad, None, None


def masked_evaluate(logits, location_masks, action):
    """Fused, differentiable replacement of `Categorical.forward` + `dist.log_probs(action)` + `dist.entropy()` and of the
    `bx` term in `Policy.evaluate_actions` (acktr/model.py:90-96, acktr/distributions.py:71-101):
    logits [E,M] (requires_grad), location_masks [E,M], action int64 [E] or [E,1] ->
    (action_log_probs [E,1], dist_entropy = dist.entropy().mean(), prob_loss = bad_prob.mean() — the only way the loop
    consumes `bx`, acktr/algo/acktr_pipeline.py:66)."""
    if logits.device.type != "cuda":
        raise RuntimeError("masked_evaluate needs the logits on a HIP device")
    x = logits.to(torch.float32).contiguous()
    m = location_masks.to(device=x.device, dtype=torch.float32).contiguous()
    a = action.to(device=x.device, dtype=torch.int64).reshape(-1).contiguous()
    if x.shape != m.shape or x.dim() != 2 or a.numel() != x.shape[0]:
        raise ValueError("logits and location_masks must both be [E, M], action [E]")
    logp, ent, bad = _MaskedEvaluate.apply(x, m, a)
    return logp.unsqueeze(1), ent.mean(), bad.sum() / float(x.numel())
