"""ctypes binding of libbpp_hip.so (include/bpp_abi.h) -- the only way the package reaches the kernels.

There is deliberately NO fallback: if the HIP library cannot be built/loaded, or a call fails, a
RuntimeError is raised.  The CPU oracle under oracle/ is never imported from here.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "bpp_kernels.hip")
# build() only ever writes BUILD_LIB.  BPP_HIP_LIB: LOAD another build of the same library instead (profiling /
# diagnostic builds of tools/build_variant.sh abl -DBPP_ENABLE_ABLATION, tools/stress_stats.py) -- never built to, never a fallback
BUILD_LIB = os.path.join(CSRC, "libbpp_hip.so")
LIB = os.environ.get("BPP_HIP_LIB") or BUILD_LIB
HDR = os.path.join(os.path.dirname(HERE), "include", "bpp_abi.h")
DEPS = [SRC, HDR, os.path.join(CSRC, "bpp_tile_kernel.inl"), os.path.join(CSRC, "bpp_tile_body.inl"), os.path.join(CSRC, "bpp_stream_gen.inl"), os.path.join(CSRC, "bpp_heads.inl"), os.path.join(CSRC, "bpp_rt_kernels.inl"), os.path.join(CSRC, "bpp_stats.inl"), os.path.join(os.path.dirname(HERE), "include", "bpp_gen.inl")]

ABI_VERSION = 16
STREAM_RNG_MT19937, STREAM_RNG_COUNTER = 0, 1
RULE_UTILS, RULE_SPACE = 0, 1
RESET_INIT, RESET_ADVANCE = 0, 1
REDUCE_LANES = 1024
REDUCE_SCRATCH_BYTES = REDUCE_LANES * 4 * 8 + 64

SYMBOLS = ["bpp_abi_version", "bpp_last_error", "bpp_limits", "bpp_reset", "bpp_step", "bpp_mask_from_obs",
           "bpp_mask_from_hmap", "bpp_sample_feasible", "bpp_episode_stats", "bpp_rollout_uniform", "bpp_masked_act", "bpp_gen_cut2", "bpp_gen_cut1", "bpp_gen_rs",
           "bpp_get_knobs", "bpp_set_knobs", "bpp_launch_info", "bpp_stream_sizes", "bpp_stream_init", "bpp_stream_refill",
           "bpp_rollout_uniform_stream", "bpp_masked_evaluate", "bpp_masked_evaluate_backward", "bpp_episode_acc_reduce", "bpp_rollout_uniform_sets", "bpp_fetch_to_host", "bpp_wait", "bpp_gather_finished", "bpp_epsilon_override", "bpp_side_create", "bpp_side_destroy", "bpp_mark", "bpp_wait_mark", "bpp_step_dropin", "bpp_masked_act_counter"]


class Batch(ctypes.Structure):
    """struct bpp_batch"""
    _fields_ = [("num_envs", ctypes.c_int32), ("W", ctypes.c_int32), ("L", ctypes.c_int32), ("H", ctypes.c_int32),
                ("rotation", ctypes.c_int32), ("mask_rule", ctypes.c_int32), ("pool_size", ctypes.c_int32),
                ("pool_len", ctypes.c_int32), ("env_id_base", ctypes.c_int64), ("env_id_total", ctypes.c_int64),
                ("seq_pool", ctypes.c_void_p), ("hmap", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("ep_acc", ctypes.c_void_p), ("pool_mode", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                ("seq_cache", ctypes.c_void_p)]


class Stream(ctypes.Structure):
    """struct bpp_stream"""
    _fields_ = [("num_envs", ctypes.c_int32), ("depth", ctypes.c_int32), ("pool_len", ctypes.c_int32), ("W", ctypes.c_int32),
                ("L", ctypes.c_int32), ("H", ctypes.c_int32), ("bound_lo", ctypes.c_int32), ("bound_hi", ctypes.c_int32),
                ("env_id_base", ctypes.c_int64), ("seed0", ctypes.c_uint64), ("ring", ctypes.c_void_p), ("mt", ctypes.c_void_p),
                ("work", ctypes.c_void_p), ("gen_next", ctypes.c_void_p), ("state", ctypes.c_void_p), ("overflow", ctypes.c_void_p),
                ("rng", ctypes.c_int32), ("reserved1", ctypes.c_int32)]


POOL_STATIC, POOL_RING = 0, 1
SEQ_CACHE_BYTES_PER_BIN = 2 * 128 + 8 + 8   # BPP_SEQ_CACHE_BYTES(E) / E
ROLLOUT_CONTINUE = 1


def eps_q24(eps):
    """epsilon as the 24-bit fraction bpp_epsilon_override takes"""
    q24 = int(round(float(eps) * (1 << 24)))
    if not 0 <= q24 <= 1 << 24:
        raise ValueError("epsilon must be in [0, 1]")
    return q24


def rollout_eps_flags(eps):
    """BPP_ROLLOUT_EPS(q24): the flags bits of bpp_rollout_uniform_sets that carry epsilon, as the int32 the ABI takes"""
    v = (min(eps_q24(eps), (1 << 24) - 1) << 8) & 0xFFFFFFFF      # the field holds 24 bits: eps = 1.0 is clamped, not wrapped to 0
    return v - (1 << 32) if v & 0x80000000 else v


class StepOut(ctypes.Structure):
    """struct bpp_step_out"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len",
                                               "next_action")] + [("sample_seed", ctypes.c_uint64),
                                                                  ("sample_step", ctypes.c_uint64),
                                                                  ("host_reward", ctypes.c_void_p), ("host_done", ctypes.c_void_p)]


class Knobs(ctypes.Structure):
    """struct bpp_knobs"""
    _fields_ = [("bins_per_wave", ctypes.c_int32), ("waves_per_group", ctypes.c_int32), ("xcd_remap", ctypes.c_int32),
                ("force_generic", ctypes.c_int32), ("ablate", ctypes.c_int32), ("legacy_fast", ctypes.c_int32),
                ("tile_groups", ctypes.c_int32), ("stream_legacy", ctypes.c_int32), ("stream_overlap", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 3)]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force=False, verbose=False):
    """Compile csrc/bpp_kernels.hip for gfx950 into csrc/libbpp_hip.so (in-tree; no-op when fresh)."""
    if LIB != BUILD_LIB:        # an explicitly chosen build is loaded as it is
        if not os.path.exists(LIB):
            raise RuntimeError("BPP_HIP_LIB=%s does not exist (it is never built implicitly)" % LIB)
        return LIB

    def fresh():
        return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)

    if not force and fresh():
        return LIB
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:      # several ranks may get here at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not fresh():
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-pass-failed", "-fPIC",
                   "-shared", "-o", tmp, SRC]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            # not a fallback: the product itself gets compiled (hipcc cross-compiles without a GPU)
            try:
                build()
            except Exception as exc:  # noqa: BLE001
                raise RuntimeError("libbpp_hip.so is missing and could not be built with hipcc (%s); run "
                                   "`python -c 'import __graft_entry__ as g; g.build()'`" % (exc,))
        L = ctypes.CDLL(LIB)
        L.bpp_abi_version.restype = ctypes.c_int
        L.bpp_last_error.restype = ctypes.c_char_p
        L.bpp_limits.argtypes = [ctypes.POINTER(ctypes.c_int32)]
        L.bpp_reset.argtypes = [ctypes.POINTER(Batch), ctypes.c_int32, ctypes.POINTER(StepOut), ctypes.c_void_p]
        L.bpp_step.argtypes = [ctypes.POINTER(Batch), ctypes.c_void_p, ctypes.POINTER(StepOut), ctypes.c_void_p]
        L.bpp_mask_from_obs.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 6 + [ctypes.c_void_p]
        L.bpp_mask_from_hmap.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 6 + [ctypes.c_void_p]
        L.bpp_sample_feasible.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.bpp_epsilon_override.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64,
                                           ctypes.c_uint32, ctypes.c_void_p]
        L.bpp_rollout_uniform.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(StepOut), ctypes.c_void_p, ctypes.c_uint64,
                                          ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_rollout_uniform_sets.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(StepOut), ctypes.c_int32, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_void_p]
        L.bpp_masked_act.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64,
                                     ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_masked_evaluate.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_masked_evaluate_backward.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_gen_cut2.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 7 + [ctypes.c_uint64, ctypes.c_int32]
        L.bpp_gen_cut1.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 5 + [ctypes.c_void_p, ctypes.c_int32,
                                                                                     ctypes.c_uint64, ctypes.c_int32]
        L.bpp_gen_rs.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 5 + [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64,
                                                                          ctypes.c_int32]
        L.bpp_episode_stats.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.bpp_wait.argtypes = [ctypes.c_void_p]
        L.bpp_mark.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        L.bpp_wait_mark.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        L.bpp_masked_act_counter.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_int32, ctypes.c_void_p]
        L.bpp_step_dropin.argtypes = [ctypes.POINTER(Batch), ctypes.c_void_p, ctypes.POINTER(StepOut), ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_uint32, ctypes.c_void_p]
        L.bpp_gather_finished.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        L.bpp_fetch_to_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.bpp_episode_acc_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.bpp_stream_sizes.argtypes = [ctypes.POINTER(Stream), ctypes.POINTER(ctypes.c_int64)]
        L.bpp_stream_init.argtypes = [ctypes.POINTER(Stream), ctypes.c_void_p]
        L.bpp_stream_refill.argtypes = [ctypes.POINTER(Stream), ctypes.c_void_p]
        L.bpp_rollout_uniform_stream.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(StepOut), ctypes.c_void_p, ctypes.c_uint64,
                                                 ctypes.c_uint64, ctypes.c_int32, ctypes.POINTER(Stream), ctypes.c_int32,
                                                 ctypes.c_void_p, ctypes.c_void_p]
        L.bpp_side_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        L.bpp_side_destroy.argtypes = [ctypes.c_void_p]
        L.bpp_launch_info.argtypes = [ctypes.c_int32] * 5 + [ctypes.POINTER(ctypes.c_int32)]
        L.bpp_get_knobs.argtypes = [ctypes.POINTER(Knobs)]
        L.bpp_set_knobs.argtypes = [ctypes.POINTER(Knobs)]
        if L.bpp_abi_version() != ABI_VERSION:
            raise RuntimeError("libbpp_hip.so ABI version %d != %d" % (L.bpp_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libbpp_hip error %d: %s" % (rc, lib().bpp_last_error().decode()))


def get_knobs():
    """Current launch-shape knobs as a dict (include/bpp_abi.h: bpp_knobs)."""
    k = Knobs()
    check(lib().bpp_get_knobs(ctypes.byref(k)))
    return {n: int(getattr(k, n)) for n in ("bins_per_wave", "waves_per_group", "xcd_remap", "force_generic", "ablate", "legacy_fast", "tile_groups",
                                              "stream_legacy", "stream_overlap")}


def set_knobs(**kw):
    """Change launch-shape knobs for the whole process (results never depend on them); returns the previous
    settings so a caller can restore them: `old = set_knobs(force_generic=1); ...; set_knobs(**old)`."""
    old = get_knobs()
    new = dict(old)
    for name, v in kw.items():
        if name not in new:
            raise TypeError("unknown knob %r" % (name,))
        new[name] = int(v)
    k = Knobs(new["bins_per_wave"], new["waves_per_group"], new["xcd_remap"], new["force_generic"], new["ablate"],
              new["legacy_fast"], new["tile_groups"], new["stream_legacy"], new["stream_overlap"])
    check(lib().bpp_set_knobs(ctypes.byref(k)))
    return old


KERNEL_NAMES = {0: "bpp_kernel (cell scan)", 1: "bpp_fast_kernel (prefix image, runtime geometry)",
                2: "bpp_tile_kernel (prefix image, compile-time geometry)"}


def launch_info(E, size, rotation=False):
    """Kernel and launch shape used for a geometry under the current knobs (bpp_launch_info)."""
    out = (ctypes.c_int32 * 6)()
    check(lib().bpp_launch_info(int(E), int(size[0]), int(size[1]), int(size[2]), int(bool(rotation)), out))
    return {"kernel": int(out[0]), "kernel_name": KERNEL_NAMES.get(int(out[0]), "?"), "K": int(out[1]), "bins_per_wave": int(out[2]),
            "waves_per_group": int(out[3]), "workgroups": int(out[4]), "lds_bytes": int(out[5])}


def limits():
    out = (ctypes.c_int32 * 2)()
    check(lib().bpp_limits(out))
    return int(out[0]), int(out[1])
