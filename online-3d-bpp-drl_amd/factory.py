"""`make_vec_envs` with the reference's signature (acktr/envs.py:77-118) returning a BppVecEnv.

    envs = make_vec_envs(env_name, seed, num_processes, gamma, log_dir, device, allow_early_resets, args=args)

Pass `stream=dict(bound=(lo, hi), seed=s, depth=D)` instead of a pool for the endless device-generated CUT-2 supply.
`args` is the reference's argparse namespace (acktr/arguments.py): `container_size`, `enable_rotation`,
`data_type` ('cut1' | 'cut2' | 'rs', bin3D.py:21-32) and `box_size_set` are honoured; `gamma`,
`allow_early_resets`, `num_frame_stack` are accepted for signature compatibility (the reference disables
VecNormalize's filters anyway, acktr/envs.py:112).  `log_dir` (acktr/envs.py:54-58 wraps every worker in
`bench.Monitor(env, os.path.join(log_dir, str(rank)))`): when given, `step()` appends the finished episodes to
`<log_dir>/<shard rank>.monitor.csv` in Monitor's format -- ONE file per shard with an extra `bin` column instead of one file
per bin (vec_env.MonitorCsv; the reference's `bench.load_results(log_dir)` reads it); `log_dir=None` writes nothing.  The
tensor-native `step_tensors` path never writes rows (episode statistics there come from `EpisodeStats`).
Unlike the reference, `seed` really seeds the item
sequences (the reference's `env.seed` is a no-op and its forked workers all draw the same stream)."""
from . import sequences
from .vec_env import BppVecEnv, MonitorCsv


def make_pool(container_size, data_type="cut2", box_size_set=None, enable_rotation=False, seed=0, pool_size=4096):
    size = tuple(int(v) for v in container_size)
    if box_size_set:
        lo = tuple(int(v) for v in box_size_set[0])
        hi = tuple(int(v) for v in box_size_set[-1])
    else:
        lo, hi = (2, 2, 2), (5, 5, 5)
    if data_type == "cut2":       # bin3D.py:30-32: MDlayerBoxCreator(container_size, [box_set[0][0], box_set[-1][0]])
        return sequences.cut2_pool(size, pool_size, seed=seed, bound=(lo[0], hi[0]))
    if data_type == "cut1":       # bin3D.py:24-29: CuttingBoxCreator(container_size, low + up, can_rotate)
        return sequences.cut1_pool(size, pool_size, seed=seed, box_range=lo + hi, rotation=enable_rotation)
    if data_type == "rs":         # bin3D.py:21-23: RandomBoxCreator(box_set); enough items to fill any bin
        vmin = max(1, min(x * y * z for x, y, z in (box_size_set or [lo])))
        length = min(4096, size[0] * size[1] * size[2] // vmin + 2)
        return sequences.rs_pool(size, pool_size, length, seed=seed, box_set=box_size_set)
    raise ValueError("unknown data_type %r" % (data_type,))


def make_vec_envs(env_name, seed, num_processes, gamma, log_dir, device, allow_early_resets,
                  num_frame_stack=None, args=None, pool=None, pool_size=4096, **env_kwargs):
    if args is None:
        raise ValueError("args (the reference's argparse namespace) is required")
    size = tuple(int(v) for v in args.container_size)
    rot = bool(getattr(args, "enable_rotation", False))
    if pool is None and env_kwargs.get("stream") is None:
        pool = make_pool(size, getattr(args, "data_type", getattr(args, "item_seq", "cut2")),
                         getattr(args, "box_size_set", None), rot, seed=int(seed), pool_size=pool_size)
    env_kwargs.setdefault("fresh_outputs", True)   # reference semantics: earlier results stay valid
    env_kwargs.setdefault("eager_infos", True)     # the reference loop reads the finished episodes' infos every step (main.py:159-162)
    env = BppVecEnv(int(num_processes), size, enable_rotation=rot, pool=pool, device=device, **env_kwargs)
    env.venv = _vec_normalize_holder()
    if log_dir is not None:     # acktr/envs.py:54-58
        env.monitor = MonitorCsv(log_dir, rank=env.env_id_base // max(env.E, 1), env_id=env_name, env_id_base=env.env_id_base,
                                 t_start=env._tstart)
    return env


class _ObRmsHolder(object):
    """Stands where the reference has VecNormalize: the training loop only reads / writes `.ob_rms` through
    utils.get_vec_normalize(envs) (main.py:77,190; acktr/utils.py:96-102); the reference disables both of
    VecNormalize's filters (acktr/envs.py:112), so there is nothing else to hold."""
    ob_rms = None


def _vec_normalize_holder():
    """utils.get_vec_normalize walks `.venv` until it finds an instance of the reference's own VecNormalize class.
    When that class is loaded in this process (the reference's training loop is what is running) the holder is an
    instance of it, created without running its constructor, so `setattr(utils.get_vec_normalize(envs), 'ob_rms', ...)`
    of the --pretrain path and the `getattr(..., 'ob_rms', None)` of the save path work on the unmodified main.py."""
    import sys
    mod = sys.modules.get("acktr.envs")
    cls = getattr(mod, "VecNormalize", None) if mod is not None else None
    if cls is not None:
        try:
            holder = cls.__new__(cls)
            holder.ob_rms = None
            return holder
        except Exception:  # noqa: BLE001 -- any oddity of a foreign class: fall back to the plain holder
            pass
    return _ObRmsHolder()
