#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- time the UNMODIFIED reference Python in the build container (the only place
/root/reference exists; it cannot travel to the GPU box, so bench.py's cpu_baseline uses the C oracle
instead and this script's output is committed under profiles/ for context).

  R1  reference plumbing as-is (BASELINE.json config[0]): acktr.envs.make_vec_envs -> ShmemVecEnv(fork)
      with 16 PackingGame workers, --item-seq rs, plus the parent-side per-observation mask loop of
      main.py:163-169, uniform-random-feasible policy in place of the network.
  R2  one process, PackingGame.step + acktr.utils.get_possible_position on the same CUT-2 pool the GPU
      bench uses (per-core figure).
"""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT  # the script's own directory would shadow the `oracle` package
from oracle import ref_shims  # noqa: E402

ref_shims.install()
from acktr.envs import make_vec_envs  # noqa: E402
from acktr.utils import get_possible_position, get_rotation_mask  # noqa: E402
from envs.bpp0 import PackingGame  # noqa: E402


def r1(steps=150):
    args = types.SimpleNamespace(enable_rotation=False, container_size=(10, 10, 10), data_type="rs",
                                 box_size_set=[(i, j, k) for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)])
    devnull, stdout = open(os.devnull, "w"), sys.stdout
    sys.stdout = devnull
    try:
        envs = make_vec_envs("Bpp-v0", 1, 16, 1.0, tempfile.mkdtemp(), torch.device("cpu"), False, args=args)
    finally:
        sys.stdout = stdout
    rng = np.random.RandomState(0)
    obs = envs.reset()

    def masks(obs):
        return np.array([get_possible_position(o, args.container_size) for o in obs])

    m = masks(obs)
    t_mask = t_step = 0.0
    for t in range(steps + 20):
        a = torch.tensor([[rng.choice(np.flatnonzero(r))] for r in m])
        t0 = time.perf_counter()
        obs, rew, done, infos = envs.step(a)
        t1 = time.perf_counter()
        m = masks(obs)
        t2 = time.perf_counter()
        if t >= 20:
            t_step += t1 - t0
            t_mask += t2 - t1
    envs.close()
    return {"env_steps_per_s": 16 * steps / (t_step + t_mask), "ms_envs_step": t_step / steps * 1e3,
            "ms_mask_loop": t_mask / steps * 1e3, "envs": 16, "workers": 16}


def r2(size, rotation, steps=400):
    import bpp_amd
    pool = bpp_amd.sequences.cut2_pool(size, 64, seed=0)
    seqs = [[tuple(int(v) for v in it[:3]) for it in s] for s in pool]
    env = PackingGame(box_creator=ref_shims.make_replay_creator(seqs, size), container_size=size, enable_rotation=rotation)
    rng = np.random.RandomState(1)
    obs = env.reset()
    fn = get_rotation_mask if rotation else get_possible_position
    t0 = time.perf_counter()
    for _ in range(steps):
        o = torch.from_numpy(obs.astype(np.float32))
        m = np.asarray(fn(o, size))
        obs, r, d, info = env.step(int(rng.choice(np.flatnonzero(m))))
        if d:
            obs = env.reset()
    return {"env_steps_per_s_per_core": steps / (time.perf_counter() - t0)}


def port(size, rotation, seconds=4.0):
    """oracle/ref_port.py (what bench.py times on the GPU box) on the same pool, policy and core: its speed next to R2's."""
    import bpp_amd
    from oracle import ref_port
    pool = bpp_amd.sequences.cut2_pool(size, 64, seed=0)
    rows = [[tuple(int(v) for v in it[:3]) for it in s] for s in pool]
    steps, dt = ref_port.rollout(rows, size, rotation, seconds, seed=1)
    return {"env_steps_per_s_per_core": steps / dt}


if __name__ == "__main__":
    out = {"host_cores": os.cpu_count(), "python": sys.version.split()[0], "numpy": np.__version__,
           "R1_reference_plumbing_rs_16env": r1(),
           "R2_single_core_cut2_10": r2((10, 10, 10), False),
           "R2_single_core_cut2_10_rot": r2((10, 10, 10), True),
           "R2_single_core_cut2_20": r2((20, 20, 20), False, steps=120),
           "python_port_single_core_cut2_10": port((10, 10, 10), False),
           "python_port_single_core_cut2_10_rot": port((10, 10, 10), True),
           "python_port_single_core_cut2_20": port((20, 20, 20), False)}
    for k in ("cut2_10", "cut2_10_rot", "cut2_20"):
        out["port_over_reference_" + k] = round(out["python_port_single_core_" + k]["env_steps_per_s_per_core"]
                                                / out["R2_single_core_" + k]["env_steps_per_s_per_core"], 3)
    print(json.dumps(out, indent=1))
