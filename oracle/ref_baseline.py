"""TEST INFRASTRUCTURE -- the UNMODIFIED reference timed on this box's host cores (bench.py's `cpu_baseline` leg).

What runs is the reference's own Python, imported through oracle/ref_shims.py from `root` (oracle/_ref/ on the GPU box:
byte-for-byte copies made by oracle/make_ref.py; /root/reference is never read by bench.py).

  R1  reference plumbing AS IT IS (BASELINE.json config[0]): `acktr.envs.make_vec_envs('Bpp-v0', seed, 16, gamma,
      log_dir, cpu, False, args)` -> VecPyTorch(VecNormalize(ShmemVecEnv(16 forked PackingGame + Monitor workers)))
      (acktr/envs.py:77-118), --item-seq rs, driven by a loop of main.py:148-174's shape: envs.step(action), the infos
      scan, one get_possible_position per observation row in the PARENT (main.py:163-169), the mask / bad-mask lists; a
      uniform-random-feasible choice stands where actor_critic.act does.
  R2  the reference parallelised fairly: one forked worker per usable core, each stepping its own unmodified
      PackingGame (+ Monitor) on the bench's own CUT-2 pool and computing acktr.utils.get_possible_position /
      get_rotation_mask of its own observation in the worker, same uniform-feasible policy.

Both are time-bounded samples; both are baselines only.  Nothing here is imported by the product.
"""
import contextlib
import io
import os
import tempfile
import time
import types


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


def r1_plumbing(root, seconds, num_processes=16, rotation=False, size=(10, 10, 10)):
    """R1: env steps/s of the reference's own 16-worker ShmemVecEnv + parent-side mask loop."""
    import numpy as np
    import torch
    from oracle import ref_shims
    ref_shims.install(root)
    from acktr.envs import make_vec_envs
    from acktr.utils import get_possible_position, get_rotation_mask
    args = types.SimpleNamespace(enable_rotation=rotation, container_size=tuple(size), data_type="rs",
                                 box_size_set=[(i, j, k) for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)])
    torch.set_num_threads(1)                                                    # main.py:61
    device = torch.device("cpu")
    with _quiet():
        envs = make_vec_envs("Bpp-v0", 1, num_processes, 1.0, tempfile.mkdtemp(), device, False, args=args)   # main.py:63
    rng = np.random.RandomState(0)
    try:
        obs = envs.reset()                                                      # main.py:121-129
        location_masks = []
        for observation in obs:
            location_masks.append(get_rotation_mask(observation, args.container_size) if rotation
                                  else get_possible_position(observation, args.container_size))
        location_masks = torch.FloatTensor(np.array(location_masks)).to(device)
        episode_rewards, episode_ratio = [], []
        steps = t_step = t_mask = 0
        warm = 10
        t_end = None
        while True:
            action = torch.tensor([[rng.choice(np.flatnonzero(row))] for row in location_masks.numpy()])
            t0 = time.perf_counter()
            location_masks = []
            obs, reward, done, infos = envs.step(action)                        # main.py:158
            for i in range(len(infos)):                                         # main.py:159-162
                if "episode" in infos[i].keys():
                    episode_rewards.append(infos[i]["episode"]["r"])
                    episode_ratio.append(infos[i]["ratio"])
            t1 = time.perf_counter()
            for observation in obs:                                             # main.py:163-169
                if not rotation:
                    box_mask = get_possible_position(observation, args.container_size)
                else:
                    box_mask = get_rotation_mask(observation, args.container_size)
                location_masks.append(box_mask)
            location_masks = torch.FloatTensor(np.array(location_masks)).to(device)
            masks = torch.FloatTensor([[0.0] if done_ else [1.0] for done_ in done])              # main.py:172-173
            bad_masks = torch.FloatTensor([[0.0] if "bad_transition" in info.keys() else [1.0] for info in infos])
            t2 = time.perf_counter()
            if warm > 0:
                warm -= 1
                if warm == 0:
                    t_end = time.perf_counter() + seconds
                continue
            steps += 1
            t_step += t1 - t0
            t_mask += t2 - t1
            if time.perf_counter() >= t_end:
                break
    finally:
        envs.close()
    del masks, bad_masks
    return {"value": num_processes * steps / (t_step + t_mask), "unit": "env steps/s", "envs": num_processes,
            "workers": num_processes, "lock_steps": steps, "seconds": t_step + t_mask,
            "ms_per_lock_step_envs_step": t_step / steps * 1e3, "ms_per_lock_step_parent_mask_loop": t_mask / steps * 1e3,
            "episodes_finished": len(episode_rewards),
            "what": "acktr.envs.make_vec_envs('Bpp-v0', 1, %d, 1.0, log_dir, cpu, False, args) [ShmemVecEnv, fork], --item-seq rs, "
                    "loop of main.py:148-174 with a uniform-feasible choice in place of actor_critic.act" % num_processes}


def _r2_worker(job):
    root, rows, size, rotation, seconds, k, n = job
    import numpy as np
    import torch
    from oracle import ref_shims
    ref_shims.install(root)
    from acktr.utils import get_possible_position, get_rotation_mask
    from baselines import bench
    from envs.bpp0 import PackingGame
    torch.set_num_threads(1)
    with _quiet():
        cr = ref_shims.make_replay_creator(rows, size, env_id=k, env_total=n)
        env = bench.Monitor(PackingGame(box_creator=cr, container_size=size, enable_rotation=rotation), None,
                            allow_early_resets=False)
    fn = get_rotation_mask if rotation else get_possible_position
    rng = np.random.RandomState(1000 + k)
    obs = env.reset()
    steps, episodes = 0, 0
    t0 = time.perf_counter()
    while True:
        for _ in range(10):
            o = torch.from_numpy(obs.astype(np.float32))      # what the parent's loop is handed (acktr/envs.py:176,190)
            m = np.asarray(fn(o, size))
            obs, r, d, info = env.step(int(rng.choice(np.flatnonzero(m))))
            if d:                                             # shmem_vec_env.py:128-129
                obs = env.reset()
                episodes += 1
        steps += 10
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return steps, dt, episodes


def r2_workers(root, pool, size, rotation, seconds, cores):
    """R2: sum over `cores` forked workers of (env steps / own busy seconds)."""
    import multiprocessing as mp
    import numpy as np
    pool = np.asarray(pool)
    term = tuple(int(v) for v in size)
    rows = []
    for row in pool:                                          # drop the pad entries: the creator appends the terminator itself
        seq = [tuple(int(v) for v in it[:3]) for it in row]
        while seq and seq[-1] == term:
            seq.pop()
        rows.append(seq)
    jobs = [(root, rows, term, bool(rotation), seconds, k, cores) for k in range(cores)]
    with mp.get_context("fork").Pool(cores) as p:
        res = p.map(_r2_worker, jobs)
    rate = sum(s / dt for s, dt, _ in res)
    return {"value": rate, "unit": "env steps/s", "cores": cores, "per_core": rate / cores,
            "seconds": max(dt for _, dt, _ in res), "env_steps": sum(s for s, _, _ in res), "episodes_finished": sum(e for _, _, e in res),
            "what": "%d forked workers, each: unmodified PackingGame.step (+ bench.Monitor) and acktr.utils.%s on its own observation, "
                    "the bench's CUT-2 pool (replayed through a BoxCreator subclass), uniform-feasible policy"
                    % (cores, "get_rotation_mask" if rotation else "get_possible_position")}


def main():
    """python -m oracle.ref_baseline [root] [seconds]: print both baselines as JSON (run in its own process by bench.py so that
    the reference's modules never enter the bench process)."""
    import json
    import sys
    import numpy as np
    root = sys.argv[1]
    spec = json.loads(sys.argv[2])
    pool = np.load(spec["pool"])["pool"]
    out = {}
    try:
        out["R1"] = r1_plumbing(root, spec["r1_seconds"])
    except Exception as exc:  # noqa: BLE001
        out["R1"] = {"error": repr(exc)}
    try:
        out["R2"] = r2_workers(root, pool, tuple(spec["size"]), spec["rotation"], spec["r2_seconds"], spec["cores"])
    except Exception as exc:  # noqa: BLE001
        out["R2"] = {"error": repr(exc)}
    print("REF_BASELINE=" + json.dumps(out))


if __name__ == "__main__":
    sys_path0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path[0] = sys_path0
    main()
