#!/usr/bin/env python3
"""Generate tests/golden/pretrained_eval_cut2_10{,_rot}.npz by RUNNING THE UNMODIFIED REFERENCE (build container only):

    python tests/golden/make_pretrained_eval.py [--procs 7]

What the reference's test mode evaluates (main.py:26-29 -> unified_test.py:29-67 with acktr/model_loader.py:9-35): its
pretrained checkpoints on ALL 2 100 trajectories of dataset/cut_2.pt.  Here, per trajectory i:

  * env = the reference's PackingGame(test=True, data_name=dataset/cut_2.pt, enable_rotation=...) -> its own
    LoadBoxCreator (binCreator.py:42-72); `box_creator.index = i - 1; env.reset()` plays trajectory i;
  * policy = the reference's acktr.model.Policy with pretrained_models/default_cut_2.pt (rotation_cut_2.pt), loaded with
    the key rewrites of main.py:66-76; every step: the TRUE mask from acktr.utils.get_possible_position /
    get_rotation_mask on the observation (main.py:163-169), Policy.act(obs, None, None, mask, deterministic=True), one
    observation per call as model_loader.py's evaluate does; `env.step([action])` until done.

Stored per trajectory: the actions taken (int16, -1 padded), the number of steps, the terminal info's `ratio` (float64)
and `counter`, and the float64 sum of the rewards in step order.  Nothing is computed by this repository's code; the
only shim is a cache around torch.load of the dataset file (LoadBoxCreator.reset re-reads the 800 KB file per episode;
the cache hands every call its own deep copy, so the creator sees exactly what a fresh load gives it).
"""
import copy
import multiprocessing as mp
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

SIZE = (10, 10, 10)
CASES = (("pretrained_eval_cut2_10", "default_cut_2.pt", False), ("pretrained_eval_cut2_10_rot", "rotation_cut_2.pt", True))


def _worker(job):
    ckpt, rot, lo, hi = job
    import contextlib
    import io
    torch.set_num_threads(1)
    import make_golden as mg                      # installs the shims, imports the reference
    from oracle import ref_shims
    ds = os.path.join(ref_shims.REFERENCE_ROOT, "dataset", "cut_2.pt")
    real_load = torch.load
    cache = {}

    def cached_load(path, *a, **k):
        if os.path.abspath(str(path)) == os.path.abspath(ds):
            if "d" not in cache:
                cache["d"] = real_load(path, *a, **k)
            return copy.deepcopy(cache["d"])
        return real_load(path, *a, **k)

    pol = mg.pretrained_policy(os.path.join(ref_shims.REFERENCE_ROOT, "pretrained_models", ckpt), SIZE, rot)
    torch.load = cached_load
    with contextlib.redirect_stdout(io.StringIO()):
        env = mg.PackingGame(container_size=SIZE, test=True, data_name=ds, enable_rotation=rot)
    out = []
    for i in range(lo, hi):
        env.box_creator.index = i - 1
        obs = env.reset()
        acts, ret = [], 0.0
        while True:
            o = torch.FloatTensor(np.asarray(obs, np.float32)).unsqueeze(0)
            m = mg.loop_masks(o, SIZE, rot)
            a = int(pol(o, m)[0])
            obs, r, d, info = env.step([a])
            acts.append(a)
            ret += r
            if d:
                break
        out.append((i, acts, float(info["ratio"]), int(info["counter"]), float(ret)))
    return out


def main():
    procs = int(sys.argv[sys.argv.index("--procs") + 1]) if "--procs" in sys.argv else 7
    n = 2100
    for name, ckpt, rot in CASES:
        chunk = (n + procs * 4 - 1) // (procs * 4)
        jobs = [(ckpt, rot, lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]
        with mp.Pool(procs) as pool:
            res = [r for part in pool.map(_worker, jobs) for r in part]
        res.sort()
        T = max(len(r[1]) for r in res)
        actions = np.full((n, T), -1, np.int16)
        for i, acts, _, _, _ in res:
            actions[i, :len(acts)] = acts
        steps = np.array([len(r[1]) for r in res], np.int32)
        ratio = np.array([r[2] for r in res], np.float64)
        counter = np.array([r[3] for r in res], np.int32)
        ret = np.array([r[4] for r in res], np.float64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), actions=actions, steps=steps, ratio=ratio, counter=counter, ep_ret=ret,
                            size=np.array(SIZE, np.int32), rotation=np.int32(rot))
        print("%-28s %d trajectories: mean utilisation %.4f, mean items %.3f, longest episode %d steps, completely packed %d"
              % (name, n, ratio.mean(), counter.mean(), T, int((ratio == 1.0).sum())))


if __name__ == "__main__":
    main()
