#!/usr/bin/env python3
"""Golden vectors for the lookahead consumers, recorded by RUNNING THE UNMODIFIED REFERENCE
(build container only):   python tests/golden/make_lookahead_golden.py

  * lookahead_branch_10.npz -- what acktr/reorder.py:245-262 and MCTS/node.py:92-137 do with the env: play a prefix,
    `copy.deepcopy(env)` it B times, step every copy with a different action (two levels deep), record each copy's
    observation / reward / done / counter / ratio and the masks acktr.utils gives for the observations.
  * windows_20_to_10.npz -- multi_bin/multi_bin.py:7-17,28-45: 10x10 windows (stride 10) of random 20x20 pallets,
    `get_possible_position(window_obs, (10,10,10))` per window.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402

ref_shims.install()

from acktr.utils import get_possible_position, get_rotation_mask  # noqa: E402
from envs.bpp0 import PackingGame  # noqa: E402


def rot_mask(obs, size):
    return np.asarray(get_rotation_mask(torch.from_numpy(np.asarray(obs, np.float32)), size))


def pos_mask(obs, size):
    return get_possible_position(torch.from_numpy(np.asarray(obs, np.float32)), size)


def branch_case():
    size, rot = (10, 10, 10), True
    g = dict(np.load(os.path.join(HERE, "rollout_cut2_10_rot.npz")))
    pool = g["pool"][:4]
    term = tuple(int(v) for v in pool[0, -1, :3])
    rng = np.random.RandomState(7)
    recs = []
    for p in range(pool.shape[0]):
        seq = [tuple(int(v) for v in it[:3]) for it in pool[p]]
        env = PackingGame(box_creator=ref_shims.make_replay_creator([seq], term), container_size=size, enable_rotation=rot)
        obs = env.reset()
        prefix = []
        for t in range(4 + p):                        # a prefix of feasible placements
            m = rot_mask(obs, size)
            a = None
            for _ in range(60):                       # a placement that does not end the episode (mask rule U vs rule S, index A)
                c = int(rng.choice(np.flatnonzero(m)))
                if not copy.deepcopy(env).step([c])[2]:
                    a = c
                    break
            if a is None:
                break                                 # nothing fits any more: the prefix ends here
            obs, r, d, info = env.step([a])
            prefix.append(a)
            assert not d
        m0 = rot_mask(obs, size)
        feas = np.flatnonzero(m0)
        level1 = [int(feas[0]), int(feas[len(feas) // 2]), int(feas[-1]), 99]      # three feasible, one likely infeasible
        out = dict(prefix=np.array(prefix, np.int64), level1=np.array(level1, np.int64), obs0=obs.astype(np.int32), mask0=m0)
        o1, r1, d1, c1, q1, m1, a2s, o2, r2, d2 = [], [], [], [], [], [], [], [], [], []
        for a in level1:
            sim = copy.deepcopy(env)                  # acktr/reorder.py:247
            ob, r, d, info = sim.step([a])
            o1.append(ob.astype(np.int32)), r1.append(np.float64(r)), d1.append(bool(d))
            c1.append(info["counter"]), q1.append(np.float64(info["ratio"]))
            mm = rot_mask(ob, size)
            m1.append(mm)
            a2 = int(np.flatnonzero(mm)[-1])
            a2s.append(a2)
            if not d:
                ob2, rr, dd, _ = sim.step([a2])       # second level of the same branch
            else:
                ob2, rr, dd = ob, 0.0, True
            o2.append(ob2.astype(np.int32)), r2.append(np.float64(rr)), d2.append(bool(dd))
        out.update(obs1=np.stack(o1), rew1=np.array(r1), done1=np.array(d1), counter1=np.array(c1), ratio1=np.array(q1),
                   mask1=np.stack(m1), level2=np.array(a2s, np.int64), obs2=np.stack(o2), rew2=np.array(r2), done2=np.array(d2))
        recs.append(out)
    flat = {"pool": pool, "size": np.array(size), "n": np.array(len(recs))}
    for p, rec in enumerate(recs):
        for k, v in rec.items():
            flat["%d_%s" % (p, k)] = v
    np.savez_compressed(os.path.join(HERE, "lookahead_branch_10.npz"), **flat)
    print("lookahead_branch_10: %d roots x 4 branches x 2 levels" % len(recs))


def sliping_window(plain, new_plain_size, stride=10):     # multi_bin/multi_bin.py:7-17
    x_np, y_np = new_plain_size[0], new_plain_size[1]
    for i in range(0, plain.shape[0] - x_np + 1, stride):
        for j in range(0, plain.shape[1] - y_np + 1, stride):
            yield plain[i:i + x_np, j:j + y_np], i, j


def windows_case():
    rng = np.random.RandomState(3)
    n, big, win = 24, (20, 20), (10, 10, 10)
    hm = rng.randint(0, 11, size=(n,) + big).astype(np.int32)
    hm[:6] = np.repeat(np.repeat(rng.randint(0, 8, size=(6, 4, 4)), 5, 1), 5, 2)     # blocky pallets: many feasible spots
    hm[6] = 0
    items = rng.randint(1, 6, size=(n, 3)).astype(np.int32)
    masks, offs = [], []
    for k in range(n):
        row = []
        for new_plain, dx, dy in sliping_window(hm[k], win):
            obs = np.zeros(400)                                                      # multi_bin/multi_bin.py:39-43
            obs[0:100] = np.reshape(new_plain, (-1,))
            obs[100:200], obs[200:300], obs[300:400] = items[k]
            row.append(np.array(pos_mask(obs, win), np.int8))
            if k == 0:
                offs.append((dx, dy))
        masks.append(np.stack(row))
    np.savez_compressed(os.path.join(HERE, "windows_20_to_10.npz"), hmap=hm, items=items, masks=np.stack(masks),
                        offsets=np.array(offs, np.int64))
    print("windows_20_to_10: %d pallets x %d windows" % (n, len(offs)))


if __name__ == "__main__":
    branch_case()
    windows_case()
