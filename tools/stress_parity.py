#!/usr/bin/env python3
"""Randomised long-running parity stress (GPU box): random geometry / bin count / rotation / pool / knobs,
hundreds of lock-steps through the native driver, everything compared bit for bit with the oracle.
    python tools/stress_parity.py --trials 200 --seed 1
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bpp_amd
from oracle import oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=100)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rng = np.random.RandomState(args.seed)
orc.build()
paths = {"fast": 0, "generic": 0}
for trial in range(args.trials):
    W, L = (int(rng.randint(1, 25)) for _ in range(2))
    if rng.rand() < 0.4:
        W, L = rng.choice([10, 20]), rng.choice([10, 20])
    if W * L > 1024:
        continue
    H = int(rng.randint(2, 30))
    size, rot = (int(W), int(L), H), bool(rng.rand() < 0.5)
    E = int(rng.choice([1, 3, 17, 64, 255, 1024, 4099]))
    hi = max(1, min(W, L, H) // 2 + 1)
    seqs = [[tuple(rng.randint(1, hi + 1, size=3)) for _ in range(rng.randint(1, 40))] for _ in range(rng.randint(1, 12))]
    if rng.rand() < 0.3:
        seqs[0][0] = size[:2] + (1,)
    pool = bpp_amd.sequences.pad_pool(seqs, size if rng.rand() < 0.7 else (1, 1, 1))
    knobs = dict(bins_per_wave=int(rng.choice([0, 1, 2, 4, 8])), waves_per_group=int(rng.choice([0, 1, 2, 4, 8])),
                 force_generic=int(rng.choice([0, 0, 1])), xcd_remap=int(rng.choice([0, 1])))
    bpp_amd._lib.set_knobs(**knobs)
    base, total = int(rng.randint(0, 50)), None
    total = base + E + int(rng.randint(0, 9))
    rule = "space" if (rng.rand() < 0.3) else "utils"
    try:
        env = bpp_amd.BppVecEnv(E, size, enable_rotation=rot, pool=pool, env_id_base=base, env_id_total=total, mask_rule=rule)
        env.reset()
        n = int(rng.randint(20, 200))
        r = env.rollout_uniform(seed=trial, step0=5, nsteps=n)
    except RuntimeError as exc:
        if "too large" in str(exc):
            continue
        raise
    ref = orc.OracleEnv(pool, size, rot, E, env_id_base=base, env_id_total=total, mask_rule=1 if rule == "space" else 0)
    ref.reset()
    o, _ = orc.rollout_uniform(ref, trial, 5, n)
    for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
        assert np.array_equal(getattr(r, k).cpu().numpy(), o[k]), (trial, size, rot, E, k, knobs)
    assert np.array_equal(r.reward.cpu().numpy()[:, 0], o["reward"])
    assert np.array_equal(env.hmap.cpu().numpy(), ref.hmap)
    assert np.array_equal(env.ep_acc.cpu().numpy(), ref.ep_acc), (trial, "per-bin episode accumulators")
    assert np.array_equal(env.episode_stats().cpu().numpy(), ref.episode_stats()), (trial, "episode statistics")
    st = env.state_numpy()
    for f in st.dtype.names:
        if f != "pad":
            assert np.array_equal(st[f], ref.state[f]), (trial, f)
    paths["generic" if (knobs["force_generic"] or (W * L) % 4 or H > 22) else "fast"] += 1
print("stress parity ok:", args.trials, "trials,", paths)
