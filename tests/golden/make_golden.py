#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE UNMODIFIED REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What is executed is the reference's own code, imported through oracle/ref_shims.py:
  * envs.bpp0.PackingGame                       (envs/bpp0/bin3D.py)   one per bin
  * baselines.bench.Monitor                     (baselines/bench/monitor.py)
  * baselines.common.vec_env.DummyVecEnv        (auto-reset, float32 obs buffer; same VecEnv contract as
                                                 the ShmemVecEnv main.py uses, acktr/envs.py:101-104)
  * acktr.envs.VecNormalize(ob=False,ret=False) and acktr.envs.VecPyTorch  (acktr/envs.py:112-113)
  * acktr.utils.get_possible_position / get_rotation_mask applied per observation row exactly as
    main.py:122-129,163-169 does
  * PackingGame.get_possible_position (rule "S", envs/bpp0/bin3D.py:72-93)
Item sequences are injected with a replaying BoxCreator (ref_shims.make_replay_creator) so that the
device env can be fed the same items.  Nothing in the fixtures is computed by this repository's code.

Every array is stored in the narrowest integer type that holds it exactly (asserted), float values
(rewards, ratios, episode returns) are stored as the reference's own float32/float64.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402

ref_shims.install()

from acktr.envs import VecNormalize, VecPyTorch  # noqa: E402
from acktr.utils import get_possible_position, get_rotation_mask  # noqa: E402
from baselines import bench  # noqa: E402
from baselines.common.vec_env.dummy_vec_env import DummyVecEnv  # noqa: E402
from envs.bpp0 import PackingGame  # noqa: E402
from envs.bpp0.mdCreator import MDlayerBoxCreator  # noqa: E402


def pad_pool(seqs, T, term):
    """[P][T][4] uint8, (x,y,z,0), padded with the terminator; last entry always a terminator."""
    P = len(seqs)
    pool = np.zeros((P, T, 4), np.uint8)
    pool[:, :, 0], pool[:, :, 1], pool[:, :, 2] = term
    for p, s in enumerate(seqs):
        assert len(s) <= T - 1, (len(s), T)
        for t, it in enumerate(s):
            pool[p, t, :3] = it
    return pool


def make_stack(pool, size, rotation, E, env_ids=None, env_total=None):
    """E reference envs; env e plays the item stream of GLOBAL bin env_ids[e] of a job of env_total bins (default: bins
    0 .. E-1 of a job of E): episode k of global bin g plays pool row (g + k * env_total) mod P (include/bpp_abi.h)."""
    term = tuple(int(v) for v in pool[0, -1, :3])
    seqs = [[tuple(int(v) for v in it[:3]) for it in s] for s in pool]
    ids = list(range(E)) if env_ids is None else [int(v) for v in env_ids]
    total = E if env_total is None else int(env_total)
    assert len(ids) == E

    def thunk(e):
        def _t():
            cr = ref_shims.make_replay_creator(seqs, term, env_id=ids[e], env_total=total)
            env = PackingGame(box_creator=cr, container_size=size, enable_rotation=rotation)
            return bench.Monitor(env, None, allow_early_resets=False)
        return _t

    dummy = DummyVecEnv([thunk(e) for e in range(E)])
    venv = VecPyTorch(VecNormalize(dummy, gamma=1.0, ob=False, ret=False), torch.device("cpu"))
    return dummy, venv


def loop_masks(obs, size, rotation):
    """main.py:122-129 / :163-169 verbatim behaviour: one call per observation row."""
    rows = []
    for observation in obs:
        if not rotation:
            rows.append(get_possible_position(observation, size))
        else:
            rows.append(get_rotation_mask(observation, size))
    return torch.FloatTensor(np.array(rows)).numpy()


def space_masks(dummy):
    return np.stack([m.env.get_possible_position().reshape(-1) for m in dummy.envs])


def exact(a, dtype):
    b = np.asarray(a).astype(dtype)
    assert np.array_equal(b.astype(np.asarray(a).dtype), a), "lossy fixture cast"
    return b


def choose_actions(policy, rng, obs, mask, size, rotation, p_random, allow_past_area):
    """The recorded rollouts' action source.  "uniform" (every fixture up to round 5): a uniform draw among the mask's
    feasible entries, with probability p_random any index (some of them out of range).  "lowest_top"
    (oracle/policies.py: lowest resulting top, then smoothest surface, then lowest index -- deep episodes, full bins),
    with probability p_random a uniform feasible entry instead (bins that play the same pool row drift apart).  A
    callable: the reference's own pretrained Policy (pretrained_policy below), greedy."""
    E, M = mask.shape
    A = size[0] * size[1]
    if policy == "uniform":
        a = np.zeros(E, np.int64)
        for e in range(E):
            if rng.rand() < p_random:
                a[e] = rng.randint(0, M + 2) - 1 if rng.rand() < 0.1 else rng.randint(0, M)
                if not allow_past_area:
                    a[e] = min(max(a[e], -1), A)   # reference asserts for idx > A without rotation
            else:
                a[e] = rng.choice(np.flatnonzero(mask[e]))
        return a
    if policy == "lowest_top":
        from oracle.policies import lowest_top_actions
        a = lowest_top_actions(obs.numpy(), mask, size, rotation)
    else:
        a = policy(obs, mask)
    for e in range(E):
        if rng.rand() < p_random:
            a[e] = rng.choice(np.flatnonzero(mask[e]))
    return a


def pretrained_policy(path, size, rotation):
    """The reference's checkpoint loaded the way main.py:66-76 / acktr/model_loader.py:19-35 load it (`module.` /
    `add_bias.` / `_bias` key rewrites, trailing-1 squeeze) into the reference's own acktr.model.Policy; returns
    f(obs [E, 4A] tensor, mask [E, M] array) -> int64 [E]: Policy.act(..., deterministic=True) with the true mask
    (main.py:150-153 with dist.mode()).  Runs on the CPU; the actions are RECORDED, so no cross-machine float claim."""
    import types
    from acktr.model import Policy
    import bpp_amd
    A = size[0] * size[1]
    args = types.SimpleNamespace(channel=4, container_size=tuple(size), pallet_size=size[0], enable_rotation=bool(rotation),
                                 hidden_size=256, device="cpu")
    model_pretrained, ob_rms = torch.load(path, map_location="cpu", weights_only=False)
    assert ob_rms is None
    ac = Policy((4 * A,), bpp_amd.spaces.Discrete(A * (1 + int(bool(rotation)))),
                base_kwargs={'recurrent': False, 'hidden_size': args.hidden_size, 'args': args})
    load_dict = {k.replace('module.', ''): v for k, v in model_pretrained.items()}
    load_dict = {k.replace('add_bias.', ''): v for k, v in load_dict.items()}
    load_dict = {k.replace('_bias', 'bias'): v for k, v in load_dict.items()}
    for k, v in load_dict.items():
        if len(v.size()) <= 3:
            load_dict[k] = v.squeeze(dim=-1)
    ac.load_state_dict(load_dict)
    ac.eval()

    def act(obs, mask):
        with torch.no_grad():
            _, action, _, _ = ac.act(obs, None, None, torch.FloatTensor(mask), deterministic=True)
        return action.reshape(-1).numpy().astype(np.int64)
    return act


def rollout_case(name, pool, size, rotation, E, steps, seed, p_random, out_dir=None, env_ids=None, env_total=None, policy="uniform"):
    """One recorded rollout of the reference stack -> <out_dir or tests/golden>/<name>.npz.  env_ids / env_total: the E
    recorded envs are the global bins env_ids of a job of env_total bins (stored in the file as `env_ids`, `env_total`)."""
    W, L, H = size
    A = W * L
    M = A * (1 + rotation)
    dummy, venv = make_stack(pool, size, rotation, E, env_ids, env_total)
    rng = np.random.RandomState(seed)
    obs = venv.reset()
    mask = loop_masks(obs, size, rotation)
    rec = dict(obs0=exact(obs.numpy(), np.uint8), mask0=exact(mask, np.uint8), smask0=exact(space_masks(dummy), np.uint8))
    acts, obss, masks, smasks, rews, rews64, dones, counters, ratios, ep_r, ep_l, ep_r_raw = ([] for _ in range(12))
    for _ in range(steps):
        a = choose_actions(policy, rng, obs, mask, size, rotation, p_random, bool(rotation))
        obs, reward, done, infos = venv.step(torch.from_numpy(a).unsqueeze(1))
        mask = loop_masks(obs, size, rotation)
        acts.append(a)
        obss.append(exact(obs.numpy(), np.uint8))
        masks.append(exact(mask, np.uint8))
        smasks.append(exact(space_masks(dummy), np.uint8))
        assert reward.dtype == torch.float32 and tuple(reward.shape) == (E, 1)
        rews.append(reward.numpy()[:, 0].copy())
        dones.append(np.asarray(done).astype(np.uint8))
        counters.append(np.array([i["counter"] for i in infos], np.int32))
        ratios.append(np.array([float(i["ratio"]) for i in infos], np.float64))
        ep_r.append(np.array([i["episode"]["r"] if "episode" in i else np.nan for i in infos], np.float64))
        ep_l.append(np.array([i["episode"]["l"] if "episode" in i else -1 for i in infos], np.int32))
        raw = np.full(E, np.nan, np.float64)
        for e, i in enumerate(infos):
            assert ("episode" in i) == bool(done[e])
            if done[e]:
                assert np.array_equal(i["mask"], np.ones(M))
                raw[e] = dummy.envs[e].episode_rewards[-1]   # sum(self.rewards) before round(.,6)
                assert round(raw[e], 6) == i["episode"]["r"]
        ep_r_raw.append(raw)
    rec.update(actions=np.stack(acts), obs=np.stack(obss), mask=np.stack(masks), smask=np.stack(smasks),
               reward=np.stack(rews), done=np.stack(dones), counter=np.stack(counters), ratio=np.stack(ratios),
               ep_r=np.stack(ep_r), ep_r_raw=np.stack(ep_r_raw), ep_l=np.stack(ep_l), pool=pool,
               size=np.array(size, np.int32), rotation=np.int32(rotation))
    if env_ids is not None:
        rec.update(env_ids=np.asarray(env_ids, np.int64), env_total=np.int64(env_total))
    path = os.path.join(out_dir or HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    nd = int(rec["done"].sum())
    print("%-22s E=%d steps=%d episodes=%d  mean mask density %.3f  -> %s (%d KB)" % (
        name, E, steps, nd, rec["mask"].mean(), os.path.basename(path), os.path.getsize(path) // 1024))


def cut2_reference_sequences(size, n, seed0):
    """CUT-2 sequences from the reference generator itself (envs/bpp0/mdCreator.py:147-166) under
    random.seed(seed0 + k); the stored trailing [10,10,10] (mdCreator.py:161) is dropped -- the pool's
    pad terminator (W,L,H) takes its place."""
    out = []
    devnull = open(os.devnull, "w")
    stdout = sys.stdout
    sys.stdout = devnull
    try:
        cr = MDlayerBoxCreator(size, [2, 5])
        for k in range(n):
            random.seed(seed0 + k)
            cr.reset()
            out.append([tuple(b) for b in cr.box_set[:-1]])
    finally:
        sys.stdout = stdout
    return out


def mask_case(name, size, n, seed, lo, hi):
    """Stand-alone mask vectors on random (not necessarily reachable) heightmaps."""
    W, L, H = size
    A = W * L
    rng = np.random.RandomState(seed)
    env = PackingGame(box_creator=ref_shims.make_replay_creator([[(1, 1, 1)]], (W, L, H)), container_size=size)
    env.reset()
    hmaps = np.zeros((n, A), np.int32)
    items = np.zeros((n, 3), np.int32)
    m_u = np.zeros((n, A), np.uint8)
    m_ur = np.zeros((n, 2 * A), np.uint8)
    m_s = np.zeros((n, A), np.uint8)
    for k in range(n):
        kind = k % 4
        if kind == 0:      # plateaus: few distinct levels, feasible positions likely
            h = np.zeros((W, L), np.int32)
            for _ in range(rng.randint(0, 6)):
                x0, y0 = rng.randint(0, W), rng.randint(0, L)
                x1, y1 = rng.randint(x0, W) + 1, rng.randint(y0, L) + 1
                h[x0:x1, y0:y1] = rng.randint(0, H + 1)
        elif kind == 1:    # noise on 2 levels
            h = rng.randint(0, 2, size=(W, L)).astype(np.int32) + rng.randint(0, H)
        elif kind == 2:    # almost flat with a few dents/spikes
            h = np.full((W, L), rng.randint(0, H), np.int32)
            for _ in range(rng.randint(1, 4)):
                h[rng.randint(0, W), rng.randint(0, L)] = rng.randint(0, H + 1)
        else:              # uniform noise
            h = rng.randint(0, H + 1, size=(W, L)).astype(np.int32)
        it = (rng.randint(lo, hi + 1), rng.randint(lo, hi + 1), rng.randint(lo, hi + 1))
        if k % 5 == 4:     # rules U and S diverge: >95% of a >=41-cell window on the max
            x, y = 7, rng.randint(6, 8)  # level, but two of its corners dented
            it = (x, y, rng.randint(1, 3))
            lvl = rng.randint(1, H - 2)
            h = np.full((W, L), lvl, np.int32)
            i0, j0 = rng.randint(0, W - x + 1), rng.randint(0, L - y + 1)
            cs = [(i0, j0), (i0 + x - 1, j0), (i0, j0 + y - 1), (i0 + x - 1, j0 + y - 1)]
            for c in rng.permutation(4)[:2]:
                h[cs[c]] = rng.randint(0, lvl)
        if k % 17 == 0:
            it = (W, L, H)            # the terminator item
        if k % 23 == 0:
            it = (W + 1, 2, 2)        # wider than the bin: empty candidate range -> all-ones
        hmaps[k] = h.reshape(-1)
        items[k] = it
        obs = np.concatenate([h.reshape(-1), np.full(A, it[0]), np.full(A, it[1]), np.full(A, it[2])]).astype(np.float32)
        m_u[k] = np.array(get_possible_position(obs, size), np.uint8)
        m_ur[k] = exact(get_rotation_mask(torch.from_numpy(obs), size), np.uint8)
        env.box_creator.box_list[0] = it
        m_s[k] = exact(env.get_possible_position(plain=h).reshape(-1), np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, hmap=hmaps, items=items, mask_utils=m_u, mask_utils_rot=m_ur, mask_space=m_s,
                        size=np.array(size, np.int32))
    print("%-22s n=%d  U!=S rows: %d -> %s (%d KB)" % (name, n, int((m_u != m_s).any(1).sum()),
                                                      os.path.basename(path), os.path.getsize(path) // 1024))


def dataset_fixture():
    """The reference's own CUT-2 item sequences for the 10x10x10 bin (dataset/cut_2.pt, 2100 python
    lists of [x,y,z] each ending with a stored [10,10,10]) repacked as the pool format."""
    d = torch.load(os.path.join(ref_shims.REFERENCE_ROOT, "dataset", "cut_2.pt"))
    seqs = []
    for s in d:
        assert list(s[-1]) == [10, 10, 10]
        seqs.append([tuple(it) for it in s[:-1]])
    T = max(len(s) for s in seqs) + 1
    pool = pad_pool(seqs, T, (10, 10, 10))
    np.savez_compressed(os.path.join(HERE, "cut2_dataset_10.npz"), pool=pool)
    print("cut2_dataset_10        P=%d T=%d" % pool.shape[:2])
    return pool


def main():
    pool10 = dataset_fixture()
    rollout_case("rollout_cut2_10", pool10[:96], (10, 10, 10), False, E=8, steps=320, seed=1, p_random=0.08)
    rollout_case("rollout_cut2_10_rot", pool10[96:192], (10, 10, 10), True, E=8, steps=320, seed=2, p_random=0.08)
    seq20 = cut2_reference_sequences((20, 20, 20), 12, seed0=100)
    pool20 = pad_pool(seq20, max(len(s) for s in seq20) + 1, (20, 20, 20))
    rollout_case("rollout_cut2_20", pool20, (20, 20, 20), False, E=4, steps=360, seed=3, p_random=0.03)
    rng = np.random.RandomState(7)
    rs = [[tuple(rng.randint(2, 6, size=3)) for _ in range(127)] for _ in range(24)]
    rollout_case("rollout_rs_10", pad_pool(rs, 128, (10, 10, 10)), (10, 10, 10), False, E=6, steps=300, seed=4,
                 p_random=0.05)
    wide = [[tuple(rng.randint(1, 8, size=3)) for _ in range(95)] for _ in range(24)]
    rollout_case("rollout_wide_8x12x9_rot", pad_pool(wide, 96, (8, 12, 9)), (8, 12, 9), True, E=6, steps=300, seed=5,
                 p_random=0.05)
    # short pool whose pad entry is a placeable item: cursors run past T and keep returning it
    short = [[tuple(rng.randint(1, 4, size=3)) for _ in range(5)] for _ in range(3)]
    rollout_case("rollout_short_5x4x6", pad_pool(short, 6, (2, 2, 1)), (5, 4, 6), True, E=5, steps=160, seed=6,
                 p_random=0.1)
    mask_case("masks_10", (10, 10, 10), 400, 11, 1, 7)
    mask_case("masks_20", (20, 20, 20), 120, 12, 1, 9)
    mask_case("masks_7x13x8", (7, 13, 8), 200, 13, 1, 8)
    deep()


def deep():
    """Round 6: fixtures whose states a COMPETENT policy reaches -- `python make_golden.py --deep` makes
    only these.  rollout_deep_*: the lowest-top heuristic (oracle/policies.py) with 2 % uniform-feasible noise on rows of
    reference-generated CUT-2 sequences (MDlayerBoxCreator, 10x10x10 and 20x20x20); rollout_pretrained_*: the
    reference's own checkpoints (pretrained_models/default_cut_2.pt, rotation_cut_2.pt) played greedily through the
    reference's Policy.act with the true mask on dataset/cut_2.pt through LoadBoxCreator (main.py:26-29 ->
    unified_test.py:29-67 evaluate these checkpoints on this file)."""
    seq10 = cut2_reference_sequences((10, 10, 10), 192, seed0=300)       # MDlayerBoxCreator under random.seed(300 + k)
    gen10 = pad_pool(seq10, max(len(s) for s in seq10) + 1, (10, 10, 10))
    rollout_case("rollout_deep_cut2_10", gen10[:96], (10, 10, 10), False, E=8, steps=240, seed=21, p_random=0.02, policy="lowest_top")
    rollout_case("rollout_deep_cut2_10_rot", gen10[96:], (10, 10, 10), True, E=8, steps=240, seed=22, p_random=0.02, policy="lowest_top")
    seq20 = cut2_reference_sequences((20, 20, 20), 12, seed0=200)
    pool20 = pad_pool(seq20, max(len(s) for s in seq20) + 1, (20, 20, 20))
    rollout_case("rollout_deep_cut2_20", pool20, (20, 20, 20), False, E=4, steps=420, seed=23, p_random=0.01, policy="lowest_top")
    ds = os.path.join(ref_shims.REFERENCE_ROOT, "dataset", "cut_2.pt")
    for name, ckpt, rot in (("rollout_pretrained_cut2_10", "default_cut_2.pt", False), ("rollout_pretrained_cut2_10_rot", "rotation_cut_2.pt", True)):
        pol = pretrained_policy(os.path.join(ref_shims.REFERENCE_ROOT, "pretrained_models", ckpt), (10, 10, 10), rot)
        dataset_case(name, ds, (10, 10, 10), E=8, steps=120, seed=24 + rot, p_random=0.0, out_dir=HERE, rotation=rot, policy=pol)


def live_case(spec_json):
    """`make_golden.py --live '<json>'`: ONE rollout recorded NOW from whatever reference tree ref_shims resolves
    (BPP_REFERENCE_ROOT; the GPU suite points it at oracle/_ref/), written to spec["out_dir"] -- the same format as the
    committed fixtures, so the same checker replays it on the HIP path (tests/test_gpu_vs_live_reference.py).
    spec: name, out_dir, size, rotation, E, steps, seed, p_random and either pool (npz path) or dataset (a
    reference dataset/*.pt played through the reference's own LoadBoxCreator, binCreator.py:42-72); optional policy
    ("uniform" | "lowest_top") or checkpoint (a reference pretrained_models/*.pt, played greedily)."""
    import json
    spec = json.loads(spec_json)
    size = tuple(spec["size"])
    policy = spec.get("policy", "uniform")
    if spec.get("checkpoint"):
        policy = pretrained_policy(spec["checkpoint"], size, bool(spec["rotation"]))
    if spec.get("dataset"):
        dataset_case(spec["name"], spec["dataset"], size, spec["E"], spec["steps"], spec["seed"], spec["p_random"], spec["out_dir"],
                     rotation=bool(spec["rotation"]), policy=policy)
        return
    pool = np.load(spec["pool"])["pool"]
    rollout_case(spec["name"], pool, size, bool(spec["rotation"]), spec["E"], spec["steps"], spec["seed"], spec["p_random"],
                 out_dir=spec["out_dir"], env_ids=spec.get("env_ids"), env_total=spec.get("env_total"), policy=policy)


def dataset_case(name, path, size, E, steps, seed, p_random, out_dir, rotation=False, policy="uniform"):
    """E reference envs built the way `--load-dataset` builds them -- PackingGame(test=True, data_name=path): the
    reference's LoadBoxCreator, unmodified -- whose creators start at trajectory index e (attribute set from outside;
    LoadBoxCreator.reset pre-increments, so episode k of env e plays trajectory e + k + 1).  Recorded like
    rollout_case; `pool` in the file holds sequences.from_dataset's rows of the same file.  policy: "uniform",
    "lowest_top" or a callable (pretrained_policy: the reference's own checkpoint, greedy, on its own test set -- the
    states unified_test.py:29-67 / model_loader.py evaluate the paper's numbers on)."""
    import contextlib
    import io
    import bpp_amd
    from envs.bpp0.binCreator import LoadBoxCreator

    def thunk(e):
        def _t():
            with contextlib.redirect_stdout(io.StringIO()):
                env = PackingGame(container_size=size, test=True, data_name=path, enable_rotation=bool(rotation))
            assert isinstance(env.box_creator, LoadBoxCreator)
            env.box_creator.index = e
            return bench.Monitor(env, None, allow_early_resets=False)
        return _t

    dummy = DummyVecEnv([thunk(e) for e in range(E)])
    venv = VecPyTorch(VecNormalize(dummy, gamma=1.0, ob=False, ret=False), torch.device("cpu"))
    rng = np.random.RandomState(seed)
    A = size[0] * size[1]
    obs = venv.reset()
    mask = loop_masks(obs, size, rotation)
    rec = dict(obs0=exact(obs.numpy(), np.uint8), mask0=exact(mask, np.uint8), smask0=exact(space_masks(dummy), np.uint8))
    keys = ("actions", "obs", "mask", "smask", "reward", "done", "counter", "ratio", "ep_r", "ep_l", "ep_r_raw")
    cols = {k: [] for k in keys}
    for _ in range(steps):
        if policy == "uniform":
            a = np.array([rng.randint(0, A) if rng.rand() < p_random else rng.choice(np.flatnonzero(mask[e])) for e in range(E)], np.int64)
        else:
            a = choose_actions(policy, rng, obs, mask, size, rotation, p_random, bool(rotation))
        obs, reward, done, infos = venv.step(torch.from_numpy(a).unsqueeze(1))
        mask = loop_masks(obs, size, rotation)
        cols["actions"].append(a)
        cols["obs"].append(exact(obs.numpy(), np.uint8))
        cols["mask"].append(exact(mask, np.uint8))
        cols["smask"].append(exact(space_masks(dummy), np.uint8))
        cols["reward"].append(reward.numpy()[:, 0].copy())
        cols["done"].append(np.asarray(done).astype(np.uint8))
        cols["counter"].append(np.array([i["counter"] for i in infos], np.int32))
        cols["ratio"].append(np.array([float(i["ratio"]) for i in infos], np.float64))
        cols["ep_r"].append(np.array([i["episode"]["r"] if "episode" in i else np.nan for i in infos], np.float64))
        cols["ep_l"].append(np.array([i["episode"]["l"] if "episode" in i else -1 for i in infos], np.int32))
        cols["ep_r_raw"].append(np.array([dummy.envs[e].episode_rewards[-1] if done[e] else np.nan for e in range(E)], np.float64))
    rec.update({k: np.stack(v) for k, v in cols.items()})
    # the device env's assignment rule is "episode k of bin g plays pool row (g + k * E) mod P" (include/bpp_abi.h); the
    # creators above play trajectory g + k + 1 = from_dataset's row g + k: lay those rows out under that rule
    ds = bpp_amd.sequences.from_dataset(path, size)
    K = steps + 1                                   # an episode takes at least one lock-step
    pool = np.empty((E * K,) + ds.shape[1:], np.uint8)
    for k in range(K):
        pool[k * E:(k + 1) * E] = ds[(np.arange(E) + k) % ds.shape[0]]
    rec.update(pool=pool, size=np.array(size, np.int32), rotation=np.int32(bool(rotation)))
    path_out = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path_out, **rec)
    d = rec["done"].astype(bool)
    print("%-22s E=%d steps=%d episodes=%d  mean final ratio %.3f  -> %s (%d KB)" % (
        name, E, steps, int(d.sum()), float(rec["ratio"][d].mean()) if d.any() else 0.0, os.path.basename(path_out),
        os.path.getsize(path_out) // 1024))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--live":
        live_case(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "--deep":
        deep()
    else:
        main()
