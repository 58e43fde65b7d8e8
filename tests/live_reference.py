"""Test infrastructure shared by tests/test_oracle_vs_live_reference.py (CPU) and tests/test_gpu_vs_live_reference.py (GPU):
rollouts recorded NOW, in child processes, from the unmodified reference in oracle/_ref/ (oracle/make_ref.py's copies --
the only reference tree that exists on the GPU box), in the format of the committed golden fixtures.

Cases (VERDICT r3, next-round #1 ii): 64 bins x 200 lock-steps of the reference stack (PackingGame + Monitor +
DummyVecEnv + VecNormalize + VecPyTorch + the per-row acktr.utils mask loop of main.py:163-169) on the 10x10x10 bin,
10x10x10 + rotation and 20x20x20 with the bench's CUT-2 pools, plus dataset/cut_2.pt played through the reference's own
LoadBoxCreator; round 6: the same under COMPETENT policies (a lowest-top heuristic; the reference's own pretrained
checkpoints played greedily).  The recordings run side by side (one process each)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "live_cut2_10": dict(size=(10, 10, 10), rotation=False, E=64, steps=200, seed=41, p_random=0.06, pool=("cut2", 256)),
    "live_cut2_10_rot": dict(size=(10, 10, 10), rotation=True, E=64, steps=200, seed=42, p_random=0.06, pool=("cut2", 256)),
    "live_cut2_20": dict(size=(20, 20, 20), rotation=False, E=64, steps=200, seed=43, p_random=0.03, pool=("cut2", 96)),
    # 192 bins SCATTERED over a full-size job of 65 536 (first / last bin, workgroup / wave / XCD-chunk boundaries, random ones): the
    # GPU twin steps all 65 536 bins -- BASELINE config 2's launch -- and compares these against the reference's own code
    "live_cut2_10_scattered_in_65536": dict(size=(10, 10, 10), rotation=False, E=192, steps=40, seed=45, p_random=0.06, pool=("cut2", 512),
                                            env_total=65536),
    # (LoadBoxCreator.reset re-reads the whole .pt file, ~0.7 s per episode: a smaller case)
    "live_dataset_cut2": dict(size=(10, 10, 10), rotation=False, E=8, steps=60, seed=44, p_random=0.06, dataset="dataset/cut_2.pt"),
    # round 6: COMPETENT policies.  live_deep_*: the lowest-top heuristic of oracle/policies.py (2 % uniform-feasible
    # noise): 10x10x10 episodes ~18 boxes deep with ~10 % of the env-steps on bins holding >= 20 boxes and completely packed
    # bins, 20x20x20 episodes > 100 boxes deep.  live_pretrained_*: the reference's own checkpoints, greedy Policy.act with the
    # true mask, on dataset/cut_2.pt through LoadBoxCreator -- what unified_test.py evaluates (utilisation ~0.72 / ~0.76)
    "live_deep_cut2_10": dict(size=(10, 10, 10), rotation=False, E=64, steps=200, seed=46, p_random=0.02, pool=("cut2", 256), policy="lowest_top"),
    "live_deep_cut2_10_rot": dict(size=(10, 10, 10), rotation=True, E=64, steps=200, seed=47, p_random=0.02, pool=("cut2", 256), policy="lowest_top"),
    "live_deep_cut2_20": dict(size=(20, 20, 20), rotation=False, E=32, steps=260, seed=48, p_random=0.01, pool=("cut2", 96), policy="lowest_top"),
    "live_pretrained_cut2_10": dict(size=(10, 10, 10), rotation=False, E=8, steps=100, seed=49, p_random=0.0, dataset="dataset/cut_2.pt",
                                    checkpoint="pretrained_models/default_cut_2.pt"),
    "live_pretrained_cut2_10_rot": dict(size=(10, 10, 10), rotation=True, E=8, steps=100, seed=50, p_random=0.0, dataset="dataset/cut_2.pt",
                                        checkpoint="pretrained_models/rotation_cut_2.pt"),
}
DEEP = [n for n in CASES if n.startswith(("live_deep", "live_pretrained"))]


def min_episodes(name, g):
    """How many finished episodes a recording must hold at least (a real number of resets went through it)."""
    if name in DEEP:
        return 5 if name.endswith("_20") else 30
    return 200 if g["actions"].shape[1] >= 64 else 30


def check_depth(name, g):
    """The deep cases hold deep states (tests/conftest.py: depth_profile)."""
    from conftest import depth_profile
    if name not in DEEP:
        return
    deep, full, ratio = depth_profile(g)
    if name.startswith("live_deep"):
        assert deep >= (0.5 if name.endswith("_20") else 0.05), (name, deep)
        assert full >= (0 if name.endswith("_20") else 1), (name, full)
        assert ratio > 0.5, (name, ratio)
    else:
        assert ratio > 0.65 and g["counter"].max() >= 22, (name, ratio)


def scattered_ids(n, total, seed):
    """n distinct global bin ids of a job of `total` bins, ascending: the edges (first / last bin, the bins either side of every
    eighth of the job = the XCD chunks of the step kernel's workgroup remap, of a 16-bin workgroup and a 4-bin wave) + random ones."""
    fixed = {0, 1, 3, 4, 15, 16, 17, total - 1, total - 2, total - 16, total - 17}
    for k in range(1, 8):
        fixed.update((k * total // 8 - 1, k * total // 8, k * total // 8 + 1))
    rng = np.random.RandomState(seed)
    ids = set(v for v in fixed if 0 <= v < total)
    while len(ids) < n:
        ids.add(int(rng.randint(0, total)))
    return sorted(ids)[:n] if len(ids) > n else sorted(ids)


def record_all(out_dir, cases=None):
    """Start one recording process per case against oracle/_ref/; returns {name: npz path}.  Raises with the child's
    stderr if one fails."""
    import bpp_amd
    from oracle import ref_shims
    assert ref_shims.copy_available(), "oracle/_ref/ is missing (python oracle/make_ref.py in the build container)"
    env = dict(os.environ, BPP_REFERENCE_ROOT=ref_shims.REF_COPY, OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    procs = {}
    for name, c in CASES.items():
        if cases is not None and name not in cases:
            continue
        spec = dict(name=name, out_dir=out_dir, size=list(c["size"]), rotation=c["rotation"], E=c["E"], steps=c["steps"],
                    seed=c["seed"], p_random=c["p_random"], policy=c.get("policy", "uniform"))
        if "checkpoint" in c:
            spec["checkpoint"] = os.path.join(ref_shims.REF_COPY, c["checkpoint"])
        if "dataset" in c:
            spec["dataset"] = os.path.join(ref_shims.REF_COPY, c["dataset"])
        else:
            pool = bpp_amd.sequences.cut2_pool(c["size"], c["pool"][1], seed=7)
            spec["pool"] = os.path.join(out_dir, name + "_pool.npz")
            np.savez(spec["pool"], pool=pool)
            if "env_total" in c:
                spec["env_total"] = c["env_total"]
                spec["env_ids"] = scattered_ids(c["E"], c["env_total"], c["seed"])
        procs[name] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--live", json.dumps(spec)],
                                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out = {}
    for name, p in procs.items():
        so, se = p.communicate(timeout=900)
        if p.returncode != 0:
            raise RuntimeError("recording %s from oracle/_ref failed:\n%s" % (name, se[-2000:]))
        out[name] = os.path.join(out_dir, name + ".npz")
    return out
