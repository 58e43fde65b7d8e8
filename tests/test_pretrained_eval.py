"""All 2 100 trajectories of the reference's dataset/cut_2.pt under the reference's OWN pretrained checkpoints (VERDICT r5
#1d).  tests/golden/pretrained_eval_cut2_10{,_rot}.npz hold, per trajectory, what the unmodified reference did --
PackingGame + LoadBoxCreator driven by acktr.model.Policy.act(deterministic=True) with the true mask, the flow of
main.py:26-29 -> unified_test.py:29-67 / model_loader.py -- : the actions it took, the terminal info's ratio and counter
and the float64 sum of its rewards (tests/golden/make_pretrained_eval.py, run once in the build container).

* CPU: the oracle and the emulated product kernels replay all 2 100 trajectories in ONE batch (bin g = trajectory g, a bin
  that has finished is left alone with BPP_ACTION_NOOP) and must end every one of them at the recorded lock-step with the
  recorded ratio / counter / return: assert_array_equal on 2 100 float64 ratios.
* GPU: the same replay on the HIP path (tile and runtime-geometry kernels), and the CONSUMER run --
  examples/evaluate_checkpoint.py: the checkpoint loaded into plain torch layers ON THE DEVICE, bpp_masked_act's mode, the
  whole test set in one batch -- whose per-trajectory results are compared with the recording (a float32 argmax computed by
  another device's GEMMs may flip on a near-tie, so: >= 97 % of the trajectories identical to the last digit, the mean
  utilisation within 0.003 of the reference's) and written to profiles/ for the README line.
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

NOOP = -2 ** 63
CASES = [("pretrained_eval_cut2_10", False, "default_cut_2.pt"), ("pretrained_eval_cut2_10_rot", True, "rotation_cut_2.pt")]


def dataset_pool():
    """tests/golden/cut2_dataset_10.npz = dataset/cut_2.pt in file order: row r = trajectory r (LoadBoxCreator's appended
    [10, 10, 10], binCreator.py:62, is the pool's padding)."""
    return load_golden("cut2_dataset_10")["pool"]


def replay(make_env, g):
    """Bin i plays trajectory i with the recorded actions; returns nothing, asserts everything."""
    pool = dataset_pool()
    n, T = g["actions"].shape
    assert n == pool.shape[0] == 2100
    env = make_env(pool, tuple(int(v) for v in g["size"]), int(g["rotation"]), n)
    env.reset()
    steps = g["steps"]
    ratio, counter, ret = np.full(n, -1.0), np.full(n, -1, np.int32), np.full(n, -1.0)
    for t in range(T):
        live = t < steps
        a = np.where(live, g["actions"][:, t].astype(np.int64), NOOP)
        o = env.step(a)
        done = o["done"].astype(bool)
        np.testing.assert_array_equal(done, live & (t == steps - 1), err_msg="episode ends at lock-step %d" % t)
        ratio[done], counter[done], ret[done] = o["ratio"][done], o["counter"][done], o["ep_ret"][done]
        np.testing.assert_array_equal(o["ep_len"][done], steps[done])
    np.testing.assert_array_equal(ratio, g["ratio"])
    np.testing.assert_array_equal(counter, g["counter"])
    np.testing.assert_array_equal(ret, g["ep_ret"])


@pytest.mark.parametrize("case,rot,ckpt", CASES)
def test_fixture_is_the_reference_evaluation(case, rot, ckpt):
    """Shape of the recording: 2 100 trajectories, every episode ends on the terminator or an infeasible item, utilisation
    in the range the paper reports for these checkpoints on CUT-2 (0.66 - 0.70 without lookahead)."""
    g = load_golden(case)
    assert g["actions"].shape[0] == 2100 and int(g["rotation"]) == int(rot)
    assert (g["steps"] == g["counter"] + 1).all()          # the failing step is counted in the length, not in the boxes
    assert 0.65 < g["ratio"].mean() < 0.72 and g["counter"].mean() > 17
    assert (g["ratio"] == 1.0).sum() >= 3                  # completely packed bins are in there


@pytest.mark.parametrize("case,rot,ckpt", CASES)
def test_oracle_replays_all_2100_trajectories(oracle, case, rot, ckpt):
    replay(lambda pool, size, r, n: oracle.OracleEnv(pool, size, r, n), load_golden(case))


@pytest.mark.parametrize("case,rot,ckpt", CASES)
def test_emulated_kernels_replay_all_2100_trajectories(emu, case, rot, ckpt):
    emu.set_knobs()
    replay(lambda pool, size, r, n: emu.EmuEnv(pool, size, r, n), load_golden(case))


@pytest.mark.parametrize("case,rot,ckpt", CASES)
def test_example_actor_is_the_reference_actor(oracle, case, rot, ckpt):
    """examples/evaluate_checkpoint.py rebuilds the reference's CNNPro actor path from plain torch layers and maps the
    checkpoint's keys onto it: on observations of real states its logits equal those of the reference's own Policy
    (acktr/model.py:265-323 + dist.linear) loaded the reference's way -- bit for bit on the CPU (same ops, same order)."""
    import sys
    import types
    import torch
    from oracle import ref_shims
    root = ref_shims.REFERENCE_ROOT if ref_shims.available() else None      # (whatever tree this process's other tests import, too)
    if root is None or not os.path.isfile(os.path.join(root, "pretrained_models", ckpt)):
        pytest.skip("no reference tree with the checkpoints here")
    ref_shims.install()
    from acktr.model import Policy
    import bpp_amd
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import evaluate_checkpoint as ev
    path = os.path.join(root, "pretrained_models", ckpt)
    M = 100 * (1 + rot)
    args = types.SimpleNamespace(channel=4, container_size=(10, 10, 10), pallet_size=10, enable_rotation=rot, hidden_size=256, device="cpu")
    state, _ = torch.load(path, map_location="cpu", weights_only=False)
    pol = Policy((400,), bpp_amd.Discrete(M), base_kwargs={"recurrent": False, "hidden_size": 256, "args": args})
    sd = {k.replace("module.", "").replace("add_bias.", "").replace("_bias", "bias"): v for k, v in state.items()}
    pol.load_state_dict({k: (v.squeeze(-1) if v.dim() <= 3 else v) for k, v in sd.items()})
    pol.eval()
    actor = ev.load_actor(path, 10, M, "cpu")
    env = oracle.OracleEnv(dataset_pool()[:64], (10, 10, 10), rot, 64)
    obs, mask = env.reset()
    for t in range(12):
        o = torch.from_numpy(obs)
        with torch.no_grad():
            _, feat, _, _ = pol.base(o, None, None)
            want = pol.dist.linear(feat)
            got = actor(o)
        assert torch.equal(got, want)
        a = want.masked_fill(torch.from_numpy(mask) == 0, -1e9).argmax(1).numpy()
        r = env.step(a)
        obs, mask = r["obs"], r["mask"]


class _GpuEnv(object):
    def __init__(self, bpp, pool, size, rot, n):
        self.env = bpp.BppVecEnv(n, size, enable_rotation=bool(rot), pool=pool)

    def reset(self):
        return self.env.reset()

    def step(self, a):
        r = self.env.step_tensors(np.asarray(a))
        return {k: getattr(r, k).cpu().numpy().reshape(-1) for k in ("done", "ratio", "counter", "ep_ret", "ep_len")}


@pytest.fixture(scope="module")
def bpp():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bpp_amd
    bpp_amd._lib.lib()
    return bpp_amd


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["tile", "rt"])
@pytest.mark.parametrize("case,rot,ckpt", CASES)
def test_hip_replays_all_2100_trajectories(bpp, case, rot, ckpt, path):
    old = bpp._lib.set_knobs(bins_per_wave=0, waves_per_group=0, xcd_remap=1, force_generic=0, legacy_fast=int(path == "rt"))
    try:
        replay(lambda pool, size, r, n: _GpuEnv(bpp, pool, size, r, n), load_golden(case))
    finally:
        bpp._lib.set_knobs(**old)


@pytest.mark.gpu
@pytest.mark.parametrize("case,rot,ckpt", CASES)
def test_hip_full_size_launch_in_deep_states_every_bin(bpp, oracle, case, rot, ckpt):
    """BASELINE config 2 / 3's launch -- 65 536 bins -- driven for 44 lock-steps by the reference's own checkpoint (the actor
    rebuilt from plain torch layers on the device, bpp_masked_act's mode), EVERY bin compared with the oracle stepping the
    same 65 536 bins with the same actions: observation, mask, reward, done, counter, ratio, episode return / length, finally
    the heightmaps.  The every-bin tests of tests/test_gpu_parity.py draw uniform-feasible actions (episodes ~8 boxes deep);
    here ~1 env-step in 6 is on a bin that already holds >= 16 boxes and bins are packed completely at full launch size."""
    import sys
    import torch
    from oracle import ref_shims
    if not ref_shims.copy_available() or not os.path.isfile(os.path.join(ref_shims.REF_COPY, "pretrained_models", ckpt)):
        pytest.skip("oracle/_ref/ without the checkpoints (python oracle/make_ref.py)")
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import evaluate_checkpoint as ev
    size, E, steps = (10, 10, 10), 65536, 44
    pool = bpp.sequences.from_dataset(os.path.join(ROOT, "tests", "golden", "cut2_dataset_10.npz"), size)
    env = bpp.BppVecEnv(E, size, enable_rotation=rot, pool=pool)
    ref = oracle.OracleEnv(pool, size, rot, E)
    actor = ev.load_actor(os.path.join(ref_shims.REF_COPY, "pretrained_models", ckpt), 10, env.action_space.n, env.device)
    obs = env.reset()
    robs, rmask = ref.reset()
    np.testing.assert_array_equal(obs.cpu().numpy(), robs)
    mask = env.location_masks
    deep = full = episodes = 0
    for t in range(steps):
        with torch.no_grad():
            logits = actor(obs).float()
        action, _ = bpp.masked_act(logits, mask, deterministic=True)
        r = env.step_tensors(action)
        o = ref.step(action.cpu().numpy().reshape(-1), copy=False)
        for k in ("obs", "mask", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(getattr(r, k).cpu().numpy().reshape(o[k].shape), o[k], err_msg="%s at lock-step %d" % (k, t))
        np.testing.assert_array_equal(r.reward.cpu().numpy()[:, 0], o["reward"], err_msg="reward at lock-step %d" % t)
        obs, mask = r.obs, r.mask
        deep += int((o["counter"] >= 16).sum())
        d = o["done"] != 0
        full += int((d & (o["ratio"] == 1.0)).sum())
        episodes += int(d.sum())
    np.testing.assert_array_equal(env.hmap.cpu().numpy(), ref.hmap)
    assert deep >= 0.08 * E * steps and full >= 50 and episodes >= 1.5 * E, (deep / float(E * steps), full, episodes)


@pytest.mark.gpu
@pytest.mark.parametrize("case,rot,ckpt", CASES)
def test_consumer_evaluation_of_the_reference_checkpoint(bpp, case, rot, ckpt):
    """examples/evaluate_checkpoint.py on the GPU box: checkpoint and dataset from oracle/_ref/ (byte-for-byte copies of the
    reference's files, oracle/make_ref.py) -- DATA files only; no reference code runs here."""
    import sys
    from oracle import ref_shims
    if not ref_shims.copy_available() or not os.path.isfile(os.path.join(ref_shims.REF_COPY, "pretrained_models", ckpt)):
        pytest.skip("oracle/_ref/ without the checkpoints (python oracle/make_ref.py)")
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import evaluate_checkpoint as ev
    g = load_golden(case)
    r = ev.evaluate(os.path.join(ref_shims.REF_COPY, "pretrained_models", ckpt), os.path.join(ref_shims.REF_COPY, "dataset", "cut_2.pt"),
                    rotation=rot)
    same = (r["ratio"] == g["ratio"]) & (r["counter"] == g["counter"])
    out = {"checkpoint": "pretrained_models/" + ckpt, "dataset": "dataset/cut_2.pt", "trajectories": int(len(same)),
           "lock_steps": int(r["lock_steps"]), "seconds": round(float(r["seconds"]), 3),
           "mean_space_utilisation": float(r["ratio"].mean()), "mean_items_packed": float(r["counter"].mean()),
           "completely_packed_bins": int((r["ratio"] == 1.0).sum()),
           "reference": {"mean_space_utilisation": float(g["ratio"].mean()), "mean_items_packed": float(g["counter"].mean()),
                         "completely_packed_bins": int((g["ratio"] == 1.0).sum()),
                         "source": "tests/golden/%s.npz (the unmodified reference env + Policy, CPU)" % case},
           "trajectories_identical_to_the_last_digit": int(same.sum()),
           "note": "replaying the reference's recorded ACTIONS gives 2100 / 2100 identical (test_hip_replays_all_2100_trajectories); "
                   "this run computes its own actions with the network on the GPU -- a float32 argmax may flip on a near-tie"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r6_eval_%s.json" % case), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    assert same.mean() >= 0.97, same.mean()
    assert abs(r["ratio"].mean() - g["ratio"].mean()) < 0.003
