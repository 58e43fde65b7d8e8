"""bench.py as the driver runs it: the JSON contract, the self-launch of N ranks when no launcher started it
(`python bench.py --gpus N`), and -- CPU only -- the command line a self-launch builds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "BPP_BENCH_CHILD")}
    env.update(extra)
    return env


def _communicate(cmd, env, timeout=600):
    """Run `cmd` in a process group of its own; on a timeout the WHOLE group is killed (bench.py starts a launcher that starts the
    ranks: killing only the child would leave grandchildren holding the pipes, and the test would wait for them for ever)."""
    import signal
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError("%r did not finish in %d s\n%s" % (cmd[-8:], timeout, err[-2000:]))
    return p.returncode, out, err


def _run(args, **extra):
    rc, out, err = _communicate([sys.executable, os.path.join(ROOT, "bench.py")] + args, _clean_env(**extra))
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_self_launch_command_line(monkeypatch):
    """No GPU needed: --gpus 3 without WORLD_SIZE re-executes under torch.distributed.run with one process per GPU on
    127.0.0.1, passing the original arguments through."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    def fake_call(cmd, env=None):      # noqa: F811 -- also looks at what the parent hands to rank 0
        seen["cmd"], seen["env"] = cmd, env
        seen["handed"] = json.load(open(env["BPP_BENCH_CPU_BASELINE_FILE"]))
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(bench, "cpu_baseline", lambda pool, size, rot, seconds: {"value": 1.0, "unit": "env steps/s", "cores": 1,
                                                                                 "kind": "port", "sample": "stub"})
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "20", "--warmup", "5"])
    for k in ("WORLD_SIZE", "RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                       # the launcher's exit code is handed back
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "3" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "3", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["BPP_BENCH_CHILD"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the reference is timed ONCE, by the parent, before the ranks exist, and handed to rank 0
    assert seen["handed"]["sample"] == "stub" and "before the 3 ranks" in seen["handed"]["timed_by"]
    assert not os.path.exists(seen["env"]["BPP_BENCH_CPU_BASELINE_FILE"])          # ... and the hand-over file is gone afterwards


@pytest.mark.gpu
def test_gpu_bench_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (no torchrun): it starts its two ranks itself; on this 1-GPU box both ranks share
    device 0 (BPP_BENCH_ONE_DEVICE=1 -> gloo for the barrier and the 32-byte statistics all-reduce)."""
    d = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--only-headline"], BPP_BENCH_ONE_DEVICE="1")
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 131072 and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["launcher"] == "self-launched torch.distributed.run"
    assert d["value"] > 1e8 and d["scaling"] == "weak"
    _check_multi_rank_line(d, 2)
    assert d["timed_gpu_work_ms"] >= 150.0
    assert d["config"]["episodes_finished"] > 0
    assert d["parity"]["mismatches"] == 0 and d["parity"]["checked_bins"] == 512      # every rank gates its own shard


@pytest.mark.gpu
def test_gpu_bench_eight_ranks_on_one_device_shard_by_global_id():
    """SURVEY 8e without an 8-GPU box: `python bench.py --gpus 8 --envs 8192` self-launched, all eight ranks on device 0
    over gloo.  Rank r owns the global bins from r * E (rank 7: 7 * 8192), the job has 65 536 bins, every rank's parity gate
    ran on its own shard, and the all-reduced statistics record is the sum of the eight shards' fixed-order reductions
    (counts exactly; the float sums up to the association of the seven final adds)."""
    import numpy as np
    d = _run(["--gpus", "8", "--envs", "8192", "--steps", "20", "--warmup", "5", "--gpu-seconds", "0.3", "--no-past-l3"],
             BPP_BENCH_ONE_DEVICE="1")
    assert d["n_gpus"] == 8 and d["config"]["total_envs"] == 65536 and d["config"]["envs_per_gpu"] == 8192
    assert "8 rank(s)" in d["config"]["sharding"] and "gloo" in d["config"]["sharding"]
    shards = d["config"]["shards"]
    assert [s["rank"] for s in shards] == list(range(8))
    assert [s["env_id_base"] for s in shards] == [r * 8192 for r in range(8)] and shards[7]["env_id_base"] == 7 * 8192
    parts, whole = np.array([s["sums"] for s in shards]), np.array(d["config"]["episode_sums"])
    assert (parts[:, 3] > 0).all()                                       # every shard finished episodes of its own
    np.testing.assert_array_equal(parts[:, 2:].sum(axis=0), whole[2:])   # lengths and episode counts: integers, exact
    np.testing.assert_allclose(parts[:, :2].sum(axis=0), whole[:2], rtol=1e-13, atol=0)
    assert d["config"]["episodes_finished"] == int(whole[3])
    assert d["parity"]["mismatches"] == 0 and d["parity"]["checked_bins"] == 8 * 256 and d["parity"]["per_workload"]["10x10x10"]["ranks"] == 8
    assert d["value"] > 1e7      # (eight processes taking turns on one device: ~6e7 measured)
    _check_multi_rank_line(d, 8)
    assert "self-launching parent" in d["cpu_baseline"]["timed_by"]


def _check_multi_rank_line(d, n):
    """an N > 1 line is complete by itself -- the reference timed on the same box in the same run
    (`cpu_baseline`, kind "reference" where oracle/_ref/ travelled), the roofline block of one GPU, and the same
    workload on rank 0 ALONE in the same run (`n1_value_same_run`), from which the scaling efficiency follows."""
    c = d["cpu_baseline"]
    assert c["value"] and c["value"] > 1e3 and c["cores"] >= 1 and c["kind"] in ("reference", "port") and c["sample"]
    assert "timed_by" in c
    r = d["roofline"]
    assert r["per_gpu"] is True and r["bound"] == "hbm" and 0.0 < r["frac"] < 1.05 and r["bytes_per_env_step"] == 2864
    assert d["n1_value_same_run"] > 1e7 and d["n1_same_run"]["reps"] >= 3
    assert abs(d["scaling_efficiency_same_run"] - d["value"] / (n * d["n1_value_same_run"])) < 1e-9
    # ranks SHARING one device take turns on it: the job cannot be faster than the device alone (eight processes with 8 192 bins each: ~0.08 of it)
    assert 0.02 < d["value"] / d["n1_value_same_run"] < 1.3


@pytest.mark.gpu
def test_gpu_bench_launcher_started_ranks_time_the_reference_on_rank0():
    """What the driver does for N > 1: `python -m torch.distributed.run ... bench.py --gpus 2` -- no self-launching parent,
    so rank 0 itself times the reference before any rank touches its GPU, and the line says so."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(bench.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--only-headline", "--gpu-seconds", "0.5", "--cpu-seconds", "8"]
    rc, out, err = _communicate(cmd, _clean_env(BPP_BENCH_ONE_DEVICE="1", OMP_NUM_THREADS="1"))
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["launcher"] == "torch.distributed.run"
    _check_multi_rank_line(d, 2)
    assert "rank 0" in d["cpu_baseline"]["timed_by"]


@pytest.mark.gpu
def test_gpu_bench_spawned_single_rank_equals_direct_run():
    """N = 1 through the self-launch path gives the same line as the direct run (same workload, value within noise),
    and the line carries both roofline fractions (L3-assisted and past the Infinity Cache)."""
    a = _run(["--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--only-headline"])
    b = _run(["--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--only-headline", "--launcher", "spawn"])
    assert a["config"]["launcher"] == "direct" and b["config"]["launcher"] == "self-launched torch.distributed.run"
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["config"]["total_envs"] == b["config"]["total_envs"] == 65536
    assert abs(a["value"] / b["value"] - 1.0) < 0.1
    for d in (a, b):
        r = d["roofline"]
        assert 0.3 < r["frac_past_l3"] <= r["frac"] * 1.05 < 1.05
        assert d["value_past_l3"] <= d["value"] * 1.05
        assert d["past_l3"]["output_span_MB"] > 1000 and d["timed_gpu_work_ms"] >= 200.0
        assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        assert "configs" not in d and d["parity"]["mismatches"] == 0


@pytest.mark.gpu
def test_gpu_bench_default_line_carries_every_single_gpu_config_and_the_parity_gate():
    """The line the driver runs (`python bench.py --gpus 1 --steps 20 --warmup 5`, here with shorter legs and without the
    CPU baseline): value = BASELINE config 2; `configs` holds config 3 (rotation), config 4 (20x20x20, 32 768 bins) and
    config 2 on the reference's own dataset pool, each with value, value_past_l3 and a roofline block; `parity` is the
    in-run gate against the oracle (0 mismatches or no line at all); `epsilon_variant` is SURVEY 8d's failure-path leg."""
    d = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--gpu-seconds", "0.6", "--extra-seconds", "0.3",
              "--eps-seconds", "0.2"])
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"] and d["config"]["envs_per_gpu"] == 65536
    assert set(d["configs"]) == {"10x10x10_rot", "20x20x20", "10x10x10_dataset_cut2"}
    want = {"10x10x10_rot": (3264, 65536), "20x20x20": (11264, 32768), "10x10x10_dataset_cut2": (2864, 65536)}
    for name, c in d["configs"].items():
        r = c["roofline"]
        assert r["bytes_per_env_step"] == want[name][0] and ("%d envs" % want[name][1]) in c["workload"]
        assert c["value"] > 1e8 and c["value_past_l3"] > 1e8 and 0.3 < r["frac_past_l3"] < 1.0 and 0.3 < r["frac"] < 1.05
        assert abs(r["achieved"] - r["bytes_per_env_step"] * want[name][1] / r["launch_us"] * 1e-3) < 1e-6 * r["achieved"]
        assert c["parity"]["mismatches"] == 0 and c["parity"]["checked_bins"] == 256 and c["parity"]["lock_steps"] >= 20
        assert c["episodes_finished"] > 0
    assert d["configs"]["10x10x10_dataset_cut2"]["pool_sequences"] == 2100
    p = d["parity"]
    assert p["mismatches"] == 0 and p["checked_bins"] == 4 * 256 and set(p["per_workload"]) == {"10x10x10"} | set(d["configs"])
    e = d["epsilon_variant"]
    assert e["epsilon"] == 0.01 and e["value"] > 1e8 and e["mean_episode_length"] < d["config"]["mean_episode_length"]
