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


def _run(args, **extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_clean_env(**extra), cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
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

    monkeypatch.setattr(subprocess, "call", fake_call)
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


@pytest.mark.gpu
def test_gpu_bench_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (no torchrun): it starts its two ranks itself; on this 1-GPU box both ranks share
    device 0 (BPP_BENCH_ONE_DEVICE=1 -> gloo for the barrier and the 32-byte statistics all-reduce)."""
    d = _run(["--gpus", "2", "--steps", "20", "--warmup", "5"], BPP_BENCH_ONE_DEVICE="1")
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 131072 and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["launcher"] == "self-launched torch.distributed.run"
    assert d["value"] > 1e8 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["timed_gpu_work_ms"] >= 150.0
    assert d["config"]["episodes_finished"] > 0


@pytest.mark.gpu
def test_gpu_bench_spawned_single_rank_equals_direct_run():
    """N = 1 through the self-launch path gives the same line as the direct run (same workload, value within noise),
    and the line carries both roofline fractions (L3-assisted and past the Infinity Cache)."""
    a = _run(["--steps", "100", "--warmup", "20", "--no-cpu-baseline"])
    b = _run(["--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--launcher", "spawn"])
    assert a["config"]["launcher"] == "direct" and b["config"]["launcher"] == "self-launched torch.distributed.run"
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["config"]["total_envs"] == b["config"]["total_envs"] == 65536
    assert abs(a["value"] / b["value"] - 1.0) < 0.1
    for d in (a, b):
        r = d["roofline"]
        assert 0.3 < r["frac_past_l3"] <= r["frac"] * 1.05 < 1.05
        assert d["value_past_l3"] <= d["value"] * 1.05
        assert d["past_l3"]["output_span_MB"] > 1000 and d["timed_gpu_work_ms"] >= 200.0
        assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
