"""Multi-GPU readiness.  The harness box has ONE GPU, so the two RCCL tests below switch themselves on only
where torch.cuda.device_count() >= 2 (a multi-GPU driver box); the build-lock test runs everywhere (CPU).

RCCL with more than one rank has never executed in this project's own runs -- these tests are what will execute it the
first time a box with two devices runs `pytest -m gpu`."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:  # noqa: BLE001
        return 0


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "BPP_BENCH_CHILD", "BPP_BENCH_ONE_DEVICE", "BPP_BENCH_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def _bench(args):
    import signal
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_clean_env(), cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=900)
    except subprocess.TimeoutExpired:       # kill the whole group: the launcher's ranks would otherwise keep the pipes open
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError("bench.py %r did not finish\n%s" % (args, err[-2000:]))
    assert p.returncode == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.skipif(_devices() < 2, reason="needs two HIP devices (RCCL across ranks)")
def test_bench_two_gpus_over_rccl_scales():
    """`python bench.py --gpus 2` on a box with two devices: one rank per GPU over real RCCL (no BPP_BENCH_ONE_DEVICE),
    weak scaling -- twice the bins, at least 1.8x the N = 1 throughput."""
    one = _bench(["--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--no-past-l3", "--gpu-seconds", "1"])
    two = _bench(["--gpus", "2", "--steps", "100", "--warmup", "20", "--no-past-l3", "--gpu-seconds", "1"])
    assert two["n_gpus"] == 2 and two["config"]["total_envs"] == 2 * one["config"]["total_envs"]
    assert "RCCL" in two["config"]["sharding"] and "2 rank(s)" in two["config"]["sharding"]
    assert two["scaling"] == "weak"
    # round 6: an N > 1 line is complete by itself -- the reference timed on this box in this run, rank 0 alone in the same run
    assert two["cpu_baseline"]["value"] > 1e3 and two["roofline"]["per_gpu"] is True
    assert two["n1_value_same_run"] > 1e8 and two["scaling_efficiency_same_run"] >= 0.9
    assert two["value"] >= 1.8 * one["value"], (one["value"], two["value"])
    assert two["config"]["episodes_finished"] > one["config"]["episodes_finished"]


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np, torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)                       # before the library is touched: launches go to the rank's device
    import bpp_amd
    bpp_amd._lib.lib()
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)       # "nccl" is RCCL on ROCm
    SIZE, TOTAL, STEPS, SEED = (10, 10, 10), 4099, 40, 5
    pool = bpp_amd.sequences.cut2_pool(SIZE, 23, seed=2)
    lo, hi = bpp_amd.shard_range(TOTAL, rank, world)
    env = bpp_amd.BppVecEnv(hi - lo, SIZE, enable_rotation=True, pool=pool, device=dev, env_id_base=lo, env_id_total=TOTAL)
    env.reset()
    obs = []
    for t in range(STEPS):
        obs.append(env.step_tensors(env.sample_feasible(seed=SEED, step=t)).obs.cpu().numpy().copy())
    stats = bpp_amd.EpisodeStats(dev).collect(env)
    mine = stats.acc.cpu().numpy().copy()
    stats.all_reduce()                                # 32 bytes over RCCL, on the device
    np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), obs=np.stack(obs), mine=mine, acc=stats.acc.cpu().numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.gpu
@pytest.mark.skipif(_devices() < 2, reason="needs two HIP devices (RCCL across ranks)")
def test_two_rccl_ranks_equal_one_global_run(tmp_path, oracle):
    """Two ranks, one GPU each, bins sharded by global id: concatenated observations == the oracle's single global run;
    the all-reduced record == the sum of the shards' fixed-order reductions, exactly (two ranks: one float64 add)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=_clean_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    parts = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    import bpp_amd
    SIZE, TOTAL, STEPS, SEED = (10, 10, 10), 4099, 40, 5
    pool = bpp_amd.sequences.cut2_pool(SIZE, 23, seed=2)

    def rollout(lo, hi):
        env = oracle.OracleEnv(pool, SIZE, True, hi - lo, env_id_base=lo, env_id_total=TOTAL)
        _, mask = env.reset()
        obs = []
        for t in range(STEPS):
            o = env.step(oracle.sample_feasible(mask, SEED, t, env_id_base=lo))
            mask = o["mask"]
            obs.append(o["obs"])
        return np.stack(obs), env.episode_stats()

    g_obs, g_acc = rollout(0, TOTAL)
    np.testing.assert_array_equal(np.concatenate([p["obs"] for p in parts], axis=1), g_obs)
    shard = [rollout(int(p["lo"]), int(p["hi"]))[1] for p in parts]
    for k, p in enumerate(parts):
        np.testing.assert_array_equal(p["mine"], shard[k])
        np.testing.assert_array_equal(p["acc"], shard[0] + shard[1])
        np.testing.assert_array_equal(p["acc"][2:], g_acc[2:])


FAKE_HIPCC = textwrap.dedent("""\
    #!/bin/bash
    # stands where hipcc is: records that it ran, takes its time, writes the -o file
    echo "compile $$" >> "$BPP_FAKE_LOG"
    sleep 0.4
    out=""
    while [ $# -gt 0 ]; do if [ "$1" = "-o" ]; then out="$2"; fi; shift; done
    echo "fake library" > "$out"
""")

BUILDER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %(root)r)
    import bpp_amd
    L = bpp_amd._lib
    L.CSRC = %(tmp)r
    L.BUILD_LIB = L.LIB = os.path.join(%(tmp)r, "libbpp_hip.so")
    L.DEPS = [os.path.join(%(tmp)r, "source.hip")]
    while not os.path.exists(os.path.join(%(tmp)r, "go")):      # all eight start building at the same moment
        time.sleep(0.005)
    assert L.build() == L.LIB and os.path.exists(L.LIB)
""")


def test_eight_ranks_build_once(tmp_path):
    """N ranks that find the library stale at the same moment (a fresh box, `bench.py --gpus 8`): ONE of them compiles,
    the others wait on the lock and then find it fresh (online-3d-bpp-drl_amd/_lib.py: build)."""
    tmp = str(tmp_path)
    open(os.path.join(tmp, "source.hip"), "w").write("// stale on purpose\n")
    fake = os.path.join(tmp, "fake_hipcc.sh")
    open(fake, "w").write(FAKE_HIPCC)
    os.chmod(fake, 0o755)
    log = os.path.join(tmp, "compiles.log")
    script = os.path.join(tmp, "builder.py")
    open(script, "w").write(BUILDER % dict(root=ROOT, tmp=tmp))
    env = dict(os.environ, HIPCC=fake, BPP_FAKE_LOG=log)
    env.pop("BPP_HIP_LIB", None)
    procs = [subprocess.Popen([sys.executable, script], env=env, stderr=subprocess.PIPE, text=True) for _ in range(8)]
    import time
    time.sleep(3.0)                                   # every interpreter has imported the package by now
    open(os.path.join(tmp, "go"), "w").close()
    for p in procs:
        _, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
    assert len(open(log).read().splitlines()) == 1    # exactly one compilation
    assert not [f for f in os.listdir(tmp) if ".tmp." in f]
