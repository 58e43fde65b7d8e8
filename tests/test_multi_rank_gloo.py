"""N > 1 path on CPU: two `gloo` ranks, each owning a contiguous shard of global bin ids.  The shard
arithmetic (shard_range, env_id_base/env_id_total sequence assignment) and the 32-byte statistics
all-reduce (EpisodeStats.all_reduce) are the product code under test; the per-shard stepping is done by
the oracle (there is no GPU here), and the concatenation of the shards must equal one global run."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bpp_amd

SIZE, TOTAL, STEPS, SEED = (10, 10, 10), 52, 25, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rollout(pool, lo, hi, total):
    from oracle import oracle as orc
    env = orc.OracleEnv(pool, SIZE, True, hi - lo, env_id_base=lo, env_id_total=total)
    _, mask = env.reset()
    obs = []
    for t in range(STEPS):
        a = orc.sample_feasible(mask, SEED, t, env_id_base=lo)
        o = env.step(a)
        mask = o["mask"]
        obs.append(o["obs"])
    return np.stack(obs), env.episode_stats()


def _worker(rank, world, port, pool, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bpp_amd.shard_range(TOTAL, rank, world)
    obs, acc = _rollout(pool, lo, hi, TOTAL)
    stats = bpp_amd.EpisodeStats("cpu")
    stats.acc += torch.from_numpy(acc)
    stats.all_reduce()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), obs=obs, acc=stats.acc.numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_equal_one_global_run(tmp_path):
    pool = bpp_amd.sequences.cut2_pool(SIZE, 23, seed=2)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), pool, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert [(int(p["lo"]), int(p["hi"])) for p in parts] == [(0, 26), (26, 52)]
    g_obs, g_acc = _rollout(pool, 0, TOTAL, TOTAL)
    np.testing.assert_array_equal(np.concatenate([p["obs"] for p in parts], axis=1), g_obs)
    for p in parts:                                   # every rank holds the global sums after the all-reduce
        np.testing.assert_allclose(p["acc"], g_acc, rtol=1e-12)
    assert g_acc[3] > 0


def test_shard_range_covers_everything_once():
    for total, world in ((65536 * 8, 8), (10, 3), (7, 8), (1, 1)):
        spans = [bpp_amd.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_episode_stats_all_reduce_is_noop_without_process_group():
    s = bpp_amd.EpisodeStats("cpu")
    s.acc += torch.tensor([4.0, 2.0, 8.0, 2.0], dtype=torch.float64)
    assert s.all_reduce().summary() == {"episodes": 2, "mean_return": 2.0, "mean_ratio": 1.0, "mean_length": 4.0}


def _gpu_worker(rank, world, port, pool, out_dir):
    """Same as _worker, but every rank steps the PRODUCT (both ranks share GPU 0 on a 1-GPU box; gloo collective)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bpp_amd.shard_range(TOTAL, rank, world)
    env = bpp_amd.BppVecEnv(hi - lo, SIZE, enable_rotation=True, pool=pool, device="cuda:0", env_id_base=lo, env_id_total=TOTAL)
    env.reset()
    obs = []
    for t in range(STEPS):
        a = env.sample_feasible(seed=SEED, step=t)
        obs.append(env.step_tensors(a).obs.cpu().numpy().copy())
    stats = bpp_amd.EpisodeStats("cuda:0").collect(env)
    acc = stats.acc.cpu()
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)        # gloo moves host tensors; on a multi-GPU node this is RCCL on device
    np.savez(os.path.join(out_dir, "grank%d.npz" % rank), obs=np.stack(obs), acc=acc.numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpu_two_ranks_of_the_product_equal_one_global_run(tmp_path):
    """The N > 1 path with the HIP kernels doing the stepping: two ranks (sharing the box's GPU) each own a shard of
    global bin ids; concatenated they equal the oracle's single global run, and the all-reduced statistics agree."""
    pool = bpp_amd.sequences.cut2_pool(SIZE, 23, seed=2)
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), pool, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "grank%d.npz" % r)) for r in range(world)]
    g_obs, g_acc = _rollout(pool, 0, TOTAL, TOTAL)
    np.testing.assert_array_equal(np.concatenate([p["obs"] for p in parts], axis=1), g_obs)
    # the all-reduced record is exactly the sum of the shards' fixed-order reductions (two ranks: one float64 add) ...
    shard_acc = [_rollout(pool, int(p["lo"]), int(p["hi"]), TOTAL)[1] for p in parts]
    for p in parts:
        np.testing.assert_array_equal(p["acc"], shard_acc[0] + shard_acc[1])
        # ... and agrees with the single global run up to the association of that last add
        np.testing.assert_array_equal(p["acc"][2:], g_acc[2:])
        np.testing.assert_allclose(p["acc"][:2], g_acc[:2], rtol=1e-12)
