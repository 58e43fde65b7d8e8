#!/usr/bin/env python3
"""Headline benchmark: env steps/s of the fused HIP environment step (BASELINE.json config[1]).

One "step" = one lock-step of all bins on this rank = ONE launch of the fused step kernel (bpp_step:
action decode, placement rule, heightmap update, reward, Monitor accumulators + episode statistics,
auto-reset, next observation, feasibility mask, and -- the benchmark's action source -- a uniform-random
draw among the feasible positions of the new mask for the next lock-step).  Inputs (pool, state,
actions) are resident in HBM.

    python bench.py --gpus 1 --steps 500 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job env steps/s, plus
`roofline` (dominant kernel = bpp_step; algorithmic bytes / HIP-event-measured launch duration vs the
8 TB/s HBM peak) and `cpu_baseline` (the C oracle, a scalar port of the reference, timed on this box's
host: 1 core, bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: what a float4 copy kernel reaches (79 % of the spec peak)


def algorithmic_bytes_per_env_step(A, M):
    """contract-mandated I/O of one env step -- int32 heightmap read + write (4A + 4A),
    float32 observation write (16A), float32 mask write (4M), 64 B of per-bin scalars (action, reward,
    done, item, state/accumulators)."""
    return 4 * A + 4 * A + 16 * A + 4 * M + 64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=65536, help="bins per GPU (weak scaling)")
    ap.add_argument("--size", type=int, nargs=3, default=[10, 10, 10])
    ap.add_argument("--rotation", action="store_true")
    ap.add_argument("--pool", type=int, default=8192, help="CUT-2 sequences in the pool")
    ap.add_argument("--pool-file", default=None,
                    help="npz with a uint8 [P][T][4] `pool` array instead of generated sequences, e.g. "
                         "tests/golden/cut2_dataset_10.npz = the reference's dataset/cut_2.pt (2100 sequences)")
    ap.add_argument("--stream", action="store_true",
                    help="endless CUT-2 supply generated on the device (bpp_stream: no sequence is ever replayed) instead of "
                         "the finite pool of BASELINE's configs; the refill kernels run inside the timed region")
    ap.add_argument("--stream-depth", type=int, default=32, help="--stream: ring rows per bin")
    ap.add_argument("--stream-refill", type=int, default=14,
                    help="--stream: lock-steps between refills (with depth >= 2 * refill + 3 the refills run beside the lock-steps)")
    ap.add_argument("--reps", type=int, default=0,
                    help="repetitions of the timed K-step region; the MEDIAN repetition is reported (0 = auto: enough "
                         "repetitions for >= 100 ms of timed work, at most 25, so that a small --steps is not a 1 ms sample)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def usable_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
        except Exception:
            pass
    return max(1, n)


def _cpu_worker(job):
    """One host core: step a private shard of bins with the oracle until the time budget is used up;
    returns (lock-steps done, seconds)."""
    pool, size, rotation, bins, base, total, budget = job
    from oracle import oracle as orc
    env = orc.OracleEnv(pool, size, rotation, bins, env_id_base=base, env_id_total=total)
    env.reset()
    done, chunk = 0, 100
    t0 = time.perf_counter()
    while True:
        orc.rollout_uniform(env, 1, done, chunk)
        done += chunk
        dt = time.perf_counter() - t0
        if dt >= budget:
            return done, dt


def cpu_baseline(pool, size, rotation, seconds):
    """The oracle (oracle/bpp_oracle.c: scalar C port of the reference step + mask) on the host, same
    workload and policy, time-bounded sample: one core alone, then one process per usable core (each with
    its own shard of bins).  Checker/baseline only -- never the product path."""
    import multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    bins = 256
    n1, dt1 = _cpu_worker((pool, size, rotation, bins, 0, bins, 0.3 * seconds))
    single = bins * n1 / dt1
    cores = min(usable_cores(), 64)
    jobs = [(pool, size, rotation, bins, c * bins, cores * bins, 0.5 * seconds) for c in range(cores)]
    with mp.get_context("fork").Pool(cores) as pw:
        res = pw.map(_cpu_worker, jobs)
    rate = sum(bins * n / dt for n, dt in res)     # every worker measured over its own busy interval
    return {"value": rate, "unit": "env steps/s", "cores": cores, "kind": "port",
            "single_core_value": single, "python_port": python_port_baseline(pool, size, rotation, cores, 0.4 * seconds),
            "reference_python_context": reference_python_context(),
            "sample": "oracle/bpp_oracle.c (scalar C restatement of PackingGame.step + acktr.utils mask); "
                      "%d processes x %d bins for %.1f s each (sum of per-process rates), and %d bins x %d lock-steps "
                      "in %.1f s on one core; same CUT-2 pool and uniform-feasible policy; os.cpu_count()=%s"
                      % (cores, bins, 0.5 * seconds, bins, n1, dt1, os.cpu_count())}


def python_port_baseline(pool, size, rotation, cores, seconds):
    """north_star asks for the reference SubprocVecEnv timed on this box's host cores in this run; the reference tree
    does not travel to the GPU box, so what is timed here is oracle/ref_port.py: a pure-Python / numpy restatement of one
    reference worker's step plus the parent's mask loop with the same numpy work per candidate position, one forked
    process per usable core -- the reference's plumbing with a perfectly parallel mask loop.  Its outputs are pinned to
    the C oracle (tests/test_ref_port.py); in the build container it runs at 1.02 - 1.10 x the speed of the live
    reference on the same core (oracle/time_reference.py, profiles/r03o_reference_python_and_port_cpu_here.json)."""
    try:
        from oracle import ref_port
        rate, longest = ref_port.timed_all_cores(pool, size, rotation, seconds, cores)
        return {"value": rate, "unit": "env steps/s", "cores": cores, "per_core": rate / cores, "kind": "port (Python)",
                "sample": "oracle/ref_port.py, %d forked workers x 1 bin for %.1f s each; same CUT-2 pool, "
                          "uniform-feasible policy" % (cores, longest),
                "speed_vs_live_reference_same_core": "1.02-1.10x (build container, profiles/r03o_reference_python_and_port_cpu_here.json)"}
    except Exception as e:       # a baseline must never take the bench line down
        return {"error": repr(e)}


def reference_python_context():
    """The UNMODIFIED reference Python cannot run on the GPU box (/root/reference is not shipped there); its
    host throughput was measured in the build container (oracle/time_reference.py) and is carried here as
    labelled context next to the C port's number -- not measured in this run, not on this box."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r03o_reference_python_and_port_cpu_here.json")))
        d = dict(d)
        d["note"] = ("measured in the build container (8 cores), not on this box and not in this run: the reference "
                     "tree does not travel to the GPU box; profiles/r03o_reference_python_and_port_cpu_here.json")
        return d
    except Exception:
        return None


def profile_evidence(key):
    """Counter-derived facts about the step kernel for this workload from profiles/hbm_traffic.json (written by
    tools/pmc_to_json.py from rocprofv3 PMC passes of the same bench command): HBM bytes per launch, VALU
    utilisation.  None when no profile of this workload is committed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        v = d.get(key)
        if isinstance(v, dict):
            return v
        if isinstance(v, (int, float)):
            return {"traffic_bytes": v}
    except Exception:
        pass
    return None


def limiter(moved_gbs, valu_util):
    """What the counters say holds the kernel back: the memory side when the bytes it really moves come close to what a
    copy kernel achieves, the vector ALUs when they are the busier resource."""
    if moved_gbs is None or valu_util is None:
        return None
    mem = moved_gbs / HBM_ACHIEVABLE_GBS
    return ("hbm: moves %.0f %% of the 6.3 TB/s a copy kernel achieves (VALU pipes %.0f %% busy)" % (100 * mem, 100 * valu_util)
            if mem >= valu_util else
            "valu issue: VALU pipes %.0f %% busy (memory side at %.0f %% of the achievable 6.3 TB/s)" % (100 * valu_util, 100 * mem))


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import bpp_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    size = tuple(args.size)
    A = size[0] * size[1]
    M = A * (2 if args.rotation else 1)
    E = args.envs
    if args.pool_file:
        import numpy as np
        pool = np.load(args.pool_file)["pool"]
    else:
        pool = bpp_amd.sequences.cut2_pool(size, args.pool, seed=0)   # identical on every rank (cpu_baseline uses it too)
    cpu_base = None
    if world == 1 and not args.no_cpu_baseline:
        # before the HIP runtime is initialised in this process: the baseline forks one worker per core
        try:
            cpu_base = cpu_baseline(pool, size, args.rotation, args.cpu_seconds)
        except Exception as exc:  # noqa: BLE001 -- the baseline leg must never take the GPU measurement down
            cpu_base = {"value": None, "unit": "env steps/s", "cores": 0, "kind": "port",
                        "sample": "cpu baseline failed: %r" % (exc,)}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    # BPP_BENCH_BACKEND=gloo + BPP_BENCH_ONE_DEVICE=1: smoke-test the multi-rank path on a 1-GPU box
    # (all ranks on device 0, gloo instead of RCCL); never set by the driver.
    backend = os.environ.get("BPP_BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    dev_index = 0 if os.environ.get("BPP_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # BPP_BENCH_FORCE_PG=1: initialise the process group (and run the barrier / stats all-reduce through it) even
    # with ONE rank, so that the RCCL branch can be exercised on a 1-GPU box; never set by the driver.
    use_pg = world > 1 or bool(os.environ.get("BPP_BENCH_FORCE_PG"))
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    env = bpp_amd.BppVecEnv(E, size, enable_rotation=args.rotation, pool=None if args.stream else pool, device=device,
                            env_id_base=rank * E, env_id_total=world * E,
                            stream=dict(bound=(2, 5), seed=0, depth=args.stream_depth, refill_every=args.stream_refill) if args.stream else None)
    stats = bpp_amd.EpisodeStats(device)
    actions = torch.empty(E, dtype=torch.int64, device=device)
    env.reset()

    def lockstep(t):
        # actions for lock-step t were drawn inside lock-step t-1 (or by sample_feasible before the loop)
        return env.step_tensors(actions, sample=(1, t + 1, actions))

    def fence():
        torch.cuda.synchronize(device)
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(device)

    # warm-up and the timed region are driven by ONE native call each (bpp_rollout_uniform: the same
    # sample+step launches, enqueued from C instead of from a Python loop)
    env.rollout_uniform(seed=1, step0=0, nsteps=args.warmup, actions=actions)
    stats.collect(env).all_reduce()   # also loads the few torch kernels the collection uses
    stats.zero_()
    # timed region = EXACTLY K lock-steps: barrier + synchronize, clock, K lock-steps, synchronize, clock (the maximum
    # over ranks is taken afterwards); repeated `reps` times and the MEDIAN repetition reported, so that a small K is not
    # a single sub-millisecond sample.  The path's only collective -- the 32-byte statistics all-reduce -- runs inside
    # the timed region once per logging interval of LOG_INTERVAL lock-steps, the reference's own cadence
    # (main.py:194-: every log_interval = 10 updates of num_steps = 5 lock-steps).
    LOG_INTERVAL = 50
    since_log = [0]

    def timed_region(step0):
        fence()
        t0 = time.perf_counter()
        env.rollout_uniform(seed=1, step0=step0, nsteps=args.steps, actions=actions)
        since_log[0] += args.steps
        if since_log[0] >= LOG_INTERVAL:
            stats.collect(env).all_reduce()
            since_log[0] = 0
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    first = timed_region(args.warmup)
    reps = args.reps if args.reps > 0 else max(1, min(25, int(0.1 / max(first, 1e-6)) + 1))
    if world > 1:   # every rank must run the same number of repetitions
        rt = torch.tensor([reps], dtype=torch.int64, device=device)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        reps = int(rt.item())
    samples = [first] + [timed_region(args.warmup + (r + 1) * args.steps) for r in range(reps - 1)]
    if world > 1:   # a repetition takes as long as its slowest rank
        tm = torch.tensor(samples, dtype=torch.float64, device=device)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        samples = tm.cpu().tolist()
    dt = sorted(samples)[len(samples) // 2]
    done_steps = args.warmup + reps * args.steps
    stats.collect(env).all_reduce()   # whatever finished since the last logging point (outside the timed regions)
    # same K lock-steps driven step by step from Python (what a Python RL loop pays per step)
    fence()
    t1 = time.perf_counter()
    env.sample_feasible(seed=1, step=done_steps, out=actions)
    for t in range(args.steps):
        lockstep(done_steps + t)
    fence()
    dt_py = time.perf_counter() - t1
    summary = stats.summary()

    # dominant kernel (bpp_step) launch duration, HIP events on the launch stream, after the timed region: one event
    # pair around n_ev lock-steps enqueued back to back by ONE native call (each lock-step is one launch of the step
    # kernel; the call's single 8 us draw of the first action is in there once), so that neither Python nor the events'
    # own ~2 us sit between two launches.  Event pairs around single launches are kept as `launch_us_event_pairs`.
    n_ev = min(args.steps, 200)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for t, (e0, e1) in enumerate(evs):
        e0.record()
        lockstep(done_steps + args.steps + t)
        e1.record()
    torch.cuda.synchronize(device)
    kern_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    pair_avg_ms = sum(kern_ms) / len(kern_ms)
    if args.stream:     # refill kernels run between / beside the lock-steps: only the pairs isolate the step kernel
        kern_avg_ms = pair_avg_ms
    else:
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        env.rollout_uniform(seed=1, step0=done_steps + args.steps + n_ev, nsteps=n_ev, actions=actions)
        b1.record()
        torch.cuda.synchronize(device)
        kern_avg_ms = b0.elapsed_time(b1) / n_ev

    if rank == 0:
        headline = size == (10, 10, 10) and not args.rotation and E == 65536 and not args.stream
        metric = "env steps/sec (whole node), %dx%dx%d bin%s, %d envs per GPU%s; bit-exact mask vs ref" % (
            size + (" + rotation" if args.rotation else "", E, ", endless device-generated CUT-2 supply" if args.stream else ""))
        if headline:    # BASELINE.json's metric string belongs to its own workload only
            try:
                metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            except Exception:
                pass
        b_alg = algorithmic_bytes_per_env_step(A, M)
        achieved = b_alg * E / (kern_avg_ms * 1e-3) / 1e9
        ev = profile_evidence("%dx%dx%d_rot%d_E%d" % (size + (int(args.rotation), E)))
        traffic = ev.get("traffic_bytes") if ev else None
        moved = traffic / (kern_avg_ms * 1e-3) / 1e9 if traffic else None
        out = {
            "metric": metric,
            "value": world * E * args.steps / dt,
            "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "reps": reps, "stats_all_reduce_every_lock_steps": LOG_INTERVAL, "rep_ms_per_step_min_median_max": [min(samples) / args.steps * 1e3, dt / args.steps * 1e3,
                                                            max(samples) / args.steps * 1e3],
            "ms_per_step": dt / args.steps * 1e3,
            "python_loop_ms_per_step": dt_py / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%dx%dx%d bin, CUT-2 sequences%s, %d envs per MI355X, uniform-random-feasible policy"
                                   % (size + (" + rotation" if args.rotation else "", E)),
                       "envs_per_gpu": E, "total_envs": world * E, "pool_sequences": int(pool.shape[0]),
                       "pool_source": ("device stream: random.Random(g) per bin, ring of %d rows, refill every %d lock-steps%s (bpp_stream)"
                                       % (args.stream_depth, args.stream_refill,
                                          " beside the lock-steps" if args.stream_depth >= 2 * args.stream_refill + 3
                                          and bpp_amd._lib.get_knobs()["stream_overlap"] else "") if args.stream
                                       else args.pool_file or "generated CUT-2 (sequences.cut2_pool, seed 0)"),
                       "sharding": "bins by global id, %d rank(s); 32-byte stats all-reduce only (%s%s)"
                                   % (world, "RCCL" if backend == "nccl" else backend,
                                      "" if use_pg else ", no process group at 1 rank"),
                       "episodes_finished": summary["episodes"], "mean_ratio": round(summary["mean_ratio"], 4),
                       "mean_episode_length": round(summary["mean_length"], 2)},
            # `achieved`/`frac`: ALGORITHMIC bytes (int32 heightmaps as in the reference's layout) per
            # launch / launch duration.  `achieved_moved`/`frac_moved`: the bytes the kernel really moves (PMC
            # counters; the state is kept as bytes, so fewer than the algorithmic ones) / the same duration -- the
            # actual HBM bandwidth.  `limiter`: what the SQ counters say bounds the kernel today.
            "roofline": {"bound": "hbm", "kernel": "bpp_step (%s)" % bpp_amd._lib.launch_info(E, size, args.rotation)["kernel_name"],
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/hbm_traffic.json)",
                         "achieved_moved": moved, "frac_moved": (moved / HBM_PEAK_GBS) if moved else None,
                         "frac_moved_of_achievable": (moved / HBM_ACHIEVABLE_GBS) if moved else None,
                         "valu_utilisation": (ev or {}).get("valu_utilisation"),
                         "limiter": limiter(moved, (ev or {}).get("valu_utilisation")),
                         "bytes_per_env_step": b_alg, "launch_us": kern_avg_ms * 1e3,
                         "launch_us_event_pairs": [kern_ms[0] * 1e3, pair_avg_ms * 1e3]},
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
