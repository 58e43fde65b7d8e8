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
    python bench.py --gpus N ...        (no launcher: starts its N ranks itself through torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job env steps/s, plus
`roofline` (dominant kernel = bpp_step; algorithmic bytes / HIP-event-measured launch duration vs the
8 TB/s HBM peak) and `cpu_baseline` (the unmodified reference Python from oracle/_ref/ -- its own ShmemVecEnv
plumbing and one worker per core -- timed on this box's host cores in this run, the C restatement beside it;
bounded samples; N=1 only).
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
LOG_INTERVAL = 50            # lock-steps between two statistics all-reduces inside the timed regions


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
                    help="npz with a uint8 [P][T][4] `pool` array (every row ending in the terminator: (10,10,10) for a 10x10x10 "
                         "bin, else the bin size) or a reference dataset/*.pt, instead of generated sequences; played in "
                         "LoadBoxCreator's order (row r = trajectory r + 1), e.g. tests/golden/cut2_dataset_10.npz = the "
                         "reference's dataset/cut_2.pt (2100 sequences)")
    ap.add_argument("--stream", action="store_true",
                    help="endless CUT-2 supply generated on the device (bpp_stream: no sequence is ever replayed) instead of "
                         "the finite pool of BASELINE's configs; the refill kernels run inside the timed region")
    ap.add_argument("--stream-cache", choices=("auto", "on", "off"), default="auto",
                    help="--stream: row cache (bpp_batch.seq_cache); auto = BppVecEnv's default")
    ap.add_argument("--stream-rng", choices=("mt19937", "counter"), default="mt19937",
                    help="--stream: mt19937 = every bin an exact random.Random(seed + id) (sequences identical to the reference "
                         "creator's under that seed); counter = the same cutting algorithm on a stateless counter-based generator "
                         "(distribution parity, SURVEY 8f2's bar)")
    ap.add_argument("--stream-depth", type=int, default=32, help="--stream: ring rows per bin")
    ap.add_argument("--stream-refill", type=int, default=14,
                    help="--stream: lock-steps between refills (with depth >= 2 * refill + 3 the refills run beside the lock-steps)")
    ap.add_argument("--reps", type=int, default=0,
                    help="repetitions of the timed K-step region; the MEDIAN repetition is reported (0 = auto: as many as "
                         "it takes for --gpu-seconds of timed GPU work, so that a small --steps is not a 1 ms sample)")
    ap.add_argument("--gpu-seconds", type=float, default=3.0,
                    help="timed GPU work of the headline leg (the past-L3 leg gets two thirds of it): long enough for an "
                         "outside utilisation sampler to see the device busy; the median repetition is what is reported")
    ap.add_argument("--launcher", choices=("auto", "direct", "spawn"), default="auto",
                    help="auto: --gpus N > 1 started without torch.distributed.run launches its N ranks itself; spawn: do "
                         "that for N = 1 as well")
    ap.add_argument("--no-past-l3", action="store_true", help="skip the rotating-output-sets (HBM-only) leg")
    ap.add_argument("--past-l3-only", action="store_true",
                    help="profiling aid: every timed region writes rotating output sets, so that a rocprofv3 kernel "
                         "summary of this command averages past-the-Infinity-Cache launches only (the JSON line is then "
                         "that leg's and says so)")
    ap.add_argument("--only-headline", action="store_true",
                    help="with the default workload (BASELINE config 2): do NOT also run configs 3 (rotation), 4 (20x20x20, 32 768 "
                         "bins), the primary dataset pool and the epsilon variant in the same line")
    ap.add_argument("--extra-seconds", type=float, default=1.0,
                    help="timed GPU work of each extra config's one-output-set leg (its past-L3 leg gets 0.7 of it)")
    ap.add_argument("--eps", type=float, default=0.01, help="epsilon of SURVEY 8d's failure-path variant leg")
    ap.add_argument("--eps-seconds", type=float, default=0.7, help="timed GPU work of the epsilon leg (0 = skip)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity gate (profiling runs)")
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
    """The CPU side of the line, timed on THIS box's host cores in THIS run (rank 0, N = 1 only; baselines, not targets):

    * `kind: "reference"` -- the UNMODIFIED reference Python from oracle/_ref/ (byte-for-byte copies made by the
      committed recipe oracle/make_ref.py; /root/reference is never read here), in a process of its own
      (oracle/ref_baseline.py): R1 = its ShmemVecEnv plumbing as it is (16 forked workers + the parent's mask loop,
      BASELINE config[0]); R2 = one forked worker per usable core running PackingGame.step + acktr.utils masks on the
      bench's own CUT-2 pool and policy.  `value` is R2 (same workload, all cores), R1 is carried beside it.
    * `ours_cpu` -- oracle/bpp_oracle.c, the scalar C restatement (the parity checker), same pool and policy.

    Without oracle/_ref/ the line falls back to `kind: "port"` (the C restatement as `value`, plus the Python
    restatement oracle/ref_port.py) and says so."""
    import multiprocessing as mp
    from oracle import oracle as orc
    from oracle import ref_shims
    orc.build()
    cores = min(usable_cores(), 64)
    have_ref = ref_shims.copy_available()
    c_budget = (0.35 if have_ref else 0.6) * seconds
    bins = 256
    n1, dt1 = _cpu_worker((pool, size, rotation, bins, 0, bins, 0.4 * c_budget))
    single = bins * n1 / dt1
    jobs = [(pool, size, rotation, bins, c * bins, cores * bins, 0.6 * c_budget) for c in range(cores)]
    with mp.get_context("fork").Pool(cores) as pw:
        res = pw.map(_cpu_worker, jobs)
    rate = sum(bins * n / dt for n, dt in res)     # every worker measured over its own busy interval
    ours = {"value": rate, "unit": "env steps/s", "cores": cores, "kind": "port", "single_core_value": single,
            "sample": "oracle/bpp_oracle.c (scalar C restatement of PackingGame.step + acktr.utils mask); %d processes x %d bins "
                      "for %.1f s each (sum of per-process rates), and %d bins x %d lock-steps in %.1f s on one core; same CUT-2 "
                      "pool and uniform-feasible policy" % (cores, bins, 0.6 * c_budget, bins, n1, dt1)}
    if not have_ref:
        ours.update({"reference_subprocvecenv_timed_here": False,
                     "why_not": "oracle/_ref/ is missing on this box (made by oracle/make_ref.py in the build container; it travels "
                                "with the working tree, not with git): the reference's own plumbing could not be timed here",
                     "python_port": python_port_baseline(pool, size, rotation, cores, 0.4 * seconds)})
        return ours
    ref = reference_baseline(pool, size, rotation, cores, 0.3 * seconds, 0.35 * seconds)
    r1, r2 = ref.get("R1", {}), ref.get("R2", {})
    if "value" not in r2:       # the reference leg failed: say why, fall back to the port as the value
        ours.update({"reference_subprocvecenv_timed_here": False, "reference_error": ref})
        return ours
    return {"value": r2["value"], "unit": "env steps/s", "cores": cores, "kind": "reference",
            "reference_subprocvecenv_timed_here": "value" in r1,
            "sample": "R2: unmodified reference Python (oracle/_ref/), %d forked workers x 1 bin for %.1f s each = %d env steps: "
                      "PackingGame.step + acktr.utils mask per worker, the bench's CUT-2 pool, uniform-feasible policy.  "
                      "R1 (reference plumbing as it is) beside it; os.cpu_count()=%s, usable cores %d"
                      % (cores, r2.get("seconds", 0.0), r2.get("env_steps", 0), os.cpu_count(), cores),
            "reference_parallel_R2": r2, "reference_as_is_R1": r1, "ours_cpu": ours}


def reference_baseline(pool, size, rotation, cores, r1_seconds, r2_seconds):
    """oracle/ref_baseline.py in its own process (the reference's modules never enter this one; it forks its workers
    from a process that holds no HIP state)."""
    import subprocess
    import tempfile
    import numpy as np
    from oracle import ref_shims
    try:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "pool.npz")
            np.savez(path, pool=np.asarray(pool))
            spec = {"pool": path, "size": list(size), "rotation": bool(rotation), "cores": cores,
                    "r1_seconds": r1_seconds, "r2_seconds": r2_seconds}
            env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_baseline.py"), ref_shims.REF_COPY, json.dumps(spec)],
                                 capture_output=True, text=True, env=env, timeout=60 + 4 * (r1_seconds + r2_seconds))
        for line in out.stdout.splitlines():
            if line.startswith("REF_BASELINE="):
                return json.loads(line[len("REF_BASELINE="):])
        return {"error": "no result line", "stderr_tail": out.stderr[-400:]}
    except Exception as exc:  # noqa: BLE001 -- a baseline must never take the bench line down
        return {"error": repr(exc)}


def python_port_baseline(pool, size, rotation, cores, seconds):
    """Fallback only (no oracle/_ref/ on the box): oracle/ref_port.py, a pure-Python / numpy restatement of one reference
    worker's step plus the parent's mask loop, one forked process per usable core.  Its outputs are pinned to the C
    oracle (tests/test_ref_port.py)."""
    try:
        from oracle import ref_port
        rate, longest = ref_port.timed_all_cores(pool, size, rotation, seconds, cores)
        return {"value": rate, "unit": "env steps/s", "cores": cores, "per_core": rate / cores, "kind": "port (Python)",
                "sample": "oracle/ref_port.py, %d forked workers x 1 bin for %.1f s each; same CUT-2 pool, "
                          "uniform-feasible policy" % (cores, longest)}
    except Exception as e:       # a baseline must never take the bench line down
        return {"error": repr(e)}


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


def limiter(moved_gbs_past_l3, valu_util):
    """What holds the kernel back, comparing like with like: the bytes it moves per second PAST the Infinity Cache
    (rotating output sets: every written byte goes to HBM) against the 6.29 TB/s a copy kernel reaches from HBM, and the
    vector ALUs' busy fraction from the SQ counters."""
    if moved_gbs_past_l3 is None or valu_util is None:
        return None
    mem = moved_gbs_past_l3 / HBM_ACHIEVABLE_GBS
    return ("hbm: past the Infinity Cache the kernel moves %.0f %% of the 6.29 TB/s a copy kernel achieves from HBM (VALU pipes %.0f %% busy)"
            % (100 * mem, 100 * valu_util) if mem >= valu_util else
            "valu issue: VALU pipes %.0f %% busy (past the Infinity Cache the memory side runs at %.0f %% of the 6.29 TB/s copy rate)"
            % (100 * valu_util, 100 * mem))


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run: start the N ranks ourselves (one process per
    GPU, the launcher the task statement names) and hand its exit code back.  Rank 0 of the children prints the JSON."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, BPP_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    tmp = None
    if not args.no_cpu_baseline:
        # north_star: the reference timed "on the same box's host cores in the same run" next to EVERY N.  Timed here, once,
        # before the ranks exist (nothing else runs on the host cores), and handed to rank 0 through a file.
        import tempfile
        import bpp_amd
        size = tuple(args.size)
        pool = (bpp_amd.sequences.from_dataset(args.pool_file, size, terminator=(10, 10, 10) if size == (10, 10, 10) else size)
                if args.pool_file else bpp_amd.sequences.cut2_pool(size, args.pool, seed=0))
        try:
            cb = cpu_baseline(pool, size, args.rotation, args.cpu_seconds)
        except Exception as exc:  # noqa: BLE001
            cb = {"value": None, "unit": "env steps/s", "cores": 0, "kind": "port", "sample": "cpu baseline failed: %r" % (exc,)}
        cb["timed_by"] = "the self-launching parent process, before the %d ranks were started" % args.gpus
        fd, tmp = tempfile.mkstemp(prefix="bpp_cpu_baseline_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(cb, f)
        env["BPP_BENCH_CPU_BASELINE_FILE"] = tmp
    try:
        return subprocess.call(cmd, env=env)
    finally:
        if tmp and os.path.exists(tmp):
            os.unlink(tmp)


class Ctx(object):
    """What every workload of one bench process shares: the rank's device, the process group, the arguments."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import bpp_amd
        self.args, self.torch, self.dist, self.bpp = args, torch, dist, bpp_amd
        self.spawned = "WORLD_SIZE" in os.environ and "RANK" in os.environ
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's --nproc-per-node must equal --gpus" % (args.gpus, self.world))
        # BPP_BENCH_BACKEND=gloo + BPP_BENCH_ONE_DEVICE=1: smoke-test the multi-rank path on a 1-GPU box
        # (all ranks on device 0, gloo instead of RCCL); never set by the driver.
        self.one_device = bool(os.environ.get("BPP_BENCH_ONE_DEVICE"))
        self.backend = os.environ.get("BPP_BENCH_BACKEND", "gloo" if self.one_device else "nccl")       # "nccl" is RCCL on ROCm
        # BPP_BENCH_FORCE_PG=1: initialise the process group (and run the barrier / stats all-reduce through it) even
        # with ONE rank, so that the RCCL branch can be exercised on a 1-GPU box; never set by the driver.
        self.use_pg = self.world > 1 or bool(os.environ.get("BPP_BENCH_FORCE_PG"))
        self.device = None

    def early_device(self):
        """the rank's device is current before the library is loaded or anything launched"""
        torch = self.torch
        if self.world > 1 and torch.cuda.is_available() and not self.one_device and torch.cuda.device_count() > self.local_rank:
            torch.cuda.set_device(self.local_rank)

    def init_device(self):
        torch, dist = self.torch, self.dist
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device")
        if not self.one_device and torch.cuda.device_count() < self.world:
            raise SystemExit("--gpus %d but only %d HIP device(s) visible (one rank per GPU)" % (self.world, torch.cuda.device_count()))
        dev_index = 0 if self.one_device else self.local_rank
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        if self.use_pg:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.device, rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)

    def fence(self):
        self.torch.cuda.synchronize(self.device)
        if self.use_pg:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.device)

    def all_max(self, values):
        """element-wise maximum over ranks of a list of floats (a region takes as long as its slowest rank)"""
        if self.world == 1:
            return list(values)
        tm = self.torch.tensor(values, dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(tm, op=self.dist.ReduceOp.MAX)
        return tm.cpu().tolist()

    def all_sum_int(self, v):
        if self.world == 1:
            return int(v)
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())


PARITY_FIELDS = ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len")


def parity_gate(ctx, env, spec, actions, bins=256, lock_steps=24, seed=7):
    """BASELINE.md's "correctness gate before any number is reported", inside the bench run and OUTSIDE every timed region:
    the first `bins` bins of this rank's shard of the workload's real env -- the launch shape that is timed -- are
    stepped for `lock_steps` lock-steps (uniform-feasible draws fused into the step kernel; on one lock-step every fifth
    bin is sent to position 0, a forced failure for most) and every output of those bins -- observation, mask, reward,
    done, counter, ratio, episode return / length, finally the heightmaps -- is compared bit for bit with oracle/ (the C
    restatement, pinned to the reference: tests/) stepping the same bins with the same actions.  The oracle is the
    checker here, never the thing measured.  Returns the `parity` record; a mismatch aborts the bench line."""
    import numpy as np
    torch = ctx.torch
    from oracle import oracle as orc
    E = env.E
    n = min(int(bins), E)
    ref = orc.OracleEnv(spec["pool"], spec["size"], spec["rotation"], n, env_id_base=env.env_id_base, env_id_total=env.env_id_total)
    obs = env.reset()
    robs, rmask = ref.reset()
    mism = int(not np.array_equal(obs[:n].cpu().numpy(), robs)) + int(not np.array_equal(env.location_masks[:n].cpu().numpy(), rmask))
    env.sample_feasible(seed=seed, step=0, out=actions)
    episodes = 0
    bad = []
    for t in range(lock_steps):
        if t == lock_steps // 3:
            actions[::5] = 0
        a_host = actions[:n].cpu().numpy().copy()
        r = env.step_tensors(actions, sample=(seed, t + 1, actions))
        o = ref.step(a_host)
        for k in PARITY_FIELDS:
            got = getattr(r, k)[:n].cpu().numpy().reshape(o[k].shape)
            if not np.array_equal(got, o[k]):
                mism += 1
                bad.append((t, k))
        episodes += int(o["done"].sum())
    if not np.array_equal(env.hmap[:n].cpu().numpy(), ref.hmap):
        mism += 1
        bad.append((lock_steps, "hmap"))
    # Second block (VERDICT r5 #1c): the same bins under a COMPETENT policy -- oracle/policies.py's lowest-top heuristic,
    # computed on the host from the DEVICE's own observation and mask; the other bins of the launch keep their fused
    # uniform draws -- so that the gate also passes through bins that hold >= 20 boxes and bins packed completely,
    # states the uniform policy reaches in 0.03 % of its steps.
    from oracle.policies import lowest_top_actions
    deep_steps = 2 * lock_steps
    nd = n if spec["size"][0] * spec["size"][1] <= 100 else min(n, 64)      # (the host heuristic costs ~0.5 ms per 20x20 bin)
    obs = env.reset()
    robs, rmask = ref.reset()
    mism += int(not np.array_equal(obs[:n].cpu().numpy(), robs)) + int(not np.array_equal(env.location_masks[:n].cpu().numpy(), rmask))
    env.sample_feasible(seed=seed + 1, step=0, out=actions)
    cur_obs, cur_mask = obs[:nd].cpu().numpy(), env.location_masks[:nd].cpu().numpy()
    deep_env_steps = full_bins = deep_episodes = 0
    for t in range(deep_steps):
        actions[:nd] = torch.from_numpy(lowest_top_actions(cur_obs, cur_mask, spec["size"], spec["rotation"])).to(actions.device)
        a_host = actions[:n].cpu().numpy().copy()
        r = env.step_tensors(actions, sample=(seed + 1, t + 1, actions))
        o = ref.step(a_host)
        for k in PARITY_FIELDS:
            got = getattr(r, k)[:n].cpu().numpy().reshape(o[k].shape)
            if not np.array_equal(got, o[k]):
                mism += 1
                bad.append((lock_steps + 1 + t, k))
        cur_obs, cur_mask = r.obs[:nd].cpu().numpy(), r.mask[:nd].cpu().numpy()
        deep_env_steps += int((o["counter"][:nd] >= 20).sum())
        full_bins += int(((o["done"][:nd] != 0) & (o["ratio"][:nd] == 1.0)).sum())
        deep_episodes += int(o["done"][:nd].sum())
    if not np.array_equal(env.hmap[:n].cpu().numpy(), ref.hmap):
        mism += 1
        bad.append((lock_steps + 1 + deep_steps, "hmap"))
    deep = {"policy": "lowest resulting top, then smoothest surface, then lowest index (oracle/policies.py)", "lock_steps": deep_steps,
            "bins": nd, "env_steps_on_bins_with_20_or_more_boxes": deep_env_steps,
            "share_of_env_steps": round(deep_env_steps / float(nd * deep_steps), 4),
            "completely_packed_bins": full_bins, "episodes_finished_in_slice": deep_episodes}
    lock_steps += deep_steps
    rec = {"checked_bins": n, "lock_steps": lock_steps, "mismatches": mism, "episodes_finished_in_slice": episodes, "competent_policy_block": deep,
           "compared": list(PARITY_FIELDS) + ["hmap"], "against": "oracle/bpp_oracle.c (pinned to the reference by tests/)",
           "kernel": ctx.bpp._lib.launch_info(E, spec["size"], spec["rotation"])["kernel_name"], "bins_in_launch": E}
    total = ctx.all_sum_int(mism)
    if total:
        raise SystemExit("bench.py: PARITY GATE FAILED for %s on rank %d: %d mismatching (lock-step, output) pairs, first %r -- no "
                         "number is reported" % (spec["name"], ctx.rank, mism, bad[:4]))
    if ctx.world > 1:
        rec["checked_bins"] = n * ctx.world
        rec["ranks"] = ctx.world
    return rec


def run_workload(ctx, spec):
    """Every leg of ONE workload (a BASELINE config): parity gate, warm-up, the timed K-step regions with one output set,
    the same with the outputs rotated past the Infinity Cache, optionally the epsilon variant, the Python-driven loop."""
    args, torch, dist, bpp_amd = ctx.args, ctx.torch, ctx.dist, ctx.bpp
    device, world, rank = ctx.device, ctx.world, ctx.rank
    size, rotation, E, pool = spec["size"], spec["rotation"], spec["envs"], spec["pool"]
    A = size[0] * size[1]
    M = A * (2 if rotation else 1)
    stream = spec.get("stream")
    env = bpp_amd.BppVecEnv(E, size, enable_rotation=rotation, pool=None if stream else pool, device=device,
                            env_id_base=rank * E, env_id_total=world * E, stream=stream)
    stats = bpp_amd.EpisodeStats(device)
    actions = torch.empty(E, dtype=torch.int64, device=device)
    parity = None
    if spec.get("parity") and not stream:
        parity = parity_gate(ctx, env, spec, actions, lock_steps=spec.get("parity_steps", 24))
    env.reset()

    def lockstep(t):
        # actions for lock-step t were drawn inside lock-step t-1 (or by sample_feasible before the loop)
        return env.step_tensors(actions, sample=(1, t + 1, actions))

    # Warm-up and every timed region are driven by ONE native call each: finite pool -> bpp_rollout_uniform_sets
    # (every lock-step draws the next one's actions inside the step kernel, so a region of K lock-steps is exactly K
    # launches of the step kernel and nothing else); --stream -> bpp_rollout_uniform_stream (refills included).
    state = {"t": 0, "primed": False}

    def drive(n, sets=None, eps=0.0):
        if stream:
            env.rollout_uniform(seed=1, step0=state["t"], nsteps=n, actions=actions)
        else:
            env.rollout_uniform_sets(1, state["t"], n, actions, sets=sets, resume=state["primed"], eps=eps)
            state["primed"] = True
        state["t"] += n

    only_sets = None
    no_past_l3 = args.no_past_l3 or not spec.get("l3_seconds")
    if args.past_l3_only and not stream:
        sb = E * (16 * A + 4 * M + 29)
        only_sets = env.output_sets(max(3, int(1.07e9 / sb) + 1))
        no_past_l3 = True
    total = torch.zeros(4, dtype=torch.float64, device=device)         # job-level sums since the last clear (all ranks)
    local_total = torch.zeros(4, dtype=torch.float64, device=device)   # this rank's share of them

    def log_point():
        """One logging point of a training loop (main.py:194-): this rank's fixed-order reduction of its bins' accumulator
        rows, the path's ONLY collective -- the 32-byte all-reduce --, the record added to the job's running sums."""
        stats.collect(env)
        local_total.add_(stats.acc)
        stats.all_reduce()
        total.add_(stats.acc)
        stats.zero_()

    def clear_totals():
        total.zero_()
        local_total.zero_()

    def summary_of(t):
        a = t.cpu().tolist()
        n = max(a[3], 1.0)
        return {"episodes": int(a[3]), "mean_return": a[0] / n, "mean_ratio": a[1] / n, "mean_length": a[2] / n, "sums": a}

    drive(args.warmup, only_sets)
    log_point()   # also loads the few torch kernels the collection uses
    clear_totals()
    # timed region = EXACTLY K lock-steps: barrier + synchronize, clock, K lock-steps, synchronize, clock (the maximum
    # over ranks is taken afterwards); repeated `reps` times -- as often as it takes for the leg's seconds of timed GPU work,
    # whatever K is -- and the MEDIAN repetition reported.  The path's only collective -- the 32-byte statistics
    # all-reduce -- runs inside the timed region once per logging interval of LOG_INTERVAL lock-steps, the reference's
    # own cadence (main.py:194-: every log_interval = 10 updates of num_steps = 5 lock-steps).
    since_log = [0]
    region_events = []     # (start, end, kind, launches) HIP events on the launch stream around the K launches of timed regions
    region_count = {}

    def timed_region(sets=None, log=True, kind=None, eps=0.0):
        ctx.fence()
        # HIP events around the launches of every 16th region (and regions 2-4): recording a pair costs ~10 us of
        # host time, which a region of 20 lock-steps would feel
        kind = kind or ("past_l3" if sets is not None and sets is not only_sets else "headline")
        region_count[kind] = region_count.get(kind, 0) + 1
        sampled = 2 <= region_count[kind] <= 4 or region_count[kind] % 16 == 0      # (a leg's first region runs cold)
        if sampled:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if sampled:
            ev0.record()
        drive(args.steps, sets, eps)     # ONE native call: K launches of the step kernel (eps: + K tiny override launches), nothing else
        if sampled:
            ev1.record()
            region_events.append((ev0, ev1, kind, args.steps))
        since_log[0] += args.steps
        if log and since_log[0] >= LOG_INTERVAL:
            log_point()
            since_log[0] = 0
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    def repeat_for(target_s, sets=None, log=True, kind=None, eps=0.0):
        """Repetitions of the timed K-step region until their sum reaches `target_s` (three to begin with -- the first
        one runs cold --, then as many as the fastest of those says are still needed; every rank runs the same number)."""
        samples = [timed_region(sets, log, kind, eps) for _ in range(3)]
        if args.reps > 0:
            more = max(0, args.reps - 3)
        else:
            more = max(0, min(20000, int((target_s - sum(samples)) / max(min(samples), 1e-6)) + 1))
        if world > 1:
            rt = torch.tensor([more], dtype=torch.int64, device=device)
            dist.all_reduce(rt, op=dist.ReduceOp.MAX)
            more = int(rt.item())
        samples += [timed_region(sets, log, kind, eps) for _ in range(more)]
        return ctx.all_max(samples)

    samples = repeat_for(spec["gpu_seconds"], only_sets)
    dt = sorted(samples)[len(samples) // 2]
    log_point()   # whatever finished since the last logging point (outside the timed regions)

    # dominant kernel (bpp_step) launch duration: HIP events on the launch stream around n_ev lock-steps enqueued back
    # to back by ONE native call -- exactly n_ev launches of the step kernel between the two events
    n_ev = max(args.steps, 200)

    def event_timed(n, sets=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        e0.record()
        drive(n, sets)
        e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) / n

    def region_launch_ms(kind):
        """Average duration of the step kernel's launches INSIDE the timed regions: HIP events recorded on the launch
        stream right before the first and right behind the last of a region's K launches (nothing else is enqueued
        between them), median over the regions."""
        v = sorted(e0.elapsed_time(e1) / n for e0, e1, k, n in region_events if k == kind)
        return v[len(v) // 2] if v else None

    if stream:     # refill kernels run between / beside the lock-steps: event pairs around single launches
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(n_ev, 200))]
        env.sample_feasible(seed=1, step=state["t"], out=actions)
        for t, (e0, e1) in enumerate(evs):
            e0.record()
            lockstep(state["t"] + t)
            e1.record()
        torch.cuda.synchronize(device)
        state["t"] += len(evs)
        kern_avg_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
        kern_b2b_ms = None
    else:
        kern_b2b_ms = sorted(event_timed(n_ev, only_sets) for _ in range(3))[1]      # n_ev launches back to back, one event pair
        kern_avg_ms = region_launch_ms("headline") or kern_b2b_ms

    # ---- past the Infinity Cache: the same lock-steps writing R rotating output sets (> 1 GB span), so that no output
    # byte can stay in the 256 MiB L3 -- the HBM-only figure next to the L3-assisted one (one 133 MB set fits the L3)
    past = None
    if not stream and not no_past_l3:
        set_bytes = E * (16 * A + 4 * M + 29)
        R = max(3, int(1.07e9 / set_bytes) + 1)
        sets = env.output_sets(R)
        drive(2 * R, sets)
        s_l3 = repeat_for(spec["l3_seconds"], sets, log=False)
        dt_l3 = sorted(s_l3)[len(s_l3) // 2]
        kern_l3_ms = region_launch_ms("past_l3") or sorted(event_timed(n_ev, sets) for _ in range(3))[1]
        past = {"output_sets": R, "output_span_MB": round(R * set_bytes / 1e6, 1), "reps": len(s_l3),
                "ms_per_step": dt_l3 / args.steps * 1e3, "value": world * E * args.steps / dt_l3, "launch_us": kern_l3_ms * 1e3}
        del sets

    # same K lock-steps driven step by step from Python (what a Python RL loop pays per step)
    ctx.fence()
    t1 = time.perf_counter()
    env.sample_feasible(seed=1, step=state["t"], out=actions)
    for t in range(args.steps):
        lockstep(state["t"] + t)
    ctx.fence()
    dt_py = time.perf_counter() - t1
    state["t"] += args.steps
    state["primed"] = False
    log_point()
    summary = summary_of(total)
    shards = None
    if world > 1:   # every rank's own share beside the all-reduced record (tests: their sum is the record; rank r owns bins from r * E)
        mine = torch.cat([torch.tensor([float(env.env_id_base)], dtype=torch.float64, device=device), local_total])
        if ctx.backend != "nccl":
            mine = mine.cpu()
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        shards = [{"rank": r, "env_id_base": int(v[0].item()), "sums": v[1:].cpu().tolist()} for r, v in enumerate(parts)]

    # ---- SURVEY 8d's variant: epsilon = 1 % of the actions uniformly random over ALL entries (the failure path: more
    # episodes end, more bins restart per lock-step).  Same driver; every draw is followed by bpp_epsilon_override.
    eps = None
    if spec.get("eps_seconds") and not stream:
        drive(max(args.warmup, 30), None, args.eps)       # into the variant's own steady state (shorter episodes)
        log_point()
        clear_totals()
        s_eps = repeat_for(spec["eps_seconds"], None, log=True, kind="eps", eps=args.eps)
        dt_eps = sorted(s_eps)[len(s_eps) // 2]
        log_point()
        es = summary_of(total)
        eps = {"epsilon": args.eps, "value": world * E * args.steps / dt_eps, "ms_per_step": dt_eps / args.steps * 1e3, "reps": len(s_eps),
               "region_us_per_lock_step_on_stream": (region_launch_ms("eps") or 0.0) * 1e3 or None,
               "what": "every lock-step = the step kernel + one bpp_epsilon_override launch (one thread per bin)",
               "episodes_finished": es["episodes"], "mean_episode_length": round(es["mean_length"], 2), "mean_ratio": round(es["mean_ratio"], 4)}

    res = {"spec": spec, "A": A, "M": M, "E": E, "samples": samples, "dt": dt, "past": past, "kern_avg_ms": kern_avg_ms,
           "kern_b2b_ms": kern_b2b_ms, "dt_py": dt_py, "summary": summary, "shards": shards, "eps": eps, "parity": parity,
           "only_sets": only_sets is not None,
           "kernel_name": bpp_amd._lib.launch_info(E, size, rotation)["kernel_name"],
           "stream_spec": dict(env.stream_spec) if stream else None,
           "stream_overlap": bool(bpp_amd._lib.get_knobs()["stream_overlap"]) if stream else None}
    del env, stats, actions
    torch.cuda.synchronize(device)
    torch.cuda.empty_cache()
    return res


def roofline_block(ctx, res):
    """`achieved`/`frac`: ALGORITHMIC bytes (int32 heightmaps as in the reference's layout) per launch /
    launch duration with ONE output set (133 MB at the headline size: it stays in the 256 MiB Infinity Cache, so
    this figure is L3-assisted).  `frac_past_l3`: the same with the outputs rotated over > 1 GB -- the HBM-only
    figure.  `achieved_moved*`: the bytes the kernel really moves per launch (PMC counters at the L2's fabric
    side; the state is kept as bytes, so fewer than the algorithmic ones) / the same durations."""
    spec, A, M, E, past = res["spec"], res["A"], res["M"], res["E"], res["past"]
    size = spec["size"]
    kern_avg_ms, kern_b2b_ms = res["kern_avg_ms"], res["kern_b2b_ms"]
    b_alg = algorithmic_bytes_per_env_step(A, M)
    achieved = b_alg * E / (kern_avg_ms * 1e-3) / 1e9
    ev = profile_evidence("%dx%dx%d_rot%d_E%d" % (size + (int(spec["rotation"]), E)))
    traffic = ev.get("traffic_bytes") if ev else None
    moved = traffic / (kern_avg_ms * 1e-3) / 1e9 if traffic else None
    ach_l3 = b_alg * E / (past["launch_us"] * 1e-6) / 1e9 if past else None
    moved_l3 = traffic / (past["launch_us"] * 1e-6) / 1e9 if past and traffic else None
    return {"bound": "hbm", "kernel": "bpp_step (%s)" % res["kernel_name"],
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_is": "L3-assisted: one output set (%.0f MB) %s the 256 MiB Infinity Cache; frac_past_l3 is the HBM-only figure"
                       % (E * (16 * A + 4 * M + 29) / 1e6, "fits" if E * (16 * A + 4 * M + 29) < 256 * 2 ** 20 else "does NOT fit"),
            "achieved_past_l3": ach_l3, "frac_past_l3": (ach_l3 / HBM_PEAK_GBS) if ach_l3 else None,
            "launch_us_past_l3": past["launch_us"] if past else None,
            "traffic": traffic, "traffic_unit": "bytes per launch between L2 and fabric (rocprofv3 PMC, profiles/hbm_traffic.json)",
            "traffic_source": (ev or {}).get("source"),
            "achieved_moved": moved, "frac_moved": (moved / HBM_PEAK_GBS) if moved else None,
            "achieved_moved_past_l3": moved_l3, "frac_moved_past_l3": (moved_l3 / HBM_PEAK_GBS) if moved_l3 else None,
            "frac_moved_past_l3_of_hbm_copy_rate": (moved_l3 / HBM_ACHIEVABLE_GBS) if moved_l3 else None,
            "valu_utilisation": (ev or {}).get("valu_utilisation"),
            "limiter": limiter(moved_l3, (ev or {}).get("valu_utilisation")),
            "bytes_per_env_step": b_alg, "launch_us": kern_avg_ms * 1e3,
            "launch_us_is": ("HIP events on the launch stream around the K step-kernel launches of every 16th timed region, median "
                             "over the regions of elapsed / K (at small K this includes the idle gap in front of a region's "
                             "first kernel; launch_us_back_to_back = one event pair around >= 200 queued launches)" if not spec.get("stream") else
                             "HIP event pairs around single launches (refill kernels run beside the lock-steps)"),
            "launch_us_back_to_back": kern_b2b_ms * 1e3 if kern_b2b_ms else None}


def workload_text(spec, res, only_sets=False):
    size = tuple(spec["size"])
    return "%dx%dx%d bin, CUT-2 sequences%s, %d envs per MI355X, uniform-random-feasible policy%s" % (
        size + (" + rotation" if spec["rotation"] else "", spec["envs"],
                " [--past-l3-only: EVERY region writes rotating output sets]" if only_sets else ""))


def n1_leg(ctx, spec, seconds=0.6):
    """N > 1 only: the SAME workload on rank 0 ALONE, in the same run on the same box (the other ranks wait at the fence),
    so that the line carries its own N = 1 figure and the scaling efficiency can be read off it: `n1_value_same_run`.
    Same timed region as the job's (K lock-steps by one native call between synchronisations), median of the repetitions
    that fit `seconds`; no collective inside (there is nobody to talk to)."""
    args, torch = ctx.args, ctx.torch
    ctx.fence()
    out = None
    if ctx.rank == 0:
        E = spec["envs"]
        env = ctx.bpp.BppVecEnv(E, spec["size"], enable_rotation=spec["rotation"], pool=spec["pool"], device=ctx.device,
                                env_id_base=0, env_id_total=E)
        actions = torch.empty(E, dtype=torch.int64, device=ctx.device)
        env.reset()
        env.rollout_uniform_sets(1, 0, args.warmup, actions, resume=False)
        t, samples, spent = args.warmup, [], 0.0
        while len(samples) < 3 or (spent < seconds and len(samples) < 20000):
            torch.cuda.synchronize(ctx.device)
            t0 = time.perf_counter()
            env.rollout_uniform_sets(1, t, args.steps, actions, resume=True)
            torch.cuda.synchronize(ctx.device)
            samples.append(time.perf_counter() - t0)
            spent += samples[-1]
            t += args.steps
        dt = sorted(samples)[len(samples) // 2]
        out = {"value": E * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "reps": len(samples),
               "what": "rank 0 alone on its GPU, %d bins, the other %d rank(s) idle at a barrier; one output set" % (E, ctx.world - 1)}
    ctx.fence()
    return out


def main():
    args = parse()
    spawned = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not spawned and (args.gpus > 1 or args.launcher == "spawn"):
        sys.exit(self_launch(args))
    ctx = Ctx(args)
    bpp_amd, world, rank = ctx.bpp, ctx.world, ctx.rank
    size = tuple(args.size)
    E = args.envs
    # (rank 0 of a launcher-started job forks the CPU baseline's workers further down: it touches its GPU -- device 0, the
    # default anyway -- only after that; every other rank makes its device current before the library is loaded)
    will_fork = ctx.rank == 0 and ctx.world > 1 and not args.no_cpu_baseline and not os.environ.get("BPP_BENCH_CPU_BASELINE_FILE")
    if not will_fork:
        ctx.early_device()
    stream = dict(bound=(2, 5), seed=0, depth=args.stream_depth, refill_every=args.stream_refill, rng=args.stream_rng,
                  cache={"auto": None, "on": True, "off": False}[args.stream_cache]) if args.stream else None
    if args.pool_file:
        # .npz ([P][T][4] `pool`) or a reference dataset/*.pt, played as the reference's LoadBoxCreator plays it (first
        # episode = trajectory 1; rows end in the terminator: the reference's literal (10,10,10) for its own 10x10x10
        # sets, the bin size for any other bin)
        pool = bpp_amd.sequences.from_dataset(args.pool_file, size, terminator=(10, 10, 10) if size == (10, 10, 10) else size)
    else:
        pool = bpp_amd.sequences.cut2_pool(size, args.pool, seed=0)   # identical on every rank (cpu_baseline uses it too)
    headline = size == (10, 10, 10) and not args.rotation and E == 65536 and not args.stream
    # The line's own workload first; when that is BASELINE's headline (config 2: the defaults), the other single-GPU
    # BASELINE configs follow in the same run -- config 3 (rotation), config 4 (20x20x20, 32 768 bins), and SURVEY 8d's
    # PRIMARY pool of config 2 (the reference's dataset/cut_2.pt) -- each with its own parity gate and roofline block.
    main_spec = {"name": "%dx%dx%d%s" % (size + ("_rot" if args.rotation else "",)), "size": size, "rotation": args.rotation, "envs": E,
                 "pool": pool, "pool_source": args.pool_file or "generated CUT-2 (sequences.cut2_pool, seed 0)", "stream": stream,
                 "gpu_seconds": args.gpu_seconds, "l3_seconds": args.gpu_seconds * 2.0 / 3.0, "parity": not args.no_parity,
                 "parity_steps": 24 if size[0] * size[1] <= 100 else 48,
                 "eps_seconds": args.eps_seconds if headline and not args.only_headline and args.eps > 0 else 0.0}
    extra_specs = []
    if headline and not args.only_headline and not args.pool_file and not args.past_l3_only:
        xs, xl = args.extra_seconds, args.extra_seconds * 0.7
        extra_specs.append({"name": "10x10x10_rot", "size": (10, 10, 10), "rotation": True, "envs": 65536, "pool": pool,
                            "pool_source": main_spec["pool_source"], "gpu_seconds": xs, "l3_seconds": xl, "parity": not args.no_parity,
                            "parity_steps": 24})
        extra_specs.append({"name": "20x20x20", "size": (20, 20, 20), "rotation": False, "envs": 32768,
                            "pool": bpp_amd.sequences.cut2_pool((20, 20, 20), 2048, seed=0),
                            "pool_source": "generated CUT-2 (sequences.cut2_pool((20,20,20), 2048, seed 0))",
                            "gpu_seconds": xs, "l3_seconds": xl, "parity": not args.no_parity, "parity_steps": 48})
        ds = os.path.join(ROOT, "tests", "golden", "cut2_dataset_10.npz")
        if os.path.exists(ds):
            extra_specs.append({"name": "10x10x10_dataset_cut2", "size": (10, 10, 10), "rotation": False, "envs": 65536,
                                "pool": bpp_amd.sequences.from_dataset(ds, (10, 10, 10), terminator=(10, 10, 10)),
                                "pool_source": "tests/golden/cut2_dataset_10.npz = the reference's dataset/cut_2.pt (2100 trajectories, "
                                               "LoadBoxCreator's order): SURVEY 8d's PRIMARY pool of config 2",
                                "gpu_seconds": 0.6 * xs, "l3_seconds": 0.5 * xl, "parity": not args.no_parity, "parity_steps": 24})
    cpu_base = None
    handed = os.environ.get("BPP_BENCH_CPU_BASELINE_FILE")
    if rank == 0 and not args.no_cpu_baseline:
        if handed and os.path.exists(handed):          # timed by the self-launching parent before the ranks started
            cpu_base = json.load(open(handed))
        else:
            # before the HIP runtime is initialised in this process: the baseline forks one worker per core.  With N > 1
            # ranks started by a launcher this is rank 0's job as well (the other ranks wait in the rendezvous of
            # init_process_group below, blocked, not spinning): the N = 2 / 4 / 8 lines carry the reference timed on
            # the same box in the same run, as north_star asks.
            try:
                cpu_base = cpu_baseline(pool, size, args.rotation, args.cpu_seconds)
            except Exception as exc:  # noqa: BLE001 -- the baseline leg must never take the GPU measurement down
                cpu_base = {"value": None, "unit": "env steps/s", "cores": 0, "kind": "port",
                            "sample": "cpu baseline failed: %r" % (exc,)}
            if world > 1:
                cpu_base["timed_by"] = "rank 0 before any rank touched its GPU (the other %d ranks waited in the rendezvous)" % (world - 1)
                ctx.early_device()
    ctx.init_device()
    res = run_workload(ctx, main_spec)
    extras = [run_workload(ctx, sp) for sp in extra_specs]
    n1 = n1_leg(ctx, main_spec) if world > 1 and not args.stream else None

    if rank == 0:
        samples, dt, past, summary = res["samples"], res["dt"], res["past"], res["summary"]
        metric = "env steps/sec (whole node), %dx%dx%d bin%s, %d envs per GPU%s; bit-exact mask vs ref" % (
            size + (" + rotation" if args.rotation else "", E,
                    ", endless device-generated CUT-2 supply (%s)" % ("exact CPython MT19937 streams" if args.stream_rng == "mt19937" else
                                                                       "counter-based generator") if args.stream else ""))
        if headline:    # BASELINE.json's metric string belongs to its own workload only
            try:
                metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            except Exception:
                pass
        sspec = res["stream_spec"]
        out = {
            "metric": metric,
            "value": world * E * args.steps / dt,
            "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "reps": len(samples), "timed_gpu_work_ms": sum(samples) * 1e3, "stats_all_reduce_every_lock_steps": LOG_INTERVAL,
            "rep_ms_per_step_min_median_max": [min(samples) / args.steps * 1e3, dt / args.steps * 1e3, max(samples) / args.steps * 1e3],
            "ms_per_step": dt / args.steps * 1e3,
            "python_loop_ms_per_step": res["dt_py"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            # the same lock-steps with the outputs rotated over > 1 GB: nothing stays in the 256 MiB Infinity Cache --
            # THE HBM-ONLY FIGURE (value is L3-assisted at this size: one 133 MB output set stays in the cache)
            "value_past_l3": past["value"] if past else None,
            "past_l3": past,
            "config": {"workload": workload_text(main_spec, res, res["only_sets"]),
                       "envs_per_gpu": E, "total_envs": world * E, "pool_sequences": int(pool.shape[0]),
                       "pool_source": ("device stream: %s, ring of %d rows, refill every %d lock-steps%s (bpp_stream)"
                                       % ("random.Random(g) per bin" if args.stream_rng == "mt19937" else
                                          "counter-based generator keyed by (seed, bin, episode), the reference's cutting algorithm",
                                          args.stream_depth, args.stream_refill,
                                          (" beside the lock-steps" if args.stream_depth >= 2 * args.stream_refill + (4 if sspec["cache"] else 3)
                                           and res["stream_overlap"] else "") +
                                          (", row cache (bpp_batch.seq_cache)" if sspec["cache"] else ", no row cache")) if args.stream
                                       else main_spec["pool_source"]),
                       "sharding": "bins by global id, %d rank(s); 32-byte stats all-reduce only (%s%s)"
                                   % (world, "RCCL" if ctx.backend == "nccl" else ctx.backend,
                                      "" if ctx.use_pg else ", no process group at 1 rank"),
                       "launcher": ("self-launched torch.distributed.run" if os.environ.get("BPP_BENCH_CHILD") else
                                    "torch.distributed.run" if spawned else "direct"),
                       "episodes_finished": summary["episodes"], "mean_ratio": round(summary["mean_ratio"], 4),
                       "mean_episode_length": round(summary["mean_length"], 2),
                       # [return sum, final-ratio sum, length sum, episodes] of the whole job (all-reduced at every logging
                       # point) and, with more than one rank, every rank's own share of it (its first global bin id beside it)
                       "episode_sums": summary["sums"], "shards": res["shards"]},
            "roofline": roofline_block(ctx, res),
        }
        # BASELINE.md section 3's gate, run inside this bench run (outside the timed regions) for every workload of the line
        gates = [(r["spec"]["name"], r["parity"]) for r in [res] + extras if r["parity"]]
        if gates:
            out["parity"] = {"checked_bins": sum(g["checked_bins"] for _, g in gates), "lock_steps": sum(g["lock_steps"] for _, g in gates),
                             "mismatches": sum(g["mismatches"] for _, g in gates),
                             "gate": "first bins of the timed env itself vs oracle/ (the pinned C restatement) on identical sequences and "
                                     "actions, every output compared bit for bit; a mismatch aborts the line",
                             "per_workload": {n: g for n, g in gates}}
        if res["eps"]:
            out["epsilon_variant"] = res["eps"]
        if extras:
            out["configs"] = {}
            for r in extras:
                sp, p2, s2 = r["spec"], r["past"], r["summary"]
                out["configs"][sp["name"]] = {
                    "workload": workload_text(sp, r), "pool_source": sp["pool_source"], "pool_sequences": int(sp["pool"].shape[0]),
                    "value": world * r["E"] * args.steps / r["dt"], "value_past_l3": p2["value"] if p2 else None,
                    "ms_per_step": r["dt"] / args.steps * 1e3, "ms_per_step_past_l3": p2["ms_per_step"] if p2 else None,
                    "reps": len(r["samples"]), "timed_gpu_work_ms": sum(r["samples"]) * 1e3, "reps_past_l3": p2["reps"] if p2 else None,
                    "launch_us": r["kern_avg_ms"] * 1e3, "launch_us_past_l3": p2["launch_us"] if p2 else None,
                    "python_loop_ms_per_step": r["dt_py"] / args.steps * 1e3,
                    "episodes_finished": s2["episodes"], "mean_ratio": round(s2["mean_ratio"], 4), "mean_episode_length": round(s2["mean_length"], 2),
                    "roofline": roofline_block(ctx, r), "parity": r["parity"]}
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        if n1 is not None:
            out["n1_value_same_run"] = n1["value"]
            out["n1_same_run"] = n1
            out["scaling_efficiency_same_run"] = out["value"] / (world * n1["value"])
            out["roofline"]["per_gpu"] = True      # rank 0's kernel on rank 0's GPU; every rank runs the same launch on its own shard
        print(json.dumps(out), flush=True)
    if ctx.use_pg:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
