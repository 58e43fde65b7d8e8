#!/usr/bin/env python3
"""Cost of the REFERENCE-SHAPED step() (acktr/envs.py:170-193 types: obs on the device, reward CPU float32 [E,1],
done numpy bool, lazily built infos) per lock-step, next to the tensor-native step_tensors().

Action source: `step` draws the next action inside the step kernel (sample=...: what stands where a policy would is
then nothing but the policy); `step+sampler_launch` keeps round 3's separate bpp_sample_feasible launch per step.
Variants: what the loop touches afterwards -- nothing; the finished bins' infos through one dict (first access pays the
gather); ALL finished bins scanned the way main.py:159-162 scans them (a Python loop over done_indices());
the same numbers as arrays (infos.episodes()); one running bin's info.

    python tools/bench_dropin_step.py [--envs 65536] [--steps 300]
One JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--quick", action="store_true", help="only tensors / step / step+episodes_arrays")
    args = ap.parse_args()
    import torch
    import bpp_amd
    size = (10, 10, 10)
    pool = bpp_amd.sequences.cut2_pool(size, 8192, seed=0)
    out = {"envs": args.envs, "steps": args.steps}
    # (fresh_outputs, eager_infos, spin_wait): the last one False = step_wait() synchronises the stream (rounds 3 - 4) instead of
    # spinning on the step's completion word
    for fresh, eager, spin in ((False, False, True), (False, True, True), (True, True, True), (False, False, False), (False, True, False)):
        env = bpp_amd.BppVecEnv(args.envs, size, pool=pool, fresh_outputs=fresh, eager_infos=eager)
        env.spin_wait = spin
        env.reset()
        a = env.sample_feasible(seed=1, step=0)
        host = {"async": 0.0, "wait": 0.0, "n": 0}

        from collections import deque
        episode_rewards, episode_ratio = deque(maxlen=10), deque(maxlen=10)     # main.py:110-111

        def run(kind, n, t0):
            nonlocal a
            for t in range(t0, t0 + n):
                if kind == "tensors":
                    env.step_tensors(a, sample=(1, t + 1, a))
                    continue
                if kind == "step+sampler_launch":
                    obs, rew, done, infos = env.step(a)
                    env.sample_feasible(seed=1, step=t + 1, out=a)
                    continue
                h0 = time.perf_counter()
                env.step_async(a, sample=(1, t + 1, a))
                h1 = time.perf_counter()
                obs, rew, done, infos = env.step_wait()
                h2 = time.perf_counter()
                host["async"] += h1 - h0
                host["wait"] += h2 - h1
                host["n"] += 1
                if kind == "step+one_finished_info":
                    idx = infos.done_indices()
                    if idx.size:
                        infos[int(idx[0])]["episode"]["r"]
                elif kind == "step+scan_finished_like_main_py":
                    for i in infos.done_indices():                      # main.py:159-162 over the finished bins
                        if "episode" in infos[i].keys():
                            episode_rewards.append(infos[i]["episode"]["r"])
                            episode_ratio.append(infos[i]["ratio"])
                elif kind == "step+episodes_arrays":
                    ep = infos.episodes()
                    ep["r"].sum(), ep["ratio"].sum()
                elif kind == "step+scan_episodes_arrays_like_main_py":
                    ep = infos.episodes()                               # main.py:159-162 on the arrays: two deque extends per step
                    episode_rewards.extend(ep["r"].tolist())
                    episode_ratio.extend(ep["ratio"].tolist())
                elif kind == "step+running_info":
                    infos[0]["ratio"]

        kinds = ["tensors", "step", "step+sampler_launch", "step+one_finished_info", "step+episodes_arrays", "step+scan_episodes_arrays_like_main_py", "step+running_info",
                 "step+scan_finished_like_main_py"]
        if args.quick or not spin:
            kinds = ["tensors", "step", "step+episodes_arrays"]
        for kind in kinds:
            n = args.steps if kind != "step+scan_finished_like_main_py" else max(10, args.steps // 10)
            run(kind, 30 if n > 30 else 3, 0)
            torch.cuda.synchronize()
            host.update(**{"async": 0.0, "wait": 0.0, "n": 0})
            t0 = time.perf_counter()
            run(kind, n, 30)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / n * 1e6
            tag = ("_fresh_outputs" if fresh else "") + ("_eager_infos" if eager else "") + ("" if spin else "_stream_synchronise")
            key = "%s%s_us_per_lockstep" % (kind, tag)
            out[key] = round(us, 1)
            if kind == "step":
                out["step%s_host_us_in_step_async" % tag] = round(host["async"] / host["n"] * 1e6, 1)
                out["step%s_host_us_in_step_wait" % tag] = round(host["wait"] / host["n"] * 1e6, 1)
        del env
        torch.cuda.empty_cache()
    out["note"] = ("tensors / step: ONE launch per lock-step (the step kernel draws the next action itself); step = step kernel (also "
                   "writing reward + done, 5 bytes per bin, into page-locked host memory) + a completion word the host spins on (bpp_mark / bpp_wait_mark; '_stream_synchronise': hipStreamSynchronize instead); +sampler_launch: a separate "
                   "bpp_sample_feasible launch per step as in round 3; +one_finished_info / +episodes_arrays: the finished bins' (r, l, ratio, "
                   "counter): without eager_infos bpp_gather_finished when somebody looks (one launch + a second sync); with eager_infos "
                   "(make_vec_envs' setting) the compaction is enqueued behind every step kernel and step_wait()'s one sync covers it; "
                   "+scan_episodes_arrays_like_main_py: main.py:159-162 on those arrays (two deque extends per step); +scan_finished_like_main_py: a Python loop "
                   "over the ~11 % of bins that finished, two dict reads each; +running_info: counter / ratio of all bins (12 B per bin)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
