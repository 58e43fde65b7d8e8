#!/usr/bin/env python3
"""Tensor-native rollout loop with a policy in the loop.

A CNN shaped like the reference's CNNPro actor (acktr/model.py:265-323: 5 x conv3x3(64) trunk, 1x1 conv(8),
linear(hidden), linear(actions)) with random weights stands in for the trained policy; everything else is
this package: observations and masks from BppVecEnv.step_tensors, action selection by bpp_masked_act
(softmax(x - 14(1-mask)) + 1e-5, acktr/distributions.py:71-84), episode statistics from the step kernel.
Nothing leaves the GPU until the summary is printed.

    python examples/rollout_with_policy.py --envs 16384 --steps 200
    python examples/rollout_with_policy.py --envs 64 --steps 2000 --graph      # the reference's scale: one HIP-graph launch per lock-step

--graph: policy forward + bpp_masked_act_counter + the fused environment step + the step counter's increment are captured ONCE
in a HIP graph (torch.cuda.CUDAGraph) and replayed: at 16 ... 1 024 bins a lock-step is ~25 kernel launches of a few
microseconds each, i.e. launch-bound; the graph turns them into one launch.  The sampler's (seed, step) live in device memory
(`counter`), so every replay draws afresh.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import bpp_amd


class Actor(nn.Module):
    def __init__(self, side, n_actions, hidden=256):
        super().__init__()
        layers, c = [], 4
        for _ in range(5):
            layers += [nn.Conv2d(c, 64, 3, padding=1), nn.ReLU()]
            c = 64
        self.share = nn.Sequential(*layers)
        self.actor = nn.Sequential(nn.Conv2d(64, 8, 1), nn.ReLU(), nn.Flatten(), nn.Linear(8 * side * side, hidden),
                                   nn.ReLU())
        self.linear = nn.Linear(hidden, n_actions)
        self.side = side

    def forward(self, obs):
        x = obs.reshape(-1, 4, self.side, self.side)       # CNNPro.forward, acktr/model.py:315-316
        return self.linear(self.actor(self.share(x)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rotation", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="run the policy under bf16 autocast")
    ap.add_argument("--graph", action="store_true", help="capture one lock-step (policy, masked_act, env step) in a HIP graph and replay it")
    ap.add_argument("--stream", action="store_true",
                    help="endless item supply generated on the device (every bin its own random.Random, nothing replayed) "
                         "instead of a pool of 4096 sequences")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    size = (10, 10, 10)
    if args.graph and args.stream:
        raise SystemExit("--graph replays ONE captured lock-step; the stream supply's refills are scheduled by host code between lock-steps")
    if args.stream:
        envs = bpp_amd.BppVecEnv(args.envs, size, enable_rotation=args.rotation, device=dev,
                                 stream=dict(bound=(2, 5), seed=0, depth=16, refill_every=8))
    else:
        pool = bpp_amd.sequences.cut2_pool(size, 4096, seed=0)
        envs = bpp_amd.BppVecEnv(args.envs, size, enable_rotation=args.rotation, pool=pool, device=dev)
    policy = Actor(10, envs.action_space.n).to(dev).eval()
    obs = envs.reset()
    masks = envs.location_masks
    stats = bpp_amd.EpisodeStats(dev)

    counter = torch.tensor([7, 0], dtype=torch.int64, device=dev)                 # (seed, step) of the sampler, read by the kernel
    act_out = (torch.empty((args.envs, 1), dtype=torch.int64, device=dev), torch.empty((args.envs, 1), dtype=torch.float32, device=dev))

    def one_step(t=None):
        # obs / masks are the env's own output buffers (fresh_outputs=False): the same tensors every lock-step
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            logits = policy(obs)
        action, logp = bpp_amd.masked_act(logits.float(), masks, counter=counter, out=act_out)
        res = envs.step_tensors(action)
        counter[1:].add_(1)
        return res

    for t in range(10):
        one_step()
    graph = None
    if args.graph:
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # (warm-up on a side stream, as torch's graph recipe asks)
            for t in range(3):
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            one_step()
    envs.episode_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
        for t in range(args.steps):
            graph.replay()
    else:
        for t in range(args.steps):
            one_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = stats.collect(envs).summary()
    # env-only share of the loop, same number of steps without the network
    a = envs.sample_feasible(1, 0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for t in range(args.steps):
        envs.step_tensors(a, sample=(1, t + 1, a))
    torch.cuda.synchronize()
    dt_env = time.perf_counter() - t1
    print("policy in the loop%s: %.2f M env steps/s (%.1f us per lock-step of %d bins); environment alone %.1f us; "
          "episodes %d, mean space utilisation %.3f, mean length %.1f"
          % (" (ONE HIP-graph launch per lock-step)" if graph is not None else "", args.envs * args.steps / dt / 1e6, dt / args.steps * 1e6, args.envs, dt_env / args.steps * 1e6,
             s["episodes"], s["mean_ratio"], s["mean_length"]))


if __name__ == "__main__":
    main()
