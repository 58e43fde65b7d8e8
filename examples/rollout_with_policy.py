#!/usr/bin/env python3
"""Tensor-native rollout loop with a policy in the loop.

A CNN shaped like the reference's CNNPro actor (acktr/model.py:265-323: 5 x conv3x3(64) trunk, 1x1 conv(8),
linear(hidden), linear(actions)) with random weights stands in for the trained policy; everything else is
this package: observations and masks from BppVecEnv.step_tensors, action selection by bpp_masked_act
(softmax(x - 14(1-mask)) + 1e-5, acktr/distributions.py:71-84), episode statistics from the step kernel.
Nothing leaves the GPU until the summary is printed.

    python examples/rollout_with_policy.py --envs 16384 --steps 200
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
    ap.add_argument("--stream", action="store_true",
                    help="endless item supply generated on the device (every bin its own random.Random, nothing replayed) "
                         "instead of a pool of 4096 sequences")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    size = (10, 10, 10)
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

    def one_step(t):
        nonlocal obs, masks
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            logits = policy(obs)
        action, logp = bpp_amd.masked_act(logits.float(), masks, seed=7, step=t)
        res = envs.step_tensors(action)
        obs, masks = res.obs, res.mask
        return res

    for t in range(10):
        one_step(t)
    envs.episode_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(10 + t)
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
    print("policy in the loop: %.2f M env steps/s (%.1f us per lock-step of %d bins); environment alone %.1f us; "
          "episodes %d, mean space utilisation %.3f, mean length %.1f"
          % (args.envs * args.steps / dt / 1e6, dt / args.steps * 1e6, args.envs, dt_env / args.steps * 1e6,
             s["episodes"], s["mean_ratio"], s["mean_length"]))


if __name__ == "__main__":
    main()
