#!/usr/bin/env python3
"""The drop-in at the REFERENCE'S OWN SCALE: what a user gets who makes INTEGRATION.md 4.1's swap
at `--num-processes 16 ... 1024` (and at 65 536 bins), measured per env count:

  * `step_us`                 latency of one reference-shaped `envs.step(action)` (obs tensor on the device, CPU reward
                              [E,1], numpy bool done, lazily built infos), env built by `bpp_amd.make_vec_envs` (its
                              defaults: fresh outputs, eager infos) and by plain `BppVecEnv`;
  * `mask_helper_us_per_row`  one `bpp_amd.get_possible_position(observation, container_size)` call -- what main.py:163-169
                              does once per observation ROW (a kernel launch + a device-to-host copy per row);
  * `loop_literal`            env steps/s of a loop of main.py:148-174's shape with ONLY the 3-line swap of INTEGRATION 4.1:
                              `envs.step`, the infos scan dict by dict, the per-row mask helper, the mask / bad-mask lists
                              (torch.multinomial over the masks stands where actor_critic.act does);
  * `loop_batched_masks`      the same loop with ONE more line changed -- `location_masks = envs.location_masks` instead of
                              the per-row helper loop -- and the finished bins read through `infos.episodes()`;
  * `loop_tensor_native`      INTEGRATION 4.2: `step_tensors`, nothing leaves the device.

The reference's own number for the first loop shape (R1: its ShmemVecEnv plumbing with 16 forked workers on the same
box's host cores) is `cpu_baseline.reference_as_is_R1` of the bench.py line of the same run.

    python tools/bench_dropin_scale.py [--envs 16,64,1024,65536] [--seconds 1.0]
One JSON line.
"""
import argparse
import json
import os
import sys
import time
import types
from collections import deque

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, seconds, min_iters=5):
    """calls of fn() per second over >= `seconds` (after 3 warm-up calls); returns (iterations, elapsed)."""
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        if n >= min_iters and time.perf_counter() - t0 >= seconds:
            break
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", default="16,64,1024,65536")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--rotation", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bpp_amd
    size = (10, 10, 10)
    device = torch.device("cuda:0")
    args = types.SimpleNamespace(container_size=size, enable_rotation=a.rotation, data_type="cut2",
                                 box_size_set=[(i, j, k) for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)])
    helper = bpp_amd.get_rotation_mask if a.rotation else bpp_amd.get_possible_position
    out = {"size": size, "rotation": a.rotation, "rows": []}
    for E in [int(v) for v in a.envs.split(",")]:
        row = {"envs": E}
        # ---- step() latency -------------------------------------------------------------------------------------------------
        for tag, make in (("factory", lambda: bpp_amd.make_vec_envs("Bpp-v0", 1, E, 1.0, None, device, False, args=args)),
                          ("plain", lambda: bpp_amd.BppVecEnv(E, size, enable_rotation=a.rotation, device=device,
                                                              pool=bpp_amd.make_pool(size, "cut2", args.box_size_set, a.rotation, seed=1)))):
            envs = make()
            envs.reset()
            act = envs.sample_feasible(seed=1, step=0)
            st = {"t": 0}

            def one_step():
                st["t"] += 1
                envs.step(act, sample=(1, st["t"], act))

            n, dt = timed(one_step, a.seconds)
            row["step_us_%s" % tag] = round(dt / n * 1e6, 1)
            del envs
        # ---- the per-row helper ----------------------------------------------------------------------------------------------
        envs = bpp_amd.make_vec_envs("Bpp-v0", 1, E, 1.0, None, device, False, args=args)
        obs = envs.reset()
        k = {"i": 0}

        def one_mask():
            k["i"] = (k["i"] + 1) % E
            helper(obs[k["i"]], args.container_size)

        n, dt = timed(one_mask, min(a.seconds, 0.5))
        row["mask_helper_us_per_row"] = round(dt / n * 1e6, 1)

        # ---- main.py:148-174's shape, the 3-line swap only ------------------------------------------------------------------
        episode_rewards, episode_ratio = deque(maxlen=10), deque(maxlen=10)
        state = {"obs": obs, "masks": None}

        def row_masks(o):
            location_masks = []
            for observation in o:                                                   # main.py:163-169
                location_masks.append(helper(observation, args.container_size))
            return torch.FloatTensor(np.array(location_masks)).to(device)

        def literal():
            if state["masks"] is None:
                state["masks"] = row_masks(state["obs"])
            action = torch.multinomial(state["masks"], 1)                           # (stands where actor_critic.act does)
            o, reward, done, infos = envs.step(action)                             # main.py:158
            for i in range(len(infos)):                                             # main.py:159-162
                if 'episode' in infos[i].keys():
                    episode_rewards.append(infos[i]['episode']['r'])
                    episode_ratio.append(infos[i]['ratio'])
            state["masks"] = row_masks(o)
            masks = torch.FloatTensor([[0.0] if done_ else [1.0] for done_ in done])          # main.py:172-173
            bad_masks = torch.FloatTensor([[0.0] if 'bad_transition' in info.keys() else [1.0] for info in infos])
            state["obs"] = o
            return masks, bad_masks

        if E <= 4096:
            n, dt = timed(literal, a.seconds, min_iters=3)
            row["loop_literal_env_steps_per_s"] = round(E * n / dt, 1)
            row["loop_literal_ms_per_lockstep"] = round(dt / n * 1e3, 3)
        else:
            row["loop_literal_env_steps_per_s"] = None      # E per-row helper calls + E dicts per lock-step: seconds per lock-step
        del envs

        # ---- one more line: the env's own batched masks, the finished bins' infos as arrays -------------------------------
        envs = bpp_amd.make_vec_envs("Bpp-v0", 1, E, 1.0, None, device, False, args=args)
        envs.reset()
        state = {"masks": envs.location_masks}

        def batched():
            action = torch.multinomial(state["masks"], 1)
            o, reward, done, infos = envs.step(action)
            ep = infos.episodes()                                                   # main.py:159-162 on arrays
            episode_rewards.extend(ep["r"].tolist())
            episode_ratio.extend(ep["ratio"].tolist())
            state["masks"] = envs.location_masks                                    # replaces main.py:163-169
            masks = torch.from_numpy(1.0 - done.astype(np.float32)).unsqueeze(1)
            return masks

        n, dt = timed(batched, a.seconds)
        row["loop_batched_masks_env_steps_per_s"] = round(E * n / dt, 1)
        row["loop_batched_masks_us_per_lockstep"] = round(dt / n * 1e6, 1)
        del envs

        # ---- tensor-native ---------------------------------------------------------------------------------
        envs = bpp_amd.BppVecEnv(E, size, enable_rotation=a.rotation, device=device,
                                 pool=bpp_amd.make_pool(size, "cut2", args.box_size_set, a.rotation, seed=1))
        envs.reset()
        act = envs.sample_feasible(seed=1, step=0)
        st = {"t": 0}

        def native():
            st["t"] += 1
            envs.step_tensors(act, sample=(1, st["t"], act))

        n, dt = timed(native, a.seconds, min_iters=50)
        row["loop_tensor_native_env_steps_per_s"] = round(E * n / dt, 1)
        row["loop_tensor_native_us_per_lockstep"] = round(dt / n * 1e6, 1)
        del envs
        torch.cuda.empty_cache()
        out["rows"].append(row)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
