#!/usr/bin/env python3
"""Timing of bpp_masked_act against the equivalent PyTorch ops of Policy.act
(acktr/model.py:56-68, acktr/distributions.py:71-84).  HBM-bound: reads 2 * 4M bytes per bin."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bpp_amd

E = 65536
out = {}
for M in (100, 200, 400):
    x = torch.randn(E, M, device="cuda")
    m = (torch.rand(E, M, device="cuda") < 0.3).float()

    def fused(t):
        return bpp_amd.masked_act(x, m, seed=1, step=t)

    def eager(t):
        lx = torch.softmax(x - (1.0 - m) * 14.0, dim=-1) + 1e-5
        d = torch.distributions.Categorical(probs=lx)
        a = d.sample()
        return a.unsqueeze(-1), d.log_prob(a).unsqueeze(-1)

    res = {}
    for name, fn in (("fused_hip", fused), ("torch_eager", eager)):
        for t in range(5):
            fn(t)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for t, (a, b) in enumerate(evs):
            a.record()
            fn(t)
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        res[name + "_us"] = round(sum(ts[5:-5]) / len(ts[5:-5]), 1)
    res["algorithmic_GBps_fused"] = round(E * (8 * M + 12) / (res["fused_hip_us"] * 1e-6) / 1e9, 1)
    out["M=%d" % M] = res
print(json.dumps({"E": E, "results": out}))
