#!/usr/bin/env python3
"""Probe: duration of consecutive rollout launches after the GPU has been idle (clock ramp).  Run on the GPU box."""
import sys, time
sys.path.insert(0, ".")
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd._lib import EventPair

N, T, L = 262144, 51, 400
e = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=7)
e.reset()
b = {}
e.rollout(T, actions=None, want=("obs", "rew", "done", "actions"), out=b)
torch.cuda.synchronize()
time.sleep(2.0)  # idle
ev = [EventPair() for _ in range(L)]
for k in range(L):
    e.rollout(T, actions=None, want=("obs", "rew", "done", "actions"), out=b, events=ev[k])
torch.cuda.synchronize()
ms = [x.elapsed_ms() for x in ev]
t = 0.0
for k in (0, 1, 2, 5, 10, 20, 40, 60, 80, 100, 150, 200, 300, 399):
    print(f"launch {k:3d} (t = {sum(ms[:k]):7.2f} ms): {ms[k] * 1e3:7.1f} us")
