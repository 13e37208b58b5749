#!/usr/bin/env python3
"""Probe: throughput of the fused rollout against the number of steps per launch (launch gap and end-of-kernel tail
are amortised over more steps).  Run on the GPU box."""
import sys, time
sys.path.insert(0, ".")
import torch
from mr_rl_amd import MRConfig, MRVecEnv

N = 262144
want = ("obs", "rew", "done", "actions")
for T in (51, 102, 255, 510, 1020):
    e = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=7)
    e.reset()
    b = {}
    reps = max(4, 30600 // T)
    for _ in range(max(2, 5100 // T)):
        e.rollout(T, actions=None, want=want, out=b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.rollout(T, actions=None, want=want, out=b)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"T={T}: {N * T * reps / el / 1e9:.2f} G env-steps/s  ({el / reps / T * 1e6:.3f} us per step)", flush=True)
    del e, b
    torch.cuda.empty_cache()
