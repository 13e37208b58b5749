#!/usr/bin/env python3
"""Probe: does the rollout kernel's duration depend on the duty cycle of the launch stream?  Phase A: 2000 launches
back to back (what bench.py's regions do).  Phase B: the same with the stream drained every 30 launches (what
tools/ab_rollout.py does when it reads its events).  Prints the mean kernel duration per block of 100 launches.
Usage (GPU box): python tools/clock_duty_probe.py [carry]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib

carry = sys.argv[1] if len(sys.argv) > 1 else "f64"
N, T, WANT = 262144, 51, ("obs", "rew", "done", "actions")
env = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True, seed=7), seed=7)
env.reset()
bufs = {}
pool = [_lib.EventPair() for _ in range(2000)]


def phase(name, drain_every, gap_s=0.0):
    for _ in range(300):
        env.rollout(T, want=WANT, out=bufs, carry=carry)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, p in enumerate(pool):
        env.rollout(T, want=WANT, out=bufs, carry=carry, events=p)
        if drain_every and (k + 1) % drain_every == 0:
            torch.cuda.synchronize()
            if gap_s:
                time.sleep(gap_s)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [p.elapsed_ms() for p in pool]
    blocks = [sum(ms[i:i + 100]) / 100 * 1e3 for i in range(0, len(ms), 100)]
    print(f"{name}: wall {wall*1e3:.1f} ms for {len(pool)} launches; kernel us per block of 100: " +
          " ".join(f"{b:.1f}" for b in blocks), flush=True)


phase("A back-to-back          ", 0)
phase("B drained every 30      ", 30)
phase("C drained every 30 + 1ms", 30, 1e-3)
phase("D back-to-back again    ", 0)
