#!/usr/bin/env python3
"""Does a captured hipGraph of fused-rollout launches close the ~6 us gap between back-to-back launches on one stream?
python tools/rollout_graph_probe.py   (one stream, all envs per launch, device step base)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_rl_amd import MRConfig, MRVecEnv
dev = torch.device("cuda", 0)
cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, noise_law="collapsed")
N, T, G = 262144, 51, 16
env = MRVecEnv(N, cfg=cfg, device=dev, seed=7)
env.reset()
env.enable_device_step_base()
bufs = {}
out = env.rollout(T, want=("obs", "rew", "done", "actions"), carry="f64", out=bufs)


def one():
    env.rollout(T, want=("obs", "rew", "done", "actions"), carry="f64", out=bufs)
    env.step_idx = 0
    env.advance_step_base(T)


def timed(f, n):
    for _ in range(300):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t_eager = timed(one, 2000)
print(f"eager, one stream: {t_eager * 1e6:.1f} us per episode = {N * T / t_eager / 1e9:.1f} G env-steps/s")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        one()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(G):
        one()
t_graph = timed(g.replay, 200) / G
print(f"graph of {G} launches (+ {G} step-base advances), one stream: {t_graph * 1e6:.1f} us per episode = {N * T / t_graph / 1e9:.1f} G env-steps/s")
env.check_status()
