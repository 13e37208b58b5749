#!/usr/bin/env python3
"""Host cost of RolloutCollector.collect() (the bench's loop: collect + gather per episode) against the device time per episode:
python tools/collect_host_profile.py [episodes] [consumer: 0 = collect only, 1 = collect + ready + release]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_rl_amd import MRConfig
from mr_rl_amd.collector import RolloutCollector
EP = int(sys.argv[1]) if len(sys.argv) > 1 else 600
consumer = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, noise_law="collapsed")
col = RolloutCollector(262144, cfg=cfg, device=dev, seed=7, streams=2, returns_interval=8)
col.reset()
for _ in range(100):
    col.collect()
    if consumer:
        col.ready(); col.release()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(EP):
    col.collect()
    if consumer:
        col.ready(); col.release()
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"consumer={consumer}: host loop {t_host / EP * 1e6:.1f} us per episode (under cProfile), until the device is done {t_all / EP * 1e6:.1f} us per episode"
      f" = {262144 * 51 * EP / t_all / 1e9:.1f} G env-steps/s")
t0 = time.perf_counter()
for _ in range(EP):
    col.collect()
    if consumer:
        col.ready(); col.release()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"without the profiler: host loop {t_host / EP * 1e6:.1f} us per episode, until the device is done {t_all / EP * 1e6:.1f} us = {262144 * 51 * EP / t_all / 1e9:.1f} G env-steps/s")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
