#!/usr/bin/env python3
"""Where a SHORT timed region of the fused rollout spends its time (the driver's `--steps 20`): one 20-step launch group
per region, (a) with the launch argument blocks built inside the region, (b) built ahead (RolloutCollector.prime), on 1
and 2 sub-shard streams.  Prints wall microseconds per region (median of 30) and the phases of (b)."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig
from mr_rl_amd.collector import RolloutCollector

N, K = 262144, 20
dev = torch.device("cuda", 0)
for S in (1, 2):
    cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7)
    col = RolloutCollector(N, cfg=cfg, device=dev, seed=7, streams=S, T=51, carry="f64")
    col.reset()
    for _ in range(400):
        col.collect()
    torch.cuda.synchronize()
    for primed in (False, True):
        walls, t_run, t_sync = [], [], []
        for rep in range(30):
            k = 3 + rep % 17 if not primed else K   # a length the cache has not seen / has seen
            for _ in range(100):
                col.collect()
            if primed:
                col.prime([K])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            col.collect(steps=K if primed else k)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if not primed:
                col._prepared = {q: v for q, v in col._prepared.items() if q[4] == 51}
            walls.append((t2 - t0) * 1e6); t_run.append((t1 - t0) * 1e6); t_sync.append((t2 - t1) * 1e6)
        print(f"streams {S} primed {primed}: region {statistics.median(walls):7.1f} us  (enqueue {statistics.median(t_run):6.1f}"
              f" + wait {statistics.median(t_sync):6.1f}),  min {min(walls):7.1f}", flush=True)
    col.check_status()
