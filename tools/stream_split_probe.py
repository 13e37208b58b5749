#!/usr/bin/env python3
"""Probe: does splitting one GPU's envs over S HIP streams (S sub-shards, kernels of different shards overlap) beat one
launch of all envs?  Usage: stream_split_probe.py [episodes] [outputs: full|none] [N].  Run on the GPU box."""
import sys, time
sys.path.insert(0, ".")
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd.dist import shard_of

EP = int(sys.argv[1]) if len(sys.argv) > 1 else 600
want = ("obs", "rew", "done", "actions") if (len(sys.argv) < 3 or sys.argv[2] == "full") else ("rew",)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
T = 51
for S in (1, 2, 4):
    envs, streams, bufs = [], [], []
    for k in range(S):
        id0, n = shard_of(N, k, S)
        e = MRVecEnv(n, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=7, env_id0=id0)
        e.reset()
        envs.append(e); streams.append(torch.cuda.Stream()); bufs.append({})
    torch.cuda.synchronize()

    def run(n_ep):
        for _ in range(n_ep):
            for e, st, b in zip(envs, streams, bufs):
                with torch.cuda.stream(st):
                    e.rollout(T, actions=None, want=want, out=b)

    run(300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(EP)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"N={N} outputs={want} streams={S}: {N * T * EP / el / 1e9:.2f} G env-steps/s  ({el / EP * 1e6:.1f} us per episode of all envs)", flush=True)
