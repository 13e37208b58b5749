#!/usr/bin/env python3
"""The reference-shaped loop (RL/MR_ddpg.py:270-311: act, step, store, ONE update per env step) on the device: DDPG.train() with the
actor inside the step kernel and the fused learner.  python tools/train_loop_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd.ddpg import DDPG
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for n in (1, 64, 4096, 262144):
    for fused in (True, False):
        env = MRVecEnv(n, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=3, track_actions=True)
        ag = DDPG(env, seed=3, obs_scale=[0.01] * 5, fused=fused, device_actor=True, min_batch=64)
        k = steps if fused else max(200, steps // 10)
        ag.train(100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rets = ag.train(k)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"N = {n:6d}  {'fused learner' if fused else 'eager learner'}: {k / el:9.0f} loop iterations/s = updates/s, {n * k / el:12.3e} env-steps/s"
              f"   ({el / k * 1e6:.0f} us per iteration; {len(rets)} episode boundaries)")
