"""Soak of the two host-driven loops: 153 000 MR_Env.step calls on the polled host record (3 000 episodes with their resets) and
100 000 iterations of DDPG.train (step kernel with actor + replay sink, update with the policy upload) -- rates, finiteness, device
status.   python tools/soak.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from mr_rl_amd import MR_Env, MRConfig, MRVecEnv
from mr_rl_amd.ddpg import DDPG
env = MR_Env(seed=1); env.reset()
t0 = time.perf_counter(); n = 0
for ep in range(3000):
    env.reset()
    for k in range(51):
        o, r, d, _ = env.step([5.0, 1.0 + 0.01 * k]); n += 1
el = time.perf_counter() - t0
print("facade", n, "steps", round(n / el), "steps/s; last obs", o, flush=True)
env.close()
cfg = MRConfig(noise_var=0.1, reward_mode="goal", auto_reset=True, init_low=(20.0, 20.0), init_high=(30.0, 30.0), min_dist2goal=25.0)
venv = MRVecEnv(1024, cfg=cfg, seed=0, track_actions=True)
ag = DDPG(venv, seed=0, obs_scale=(0.01, 0.01, 0.01, 0.01, 1.0), device_actor=True, fused=True, buffer_size=50000)
t0 = time.perf_counter()
rets = ag.train(100000)
torch.cuda.synchronize(); el = time.perf_counter() - t0
venv.check_status()
print("train 100000 iterations", round(100000 / el), "it/s; returns finite", bool(np.isfinite(rets).all()), "params finite", bool(torch.isfinite(ag.fused.online).all()), len(rets))
