#!/usr/bin/env python3
"""Where the HOST time of DDPG.train_collected goes (cProfile over the episode loop; the device is idle-waiting whenever the host
is behind): python tools/e2e_host_profile.py [math] [U] [episodes] [streams] [learner_cus] [envs]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd.ddpg import DDPG
math = sys.argv[1] if len(sys.argv) > 1 else "bf16"
U = int(sys.argv[2]) if len(sys.argv) > 2 else 0
episodes = int(sys.argv[3]) if len(sys.argv) > 3 else 200
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 2
kw = {}
if len(sys.argv) > 5:
    kw["learner_cus"] = int(sys.argv[5])
dev = torch.device("cuda", 0)
cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, noise_law="collapsed")
N = int(sys.argv[6]) if len(sys.argv) > 6 else 262144
env = MRVecEnv(N, cfg=cfg, device=dev, seed=7)
agent = DDPG(env, seed=7, obs_scale=[0.01] * 5, fused=True)
agent.train_collected(30, updates_per_episode=U, sample=4096, streams=streams, math=math, **kw)   # warm
torch.cuda.synchronize()
st = {}
pr = cProfile.Profile()
pr.enable()
agent.train_collected(episodes, updates_per_episode=U, sample=4096, streams=streams, math=math, stats=st, warm_episodes=10, **kw)
pr.disable()
print("env-steps/s %.3g  us/episode %.1f" % (st["env_steps_timed"] / st["seconds"], 1e6 * st["seconds"] / st["episodes_timed"]))
ps = pstats.Stats(pr)
ps.sort_stats("tottime").print_stats(28)
