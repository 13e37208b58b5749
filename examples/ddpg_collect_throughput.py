#!/usr/bin/env python3
"""The DDPG consumer (mr_rl_amd/ddpg.py, the PyTorch twin of RL/MR_ddpg.py) driving MRVecEnv on one GPU: how many
env-steps/s does the gym loop deliver when the actions come from the actor network (+ OU noise) instead of the
in-kernel random policy, and with the replay ring fed every step?  No host synchronisation inside the loop.

    python examples/ddpg_collect_throughput.py [num_envs] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mr_rl_amd import MRConfig, MRVecEnv  # noqa: E402
from mr_rl_amd.ddpg import DDPG  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2040
env = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=0)
agent = DDPG(env, seed=0, obs_scale=[0.01] * 5, buffer_size=10 * N)


def collect(steps, feed_buffer):
    obs = env.obs
    for _ in range(steps):
        a = agent.act(obs).float().contiguous()
        prev = agent._prep(obs).clone() if feed_buffer else None
        obs, rew, done, info = env.step(a)
        if feed_buffer:
            s2 = torch.where(done[:, None], info["final_obs"], obs)
            agent.buffer.add(prev, a, rew, done.float(), agent._prep(s2))


env.reset()
for feed in (False, True):
    collect(510, feed)  # settle clocks
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    collect(STEPS, feed)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"N={N}: actor-in-the-loop{' + replay ring' if feed else ''}: {N * STEPS / el / 1e9:.2f} G env-steps/s "
          f"({el / STEPS * 1e6:.1f} us per step)", flush=True)
env.check_status()
# a few learner updates, to show the whole consumer runs on the same tensors
for _ in range(5):
    agent.update()
print("updates ok; replay size", agent.buffer.size())
