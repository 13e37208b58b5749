#!/usr/bin/env python3
"""Smoke-level learning check of the DDPG loop on the device env (goal reward, fixed seed): a one-step task a constant action
solves -- start 10..14 units to the +x side of the goal, radius 10, a0 = 50 (a step moves 1.5 f), every episode is one step
(max_timesteps = 0): the return is +100 if the step lands inside the radius, -100 otherwise; the untrained actor (output ~ 0)
does not move.  python tools/learning_check.py [--fused 0|1] [--episodes 300] [--updates 16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd.ddpg import DDPG


# start box of x (y in [-1, 1]).  "A" (round 4): every env starts OUTSIDE the radius, the return is a step function of the action
# alone and DDPG with the script's learning rates learns it on 5-6 of 8 seeds with any learner.  "C" (round 5, the GPU test's): a
# third of the envs start inside, the value of the zero action varies with the state, the critic's action gradient carries
# signal from the first updates on: all 8 seeds learn with both learners at OU sigma 5 (profiles/r05/learning_sweep.txt).
TASK_BOX = {"A": (10.0, 14.0), "B": (6.0, 14.0), "C": (4.0, 16.0), "D": (8.0, 12.0)}


def task_cfg(task="A"):
    x0, x1 = TASK_BOX[task]
    return MRConfig(noise_var=0.1, a0=50.0, reward_mode="goal", auto_reset=True, max_timesteps=0, min_dist2goal=10.0,
                    init_low=(x0, -1.0), init_high=(x1, 1.0))


def run(fused, episodes, updates, envs=4096, ou_sigma=10.0, seed=0, learner_lib=None, host_sampler=False, task="A"):
    env = MRVecEnv(envs, cfg=task_cfg(task), seed=seed)
    agent = DDPG(env, seed=seed, obs_scale=[0.1] * 5, fused=bool(fused))
    if learner_lib and fused:        # A/B of learner builds: mr_rl_amd/variants/libmrsim_<tag>.so
        from mr_rl_amd import _lib
        agent.fused._L = _lib.load(os.path.join(ROOT, "mr_rl_amd", "variants", f"libmrsim_{learner_lib}.so"))
    if host_sampler and fused:       # experiment: the fused update on batches drawn by the host-side sampler of the ring
        def burst(n=1):
            for _ in range(n):
                agent.fused.update(agent.buffer.sample_batch(64), n=1)
            agent._updates += n
        agent.update_graphed = burst
    agent.noise.sigma = ou_sigma
    return agent, agent.train_collected(episodes, updates_per_episode=updates, sample=4096)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--episodes", type=int, default=300)
    ap.add_argument("--updates", type=int, default=16)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--ou-sigma", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--learner-lib", default=None)
    ap.add_argument("--host-sampler", type=int, default=0)
    ap.add_argument("--task", default="A", choices=sorted(TASK_BOX))
    a = ap.parse_args()
    agent, rets = run(a.fused, a.episodes, a.updates, a.envs, a.ou_sigma, a.seed, a.learner_lib, bool(a.host_sampler), a.task)
    k = max(1, len(rets) // 10)
    print("mean return per tenth of the run:", [round(sum(rets[i:i + k]) / k, 1) for i in range(0, len(rets), k)])
    with torch.no_grad():
        s = torch.tensor([[12.0, 0.0, 0.0, 0.0, 12.0]], device="cuda") * 0.1
        print("actor(12, 0) =", agent.actor(s).tolist(), " updates:", agent._updates)
