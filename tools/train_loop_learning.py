"""Does the reference-shaped loop (DDPG.train: one env step, one replay add, ONE 64-row update per iteration, the policy in the step
kernel refreshed after every update -- RL/MR_ddpg.py:262-311) learn?  The one-step goal task "C" of tools/learning_check.py, N envs
in lockstep, both bookkeeping forms and both learners; prints the mean return per tenth of the run.
   python tools/train_loop_learning.py [--envs 64] [--steps 4000] [--seeds 0 1 2 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run(envs, steps, seed, fused=True, bookkeeping="auto", ou_sigma=5.0):
    from learning_check import task_cfg
    from mr_rl_amd import MRVecEnv
    from mr_rl_amd.ddpg import DDPG
    env = MRVecEnv(envs, cfg=task_cfg("C"), seed=seed, track_actions=True)
    agent = DDPG(env, seed=seed, obs_scale=[0.1] * 5, fused=fused, device_actor=True)
    agent.device_actor.sigma = float(ou_sigma)         # exploration wide enough to reach the goal from the start box
    rets = agent.train(steps, fused_bookkeeping=bookkeeping)
    env.check_status()
    return agent, rets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3])
    a = ap.parse_args()
    for fused in (True, False):
        for seed in a.seeds:
            agent, rets = run(a.envs, a.steps if fused else min(a.steps, 1500), seed, fused=fused)
            k = max(1, len(rets) // 10)
            print("fused" if fused else "eager", "seed", seed, "updates", agent._updates, "mean return per tenth:",
                  [round(sum(rets[i:i + k]) / len(rets[i:i + k]), 1) for i in range(0, len(rets), k)], flush=True)


if __name__ == "__main__":
    main()
