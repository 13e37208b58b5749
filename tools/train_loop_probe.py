"""DDPG.train (the reference-shaped loop: one env step, one replay add, one learner update per iteration -- RL/MR_ddpg.py:262-311)
iterations per second, with the step's bookkeeping as PyTorch statements and as ONE launch (mrsim_replay_add_step).
   python tools/train_loop_probe.py [--envs 256 4096] [--steps 2000]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.ddpg import DDPG
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[1, 256, 4096])
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--forms", nargs="+", default=["pytorch", "add_step", "step"], help="bookkeeping forms to run")
    ap.add_argument("--learners", nargs="+", default=["eager", "fused"])
    a = ap.parse_args()
    cfg = MRConfig(noise_var=0.1, reward_mode="goal", auto_reset=True, init_low=(20.0, 20.0), init_high=(30.0, 30.0), min_dist2goal=25.0)
    rows = []
    for n in a.envs:
        for fused_learner in [x == "fused" for x in a.learners]:
            for fb in [{"pytorch": False, "add_step": "add_step", "step": True}[x] for x in a.forms]:
                env = MRVecEnv(n, cfg=cfg, seed=0, track_actions=True)
                agent = DDPG(env, seed=0, obs_scale=(0.01, 0.01, 0.01, 0.01, 1.0), device_actor=True, fused=fused_learner,
                             buffer_size=max(10000, 4 * n))
                agent.train(200, fused_bookkeeping=fb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                agent.train(a.steps, fused_bookkeeping=fb)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                rows.append({"envs": n, "learner": "one kernel" if fused_learner else "eager PyTorch",
                             "bookkeeping": {False: "PyTorch statements", "add_step": "one launch (mrsim_replay_add_step)", True: "in the step kernel"}[fb], "us_per_iteration": round(dt / a.steps * 1e6, 1),
                             "iterations_per_s": round(a.steps / dt), "env_steps_per_s": round(a.steps * n / dt)})
                print(json.dumps(rows[-1]), flush=True)
                agent.close()


if __name__ == "__main__":
    main()
