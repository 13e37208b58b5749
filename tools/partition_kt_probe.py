#!/usr/bin/env python3
"""Evidence for the exit-time crash of `rocprofv3 --kernel-trace` in a process that created CU-masked streams (profiles/r04/NOTES.md;
ADVICE r04): one short train_collected(learner_cus=1) run, then teardown in the order DDPG.close() prescribes (collector, events,
graphs and cached blocks first, the streams last) -- or, with --no-close, the round-4 behaviour (streams destroyed by nobody).
Run as the profiled program:  rocprofv3 --kernel-trace -d <dir> -- python3 tools/partition_kt_probe.py [--no-close]"""
import argparse, faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd.ddpg import DDPG

ap = argparse.ArgumentParser()
ap.add_argument("--no-close", action="store_true")
ap.add_argument("--envs", type=int, default=253952)
a = ap.parse_args()
env = MRVecEnv(a.envs, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=1)
agent = DDPG(env, seed=1, obs_scale=[0.01] * 5, fused=True)
rets = agent.train_collected(12, updates_per_episode=4, sample=4096, streams=8, math="bf16", learner_cus=1)
torch.cuda.synchronize()
print("trained", len(rets), "episodes; partition closed:", agent.partition.closed, flush=True)
if not a.no_close:
    agent.close()
    print("closed in order; partition:", agent.partition, flush=True)
print("exiting", flush=True)
