#!/usr/bin/env python3
"""The DDPG collection loop (RL/MR_ddpg.py:270-311: action = actor.predict(state) + actor_noise(); env.step(action)) on one
GPU with an OBSERVATION-DEPENDENT policy: how many env-steps/s does each form deliver?

  eager      mr_rl_amd.ddpg.DDPG.act (PyTorch actor + OU noise) between two env kernels          (round 2's only form)
  gym        mrsim_actor_forward -> mrsim_step, two launches per step, captured in a hipGraph of one episode
  step       mrsim_step with the actor inside the step kernel, one launch per step, hipGraph of one episode
  fused      RolloutCollector(policy=DeviceActor): the whole episode in one launch per sub-shard stream, every
             transition written to the [T, N, ...] buffers a learner reads

    python examples/ddpg_collect_throughput.py [num_envs] [episodes] [--json]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mr_rl_amd import MRConfig, MRVecEnv  # noqa: E402
from mr_rl_amd.actor import DeviceActor  # noqa: E402
from mr_rl_amd.collector import RolloutCollector  # noqa: E402
from mr_rl_amd.ddpg import DDPG  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 262144
EPISODES = int(args[1]) if len(args) > 1 else 200
T = 51
SCALE = [0.01] * 5
res = {"num_envs": N, "episode_steps": T}


def rate(steps, el):
    return N * steps / el / 1e9


# ---- eager PyTorch actor in the loop (what a user of the round-2 twin got)
env = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=0)
agent = DDPG(env, seed=0, obs_scale=SCALE, buffer_size=1024)
env.reset()


def eager(steps):
    obs = env.obs
    for _ in range(steps):
        obs, rew, done, info = env.step(agent.act(obs).float().contiguous())


eager(102)
torch.cuda.synchronize()
t0 = time.perf_counter()
eager(2 * T)
torch.cuda.synchronize()
res["eager_pytorch_actor_G"] = rate(2 * T, time.perf_counter() - t0)

# ---- device actor, gym-loop forms in a hipGraph of one episode (exact f32, then the bf16x3 arithmetic)
actor = DeviceActor.from_module(agent.actor, obs_scale=SCALE, device=env.device)
actor_bf_g = DeviceActor.from_module(agent.actor, obs_scale=SCALE, device=env.device, math="bf16x3")
for form, pol in (("gym", actor), ("step", actor), ("gym_bf16x3", actor_bf_g), ("step_bf16x3", actor_bf_g)):
    e = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=0)
    e.reset()
    e.enable_device_step_base()
    acts = torch.empty((N, 2), dtype=torch.float32, device=e.device)

    def body():
        e.step_idx = 0
        for _ in range(T):
            if form.startswith("gym"):
                e.step(pol.forward(e, out=acts))
            else:
                e.step(actor=pol)
        e.advance_step_base(T)
        e.step_idx = 0

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    reps = max(10, EPISODES // 4)
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res[f"{form}_graph_G"] = rate(reps * T, el)
    res[f"{form}_us_per_step"] = el / (reps * T) * 1e6
    e.check_status()

# ---- fused: the collector with the actor as its policy
for S in (1, 2):
    col = RolloutCollector(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=0, streams=S, policy=actor)
    col.reset()
    for k in range(30):
        col.collect(); col.ready(); col.release()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(EPISODES):
        col.collect(); col.ready(); col.release()
    col.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res[f"fused_collector_streams{S}_G"] = rate(EPISODES * T, el)
    res[f"fused_collector_streams{S}_ms_per_episode"] = el / EPISODES * 1e3
    col.check_status()

# ---- fused, 64 x 64 layer in bf16 x 3 arithmetic (f32-class accuracy) and in plain bf16 (ordinary bf16 inference)
for math in ("bf16x3", "bf16"):
    actor_bf = DeviceActor.from_module(agent.actor, obs_scale=SCALE, device=env.device, math=math)
    col = RolloutCollector(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=0, streams=2, policy=actor_bf)
    col.reset()
    for k in range(30):
        col.collect(); col.ready(); col.release()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(EPISODES):
        col.collect(); col.ready(); col.release()
    col.join()
    torch.cuda.synchronize()
    res[f"fused_collector_{math}_G"] = rate(EPISODES * T, time.perf_counter() - t0)
    col.check_status()

# flop accounting of the actor: 2 * (5*64 + 64*64 + 64*2) per env-step
flop = 2 * (5 * 64 + 64 * 64 + 64 * 2)
best = max(res["fused_collector_streams1_G"], res["fused_collector_streams2_G"])
res["actor_flop_per_env_step"] = flop
res["fused_actor_TFLOPs"] = best * 1e9 * flop / 1e12
res["speedup_vs_eager"] = best / res["eager_pytorch_actor_G"]
res["speedup_vs_eager_bf16x3"] = res["fused_collector_bf16x3_G"] / res["eager_pytorch_actor_G"]
res["speedup_vs_eager_bf16"] = res["fused_collector_bf16_G"] / res["eager_pytorch_actor_G"]
if "--json" in sys.argv:
    print(json.dumps(res))
else:
    for k, v in res.items():
        print(f"{k:40s} {v:.4g}" if isinstance(v, float) else f"{k:40s} {v}")
