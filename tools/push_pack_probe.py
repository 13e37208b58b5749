import sys, time
sys.path.insert(0, ".")
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd.ddpg import DDPG
from mr_rl_amd.actor import DeviceActor
from mr_rl_amd.collector import RolloutCollector
dev = torch.device("cuda", 0)
cfg = MRConfig(noise_var=1.0, auto_reset=True, noise_law="collapsed")
env = MRVecEnv(262144, cfg=cfg, device=dev, seed=7)
ag = DDPG(env, seed=7, obs_scale=[0.01] * 5, fused=True)
pol = DeviceActor.from_module(ag.actor, obs_scale=[0.01] * 5, device=dev, ou=True, reset_on_done=True, math="bf16", slots=2)
col = RolloutCollector(262144, cfg=cfg, device=dev, seed=7, streams=2, policy=pol)
prev = col.reset().clone()
col.collect(); b = col.ready()
torch.cuda.synchronize()
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
k = [0]
def push():
    ag.buffer.push_from_rollout(b, prev, 4096, [0.01] * 5, 12345, k[0]); k[0] += 1
print("replay push of 4096 transitions: %.1f us per call (back to back)" % timeit(push))
print("policy upload (fold + pack): %.1f us per call" % timeit(lambda: ag.sync_policy(pol, slot=0)))
