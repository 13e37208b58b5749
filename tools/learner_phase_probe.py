#!/usr/bin/env python3
"""Where the fused DDPG update (mrsim_ddpg_update) spends its time: make -C mr_rl_amd/csrc lprobe, then on the GPU box
python tools/learner_phase_probe.py  -> microseconds per phase of one update (s_memtime, 100 MHz), idle GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib
from mr_rl_amd.ddpg import DDPG
dev = torch.device("cuda", 0)
env = MRVecEnv(256, cfg=MRConfig(auto_reset=True), device=dev, seed=1)
ag = DDPG(env, seed=1, obs_scale=[0.01] * 5, fused=True)
g = torch.Generator(device=dev).manual_seed(1)
n = 10000
s = torch.randn(n, 5, device=dev, generator=g)
ag.buffer.add(s, torch.randn(n, 2, device=dev, generator=g), torch.randn(n, device=dev, generator=g),
              (torch.rand(n, device=dev, generator=g) < 0.02).float(), s + 0.01 * torch.randn(n, 5, device=dev, generator=g))
L = ag.fused
L._L = _lib.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mr_rl_amd", "variants", "libmrsim_lprobe.so"))
L.losses = torch.zeros(32, dtype=torch.float32, device=dev)
names = ["rows drawn", "targets staged", "targets y", "online staged", "critic fwd+bwd", "critic Adam+soft", "critic re-staged",
         "actor fwd+bwd", "actor Adam+soft"]
sub = ["a: layer 1", "a: layer 2 product + bn/relu + out partials", "a: tanh", "a: critic layer 1", "a: critic layer 2 + dQ/da partials",
       "a: q, dz3", "a: out-layer grads, delta 2", "a: small grads + dW2 product", "a: d h1 product"]
for nup in (1, 8):
    acc = torch.zeros(18, dtype=torch.float64)
    reps = 50
    for _ in range(reps):
        L.update(n=nup)
        torch.cuda.synchronize()
        acc += L.losses[2:20].double().cpu()
    t = (acc / reps / 100.0).tolist()     # 100 MHz -> us
    print(f"n_updates per launch = {nup}: last update of the launch, us at the end of each phase (and the phase's own time)")
    prev = 0.0
    for nm, v in zip(names, t[:9]):
        print(f"  {nm:18s} {v:7.2f}  (+{v - prev:6.2f})")
        prev = v
    prev = t[6]
    for nm, v in zip(sub, t[9:]):
        print(f"      {nm:48s} {v:7.2f}  (+{v - prev:6.2f})")
        prev = v
    print(f"      {'a: layer-1 backward':48s} {t[7]:7.2f}  (+{t[7] - prev:6.2f})")
