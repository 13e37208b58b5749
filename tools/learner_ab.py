#!/usr/bin/env python3
"""A/B of two libmrsim builds on the fused DDPG update: same ring, same seeds -> parameters compared bitwise, us per update.
python tools/learner_ab.py prev all   (mr_rl_amd/variants/libmrsim_<tag>.so)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib
from mr_rl_amd.ddpg import DDPG
dev = torch.device("cuda", 0)
res = []
for tag in sys.argv[1:]:
    env = MRVecEnv(256, cfg=MRConfig(auto_reset=True), device=dev, seed=1)
    ag = DDPG(env, seed=1, obs_scale=[0.01] * 5, fused=True)
    g = torch.Generator(device=dev).manual_seed(1)
    n = 10000
    s = torch.randn(n, 5, device=dev, generator=g)
    ag.buffer.add(s, torch.randn(n, 2, device=dev, generator=g), torch.randn(n, device=dev, generator=g),
                  (torch.rand(n, device=dev, generator=g) < 0.02).float(), s + 0.01 * torch.randn(n, 5, device=dev, generator=g))
    ag.fused._L = _lib.load(os.path.join(ROOT, "mr_rl_amd", "variants", f"libmrsim_{tag}.so"))
    ag.fused.update(n=40)
    torch.cuda.synchronize()
    snap = ag.fused.online.clone()
    t0 = time.perf_counter()
    for _ in range(20):
        ag.fused.update(n=100)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 2000 * 1e6
    res.append((tag, snap, ag.fused.online.clone(), us))
    print(f"{tag}: {us:.2f} us per update (100 updates per launch)")
for tag, snap, fin, us in res[1:]:
    print(f"{tag} vs {res[0][0]}: after 40 updates bitwise equal: {torch.equal(snap, res[0][1])}; after 2040: {torch.equal(fin, res[0][2])}; "
          f"max |diff| {float((fin - res[0][2]).abs().max()):.3e}")
