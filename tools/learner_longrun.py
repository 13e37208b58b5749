#!/usr/bin/env python3
"""Fused learner against the eager PyTorch update over MANY updates on the same batch sequence (rows of one fixed ring, drawn on the
host): parameter distance and losses as the run goes.  python tools/learner_longrun.py [updates] [lib tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib
from mr_rl_amd.ddpg import DDPG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
env = MRVecEnv(256, cfg=MRConfig(auto_reset=True), seed=0)
eager, fused = DDPG(env, seed=3), DDPG(env, seed=3, fused=True)
if len(sys.argv) > 2:
    fused.fused._L = _lib.load(os.path.join(ROOT, "mr_rl_amd", "variants", f"libmrsim_{sys.argv[2]}.so"))
g = torch.Generator().manual_seed(5)
n = 4096
s = torch.randn(n, 5, generator=g)
a = torch.randn(n, 2, generator=g) * 3
# a learnable target: reward depends on state and action
r = -(s[:, 0] - 0.3 * a[:, 0]) ** 2 + 0.5 * s[:, 1]
d = (torch.rand(n, generator=g) < 0.5).float()
s2 = s + 0.3 * torch.randn(n, 5, generator=g)
ring = tuple(x.cuda() for x in (s, a, r, d, s2))


def dist():
    num = den = 0.0
    for pe, pf in zip(list(eager.actor.parameters()) + list(eager.critic.parameters()), list(fused.actor.parameters()) + list(fused.critic.parameters())):
        num += float((pe - pf).pow(2).sum()); den += float(pe.pow(2).sum())
    return (num / den) ** 0.5


for k in range(N):
    idx = torch.randint(0, n, (64,), generator=g).cuda()
    batch = tuple(x[idx] for x in ring)
    le = eager.update(batch)
    lf = fused.update(batch)
    if k in (0, 1, 2, 5, 10, 20, 50, 100, 200, 400, 800, 1600) or k == N - 1:
        print(f"update {k + 1:5d}: critic loss eager {float(le[0]):10.5f} fused {float(lf[0]):10.5f}   actor loss eager {float(le[1]):9.5f} fused {float(lf[1]):9.5f}"
              f"   relative parameter distance {dist():.3e}")
