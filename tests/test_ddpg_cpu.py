"""CPU: the pieces of the PyTorch DDPG consumer (mr_rl_amd/ddpg.py).  Parity with the
reference's TF1/tflearn implementation is unpinned (those libraries are absent); these tests check the
mathematics each piece is documented to have (RL/MR_ddpg.py line references in the module)."""
import math

import numpy as np
import torch

from mr_rl_amd.ddpg import Actor, Critic, OUNoise, ReplayBuffer, soft_update


def test_replay_ring_semantics():
    rb = ReplayBuffer(10, device="cpu")
    for k in range(4):  # 4 batches of 3 -> 12 > 10: the two oldest are overwritten
        s = torch.full((3, 5), float(k)); a = torch.full((3, 2), float(k))
        rb.add(s, a, torch.full((3,), float(k)), torch.zeros(3), s + 0.5)
    assert rb.size() == 10
    kept = sorted(rb.r.tolist())
    assert kept == sorted([0.0] * 1 + [1.0] * 3 + [2.0] * 3 + [3.0] * 3)
    s, a, r, t, s2 = rb.sample_batch(64)   # fewer stored than requested -> all of them, no repeats
    assert len(r) == 10 and torch.equal(s2, s + 0.5)
    s, a, r, t, s2 = rb.sample_batch(4)
    assert len(r) == 4
    rb.clear()
    assert rb.size() == 0


def test_ou_noise_statistics():
    """x' = x + theta(0 - x)dt + sigma sqrt(dt) N: stationary std = sigma sqrt(dt / (2 theta dt - theta^2 dt^2))."""
    ou = OUNoise((20000, 2), device="cpu", seed=1)
    for _ in range(3000):
        x = ou()
    th, sg, dt = 0.15, 0.3, 1e-2
    t = 3000 * dt
    var_t = sg * sg / (2 * th) * (1 - math.exp(-2 * th * t))  # not yet stationary after 30 time units
    assert abs(x.mean().item()) < 0.01
    assert abs(x.std().item() / math.sqrt(var_t) - 1) < 0.03
    ou.reset(torch.tensor([True] + [False] * 19999))
    assert ou.x_prev[0].abs().sum() == 0 and ou.x_prev[1].abs().sum() > 0


def test_network_shapes_and_bounds():
    torch.manual_seed(0)
    actor, critic = Actor(), Critic()
    s = torch.randn(64, 5) * 100
    a = actor(s)
    assert a.shape == (64, 2)
    assert (a[:, 0].abs() <= 20).all() and (a[:, 1].abs() <= 2 * math.pi).all()   # tanh * action_bound
    assert actor.out.weight.abs().max() <= 3e-3 and critic.out.weight.abs().max() <= 3e-3
    q = critic(s, a)
    assert q.shape == (64, 1)
    assert critic.t1.bias is None and critic.t2.bias is not None  # the reference adds only t2.b
    # dQ/da flows to the actor
    (-critic(s, actor(s)).mean()).backward()
    assert actor.fc1.weight.grad is not None and actor.fc1.weight.grad.abs().sum() > 0


def test_soft_update():
    a, b = Actor(), Actor()
    with torch.no_grad():
        for p in a.parameters():
            p.fill_(1.0)
        for p in b.parameters():
            p.fill_(0.0)
    soft_update(b, a, 0.001)
    for p in b.parameters():
        assert torch.allclose(p, torch.full_like(p, 0.001))
