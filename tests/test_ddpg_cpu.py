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


def test_replay_sink_bookkeeping_follows_add():
    """ReplayBuffer.sink() / advance(): the struct a step kernel writes through names the ring's arrays and its CURRENT head, and
    advance(n) moves head / fill exactly as add() of n rows does (host logic only: nothing is launched)."""
    a, b = ReplayBuffer(10, device="cpu"), ReplayBuffer(10, device="cpu")
    for n in (3, 3, 7, 12, 1):
        sk = a.sink(obs_scale=(0.5, 1, 1, 1, 2))
        assert (sk.s, sk.a, sk.r, sk.done, sk.s2) == tuple(t.data_ptr() for t in (a.s, a.a, a.r, a.t, a.s2))
        assert sk.capacity == 10 and sk.head == a.head == b.head and not sk.ended2
        assert list(sk.obs_scale) == [0.5, 1.0, 1.0, 1.0, 2.0]
        a.advance(n)
        z = torch.zeros(n, 5)
        b.add(z, torch.zeros(n, 2), torch.zeros(n), torch.zeros(n), z)
        assert (a.head, a.count) == (b.head, b.count)
    ended = torch.zeros(2)
    assert a.sink(ended=ended).ended2 == ended.data_ptr() and list(a.sink().obs_scale) == [1.0] * 5


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


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the twin follows what the script does (RL/MR_ddpg.py:120-137,207-223,255-257,283-286,307)
# ---------------------------------------------------------------------------------------------------------------------
class _FakeSpace:
    high = np.array([20.0, 2 * math.pi], dtype=np.float32)


class _FakeCfg:
    auto_reset = False


class _FakeEnv:
    """the attributes DDPG reads of an MRVecEnv, on the CPU: obs = [step, env, 0, 0, 0], an episode ends every `ep` steps"""

    def __init__(self, n=1, ep=1000):
        self.num_envs, self.device, self.action_space, self.cfg = n, torch.device("cpu"), _FakeSpace(), _FakeCfg()
        self._actions_out, self.t, self.ep = None, 0, ep

    def _obs(self):
        o = torch.zeros((self.num_envs, 5))
        o[:, 0] = float(self.t)
        o[:, 1] = torch.arange(self.num_envs, dtype=torch.float32)
        return o

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, a):
        self.t += 1
        done = torch.full((self.num_envs,), self.t % self.ep == 0)
        return self._obs(), torch.full((self.num_envs,), 10.0), done, {}


def _fill(agent, n=256, seed=0):
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(n, 5, generator=g)
    agent.buffer.add(s, torch.randn(n, 2, generator=g), torch.randn(n, generator=g), (torch.rand(n, generator=g) < 0.1).float(),
                     s + 0.1 * torch.randn(n, 5, generator=g))


def test_tflearn_initialisation_moments():
    torch.manual_seed(3)
    a, c = Actor(), Critic()
    for lin in (a.fc1, a.fc2, c.fc1, c.t1, c.t2):       # fully_connected: truncated_normal(stddev 0.02) cut at two sigma, zero bias
        w = lin.weight.detach()
        assert w.abs().max() <= 0.04 + 1e-7
        if w.numel() >= 2048:
            assert abs(w.std().item() - 0.02 * 0.8796) < 1.5e-3 and abs(w.mean().item()) < 1.5e-3
        assert lin.bias is None or float(lin.bias.abs().max()) == 0.0
    for bn in (a.bn1, a.bn2, c.bn1):                     # batch_normalization: gamma ~ N(1, 0.002), beta 0, moving stats (0, 1)
        assert abs(bn.weight.mean().item() - 1.0) < 1.5e-3 and 5e-4 < bn.weight.std().item() < 4e-3
        assert float(bn.bias.abs().max()) == 0.0 and float(bn.running_mean.abs().max()) == 0.0
        assert torch.equal(bn.running_var, torch.ones(64)) and bn.eps == 1e-5
    for out in (a.out, c.out):                           # output layers: U[-3e-3, 3e-3] weights, zero bias
        assert out.weight.abs().max() <= 3e-3 and float(out.bias.abs().max()) == 0.0
    assert Actor(init="torch").fc1.bias.abs().max() > 0  # nn.Linear's defaults are still there on request


def test_reference_bn_mode_is_a_fixed_affine_map_during_the_gradient_steps():
    """tflearn.is_training is never switched on in the script: update() must not touch the batch-norm statistics, and the
    networks' outputs for a sample must not depend on the rest of the batch"""
    from mr_rl_amd.ddpg import DDPG
    agent = DDPG(_FakeEnv(4), seed=1)
    _fill(agent)
    stats = [(bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone())
             for bn in (agent.actor.bn1, agent.actor.bn2, agent.critic.bn1)]
    w0 = agent.actor.fc2.weight.detach().clone()
    for _ in range(5):
        out = agent.update()
        assert isinstance(out[0], torch.Tensor) and isinstance(out[1], torch.Tensor) and out[0].dim() == 0   # no float(): no host sync
    for bn, (m, v, k) in zip((agent.actor.bn1, agent.actor.bn2, agent.critic.bn1), stats):
        assert torch.equal(bn.running_mean, m) and torch.equal(bn.running_var, v) and torch.equal(bn.num_batches_tracked, k)
    assert not torch.equal(agent.actor.fc2.weight, w0) and not agent.actor.training and not agent.critic.training
    assert agent.actor.bn1.weight.grad is not None      # gamma / beta are trainable variables
    s = torch.randn(8, 5)
    assert torch.allclose(agent.actor(s)[:1], agent.actor(s[:1]), atol=1e-6)
    # bn_mode="train": batch statistics in update(), running statistics move, targets follow them
    agent2 = DDPG(_FakeEnv(4), seed=1, bn_mode="train")
    _fill(agent2)
    agent2.update()
    assert agent2.actor.bn1.num_batches_tracked.item() > 0
    assert torch.equal(agent2.actor_t.bn1.running_mean, agent2.actor.bn1.running_mean)


def test_target_networks_start_one_soft_update_from_their_own_initialisation():
    from mr_rl_amd.ddpg import DDPG
    ref = DDPG(_FakeEnv(1), seed=5)
    cp = DDPG(_FakeEnv(1), seed=5, target_init="copy")
    assert torch.equal(cp.actor_t.fc2.weight, cp.actor.fc2.weight) and torch.equal(cp.critic_t.t1.weight, cp.critic.t1.weight)
    d = (ref.actor_t.fc2.weight - ref.actor.fc2.weight).abs().mean().item()
    assert d > 5e-3        # two independent truncated normals of stddev 0.02, moved 0.1 % towards each other (:255-257)
    assert torch.equal(ref.actor.fc2.weight, cp.actor.fc2.weight)   # same seed, same online network


def test_sampler_without_replacement_respects_the_fill_count():
    from mr_rl_amd.ddpg import DDPG
    agent = DDPG(_FakeEnv(1), seed=0, buffer_size=500)
    _fill(agent, 100)
    agent._count_t.fill_(100.0)
    for _ in range(20):
        idx = agent._sample(64)
        assert idx.max().item() < 100 and len(set(idx.tolist())) == 64      # random.sample: distinct, only filled slots
    agent.sample_mode = "with_replacement"
    assert agent._sample(64).max().item() < 100


def test_warmup_quirk_observation_sequence():
    """RL/MR_ddpg.py:283-286,307: `state = next_state` is skipped while the ring holds fewer than min_batch transitions"""
    from mr_rl_amd.ddpg import DDPG
    seen = []
    agent = DDPG(_FakeEnv(1), seed=0, min_batch=4)
    agent.train(7, warmup_quirk=True, observe=lambda k, o: seen.append(float(o[0, 0])))
    assert seen == [0.0, 0.0, 0.0, 0.0, 4.0, 5.0, 6.0]
    assert agent.buffer.s[:4, 0].tolist() == [0.0, 0.0, 0.0, 0.0] and agent.buffer.s2[:4, 0].tolist() == [1.0, 2.0, 3.0, 4.0]
    seen2 = []
    DDPG(_FakeEnv(1), seed=0, min_batch=4).train(5, observe=lambda k, o: seen2.append(float(o[0, 0])))
    assert seen2 == [0.0, 1.0, 2.0, 3.0, 4.0]           # default: the lockstep loop advances every step


def test_soft_update_leaves_statistics_alone_in_reference_mode():
    a, b = Actor(), Actor()
    with torch.no_grad():
        a.bn1.running_mean.fill_(3.0)
    soft_update(b, a, 0.5, copy_buffers=False)
    assert float(b.bn1.running_mean.abs().max()) == 0.0
    soft_update(b, a, 0.5)
    assert torch.equal(b.bn1.running_mean, a.bn1.running_mean)
