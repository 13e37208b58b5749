"""DDPG consumer of MRVecEnv in PyTorch-ROCm.

The reference's `RL/MR_ddpg.py` is TF1 + tflearn (neither installable here), so this is a PyTorch twin of
its components with the same architecture and hyper-parameters, keeping every tensor on the GPU:

  ReplayBuffer      RL/MR_ddpg.py:16-57    deque of 10 000 transitions -> preallocated device ring
  OUNoise           :59-78                 theta=0.15, sigma=0.3, dt=1e-2, one process per env
  ActorNetwork      :80-160                5 -> 64 -> BN -> ReLU -> 64 -> BN -> ReLU -> 2, tanh * action_bound,
                                           last layer U[-3e-3, 3e-3]
  CriticNetwork     :163-249               s -> 64 -> BN -> ReLU; relu(h W1 + a W2 + b2) with 32 units (t1's bias is
                                           unused in the reference, :218-219); -> 1, last layer U[-3e-3, 3e-3]
  train             :251-323               gamma=0.99, tau=0.001, Adam 1e-3 / 1e-2, batch 64

Parity: UNPINNED (TensorFlow 1.x / tflearn are absent and training is stochastic); the tests check the
pieces' mathematics (OU statistics, soft update, ring semantics, critic target) and that the loop runs on
the device env.  Deliberate differences from the reference loop: N envs step in lockstep (one policy
forward for all), and the warm-up quirk of :283-286,307 (`state = next_state` skipped while the buffer
fills) is NOT reproduced.
"""
import math

import torch
import torch.nn as nn


class ReplayBuffer:
    """Device ring buffer of transitions (s, a, r, done, s2); uniform sampling with replacement-free
    `randperm` when the request fits, like random.sample (RL/MR_ddpg.py:37-44)."""

    def __init__(self, buffer_size, state_dim=5, action_dim=2, device="cuda", seed=123):
        self.buffer_size, self.count, self.head = int(buffer_size), 0, 0
        dev = torch.device(device)
        self.s = torch.zeros((buffer_size, state_dim), device=dev)
        self.a = torch.zeros((buffer_size, action_dim), device=dev)
        self.r = torch.zeros(buffer_size, device=dev)
        self.t = torch.zeros(buffer_size, device=dev)
        self.s2 = torch.zeros((buffer_size, state_dim), device=dev)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)

    def add(self, s, a, r, t, s2):
        """Append a batch [B,...] of transitions; the oldest entries are overwritten (deque.popleft)."""
        B = s.shape[0]
        if B >= self.buffer_size:
            s, a, r, t, s2 = (x[-self.buffer_size:] for x in (s, a, r, t, s2))
            B = self.buffer_size
        idx = (self.head + torch.arange(B, device=self.s.device)) % self.buffer_size
        self.s[idx], self.a[idx], self.r[idx], self.t[idx], self.s2[idx] = s, a, r.float(), t.float(), s2
        self.head = (self.head + B) % self.buffer_size
        self.count = min(self.count + B, self.buffer_size)

    def size(self):
        return self.count

    def sample_batch(self, batch_size):
        n = min(batch_size, self.count)
        idx = torch.randperm(self.count, device=self.s.device, generator=self.gen)[:n]
        return self.s[idx], self.a[idx], self.r[idx], self.t[idx], self.s2[idx]

    def clear(self):
        self.count = self.head = 0


class OUNoise:
    """Ornstein-Uhlenbeck exploration noise, one independent process per env (RL/MR_ddpg.py:59-78)."""

    def __init__(self, shape, sigma=0.3, theta=0.15, dt=1e-2, device="cuda", seed=0):
        self.theta, self.sigma, self.dt = theta, sigma, dt
        self.gen = torch.Generator(device=torch.device(device))
        self.gen.manual_seed(seed)
        self.x_prev = torch.zeros(shape, device=device)

    def __call__(self):
        n = torch.randn(self.x_prev.shape, device=self.x_prev.device, generator=self.gen)
        x = self.x_prev + self.theta * (0.0 - self.x_prev) * self.dt + self.sigma * math.sqrt(self.dt) * n
        self.x_prev = x
        return x

    def reset(self, mask=None):
        if mask is None:
            self.x_prev.zero_()
        else:
            self.x_prev[mask] = 0.0


def _uniform_(layer, lim):
    nn.init.uniform_(layer.weight, -lim, lim)
    if layer.bias is not None:
        nn.init.uniform_(layer.bias, -lim, lim)


class Actor(nn.Module):
    def __init__(self, state_dim=5, action_dim=2, action_bound=(20.0, 2 * math.pi)):
        super().__init__()
        self.fc1, self.bn1 = nn.Linear(state_dim, 64), nn.BatchNorm1d(64)
        self.fc2, self.bn2 = nn.Linear(64, 64), nn.BatchNorm1d(64)
        self.out = nn.Linear(64, action_dim)
        _uniform_(self.out, 3e-3)  # "Final layer weights are init to Uniform[-3e-3, 3e-3]" (:131-132)
        self.register_buffer("action_bound", torch.as_tensor(action_bound, dtype=torch.float32))

    def forward(self, s):
        h = torch.relu(self.bn1(self.fc1(s)))
        h = torch.relu(self.bn2(self.fc2(h)))
        return torch.tanh(self.out(h)) * self.action_bound  # scaled_out (:136-137)


class Critic(nn.Module):
    def __init__(self, state_dim=5, action_dim=2):
        super().__init__()
        self.fc1, self.bn1 = nn.Linear(state_dim, 64), nn.BatchNorm1d(64)
        self.t1 = nn.Linear(64, 32, bias=False)   # t1.b exists in the reference but is never used (:218-219)
        self.t2 = nn.Linear(action_dim, 32)
        self.out = nn.Linear(32, 1)
        _uniform_(self.out, 3e-3)

    def forward(self, s, a):
        h = torch.relu(self.bn1(self.fc1(s)))
        h = torch.relu(self.t1(h) + self.t2(a))
        return self.out(h)


@torch.no_grad()
def soft_update(target, online, tau):
    """target <- tau * online + (1 - tau) * target (RL/MR_ddpg.py:101-104,190-193); BN statistics are copied."""
    for pt, po in zip(target.parameters(), online.parameters()):
        pt.mul_(1.0 - tau).add_(po, alpha=tau)
    for bt, bo in zip(target.buffers(), online.buffers()):
        bt.copy_(bo)


class DDPG:
    """The training loop of RL/MR_ddpg.py:251-323 for N envs in lockstep."""

    def __init__(self, env, gamma=0.99, tau=0.001, actor_lr=1e-3, critic_lr=1e-2, min_batch=64, buffer_size=10000,
                 seed=0, obs_scale=None, device_actor=False, refresh_every=1):
        """device_actor=True: the behaviour policy of train() is evaluated INSIDE the env's step kernel
        (mr_rl_amd.actor.DeviceActor: folded eval-mode network + OU noise in libmrsim.so) instead of as eager PyTorch
        between two launches; its parameters are re-uploaded from the learner's actor every `refresh_every` updates."""
        self.env, self.gamma, self.tau, self.min_batch = env, gamma, tau, min_batch
        dev = env.device
        torch.manual_seed(seed)
        bound = torch.as_tensor(env.action_space.high, dtype=torch.float32)  # RL/MR_ddpg.py:345
        self.actor, self.actor_t = Actor(5, 2, bound).to(dev), Actor(5, 2, bound).to(dev)
        self.critic, self.critic_t = Critic().to(dev), Critic().to(dev)
        self.actor_t.load_state_dict(self.actor.state_dict())
        self.critic_t.load_state_dict(self.critic.state_dict())
        self.opt_a = torch.optim.Adam(self.actor.parameters(), lr=actor_lr)
        self.opt_c = torch.optim.Adam(self.critic.parameters(), lr=critic_lr)
        self.buffer = ReplayBuffer(buffer_size, device=dev)
        self.noise = OUNoise((env.num_envs, 2), device=dev, seed=seed)
        # optional fixed observation scaling (the reference feeds raw observations; obs are O(100))
        self.obs_scale = None if obs_scale is None else torch.as_tensor(obs_scale, dtype=torch.float32, device=dev)
        self.device_actor, self.refresh_every, self._updates = None, max(1, int(refresh_every)), 0
        if device_actor:
            from .actor import DeviceActor
            self.actor.eval()
            self.device_actor = DeviceActor.from_module(self.actor, obs_scale=obs_scale, device=dev, ou=True,
                                                        theta=self.noise.theta, sigma=self.noise.sigma, dt=self.noise.dt,
                                                        reset_on_done=True)   # as train() does with its own OUNoise
            self.actor.train()

    def _prep(self, obs):
        return obs if self.obs_scale is None else obs * self.obs_scale

    @torch.no_grad()
    def act(self, obs, explore=True):
        self.actor.eval()
        a = self.actor(self._prep(obs))
        self.actor.train()
        return a + self.noise() if explore else a

    def update(self):
        if self.buffer.size() < self.min_batch:
            return None
        s, a, r, t, s2 = self.buffer.sample_batch(self.min_batch)
        with torch.no_grad():
            y = r[:, None] + self.gamma * self.critic_t(s2, self.actor_t(s2)) * (1.0 - t[:, None])  # :295-297
        loss_c = torch.mean((y - self.critic(s, a)) ** 2)
        self.opt_c.zero_grad(set_to_none=True); loss_c.backward(); self.opt_c.step()
        loss_a = -self.critic(s, self.actor(s)).mean()  # ascent along dQ/da (:303-305)
        self.opt_a.zero_grad(set_to_none=True); loss_a.backward(); self.opt_a.step()
        soft_update(self.actor_t, self.actor, self.tau)
        soft_update(self.critic_t, self.critic, self.tau)
        self._updates += 1
        if self.device_actor is not None and self._updates % self.refresh_every == 0:
            self.actor.eval()
            self.device_actor.load_module(self.actor)    # fold batch norm, pack, upload (19 KB)
            self.actor.train()
        return float(loss_c.detach()), float(loss_a.detach())

    def train_collected(self, episodes, updates_per_episode=4, sample=4096, streams=2, math="f32"):
        """The loop of RL/MR_ddpg.py:251-323 at collection speed: every episode of all N envs is ONE fused launch group of
        the rollout kernel with this agent's actor (+ OU noise) as its in-kernel policy (RolloutCollector(policy=DeviceActor)),
        `sample` of its N x 51 transitions go into the replay ring, `updates_per_episode` learner updates follow, and the new
        parameters are uploaded before the next episode starts (the upload waits for the launches that still read the old
        block).  Needs an env config with auto_reset=True.  Returns the mean return of every episode."""
        from .actor import DeviceActor
        from .collector import RolloutCollector
        env = self.env
        if not env.cfg.auto_reset:
            raise ValueError("train_collected needs MRConfig(auto_reset=True)")
        scale = None if self.obs_scale is None else self.obs_scale.cpu().numpy()
        self.actor.eval()
        pol = DeviceActor.from_module(self.actor, obs_scale=scale, device=env.device, ou=True, theta=self.noise.theta,
                                      sigma=self.noise.sigma, dt=self.noise.dt, reset_on_done=True, math=math)
        self.actor.train()
        col = RolloutCollector(env.num_envs, cfg=env.cfg, device=env.device, seed=env.seed_value, env_id0=env.env_id0,
                               goal_table=env.goal_table, streams=streams, policy=pol)
        prev_obs = col.reset().clone()                      # the observation the first action of the episode is computed from
        T, N = col.T, col.N
        gen = torch.Generator(device=env.device)
        gen.manual_seed(12345)
        returns = []
        for k in range(episodes):
            col.collect()
            b = col.ready(k)
            obs_T = b["obs"]
            n_s = min(int(sample), T * N)
            ti = torch.randint(0, T, (n_s,), device=env.device, generator=gen)
            ei = torch.randint(0, N, (n_s,), device=env.device, generator=gen)
            s = torch.where((ti == 0)[:, None], prev_obs[ei], obs_T[(ti - 1).clamp(min=0), ei])
            # s2 of a terminal transition is the next episode's reset observation here; its target is r alone (1 - done = 0)
            self.buffer.add(self._prep(s), b["actions"][ti, ei], b["rew"][ti, ei], b["done"][ti, ei].float(),
                            self._prep(obs_T[ti, ei]))
            ended = b["final_len"] > 0
            returns.append(float(b["final_ret"][ended].mean()) if bool(ended.any()) else float("nan"))
            prev_obs = obs_T[T - 1].clone()
            col.release(k)
            for _ in range(updates_per_episode):
                self.update()
            col.join()                                       # no launch reads the parameter block any more
            self.actor.eval()
            pol.load_module(self.actor)
            self.actor.train()
        col.check_status()
        self.collector, self.device_actor = col, pol
        return returns

    def train(self, total_steps, updates_per_step=1, log_every=0):
        """Runs `total_steps` lockstep env steps (N transitions each); returns per-episode returns seen."""
        env = self.env
        obs = env.reset().clone()
        returns = []
        if self.device_actor is not None and env._actions_out is None:
            raise ValueError("DDPG(device_actor=True) needs MRVecEnv(track_actions=True): the replay ring stores the applied actions")
        for k in range(total_steps):
            if self.device_actor is not None:   # policy + exploration noise + MR_Env.step in ONE kernel
                obs2, rew, done, info = env.step(actor=self.device_actor)
                a = env._actions_out.clone()
            else:
                a = self.act(obs).float().contiguous()
                obs2, rew, done, info = env.step(a)
            # with auto_reset the returned obs of a done env is the reset obs; the transition's s2 is final_obs
            s2 = torch.where(done[:, None], info["final_obs"], obs2) if env.cfg.auto_reset else obs2
            self.buffer.add(self._prep(obs), a, rew, done.float(), self._prep(s2))
            if env.cfg.auto_reset and bool(done.any()):
                returns.append(float(info["final_ret"][done].mean()))
                if self.device_actor is None:
                    self.noise.reset(done)      # (the device actor zeroes its OU state in-kernel: reset_on_done)
            obs = obs2.clone()
            for _ in range(updates_per_step):
                self.update()
            if log_every and (k + 1) % log_every == 0 and returns:
                print(f"step {k + 1}: mean return of last finished episodes {returns[-1]:.2f}")
        return returns
