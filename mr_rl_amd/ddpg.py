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
pieces' mathematics (OU statistics, soft update, ring semantics, critic target, initialisation moments, the
batch-norm mode) and that the loop runs on the device env.

What the script does and the twin follows by default (each one switchable):
  * batch norm (`bn_mode="reference"`): the script never calls tflearn.is_training(True), so tflearn's
    batch_normalization runs on its moving statistics -- initialised to mean 0 / variance 1 and never updated -- in
    predict() AND during the gradient steps: a per-feature affine map gamma x / sqrt(1 + eps) + beta with trainable
    gamma, beta.  `bn_mode="train"` = batch statistics during update() (what a PyTorch user would write).
  * initialisation (`init="tflearn"`): fully_connected defaults to truncated_normal(stddev=0.02) weights and zero
    biases (:122,125,207,213-214); batch_normalization to gamma ~ N(1, 0.002), beta = 0; the output layers to
    U[-3e-3, 3e-3] weights (:131-134,222-223) and zero biases.  `init="torch"` keeps nn.Linear's defaults.
  * target networks (`target_init="reference"`): the script builds the targets as independently initialised networks
    and "initialises" them with ONE soft update at tau = 0.001 (:255-257), not a copy.  `"copy"` = hard copy.
  * warm-up (`DDPG.train(warmup_quirk=True)`): `state = next_state` sits after the `continue` of the warm-up branch
    (:283-286,307), so while the ring holds fewer than min_batch transitions the policy keeps seeing the episode's
    reset observation.  With N >= min_batch lockstep envs the ring is full after the first step and the quirk is
    inert; it matters for small N.  Off by default (N envs in lockstep is already a different loop).
Deliberate difference: N envs step in lockstep (one policy forward for all).

The learner is device-resident: update() returns tensors (no host sync), `update_graphed()` replays the whole update
(sample -> critic target -> critic step -> actor step -> two soft updates) as ONE captured hipGraph, and the behaviour
policy's parameter block is folded / packed / uploaded on the device (actor.DeviceActor.load_module_device).
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
        # contiguous slice copies (two segments when the batch wraps): no index tensors, five / ten copy kernels
        first = min(B, self.buffer_size - self.head)
        for dst, src in ((self.s, s), (self.a, a), (self.r, r), (self.t, t), (self.s2, s2)):
            dst[self.head:self.head + first].copy_(src[:first])
            if first < B:
                dst[:B - first].copy_(src[first:])
        self.head = (self.head + B) % self.buffer_size
        self.count = min(self.count + B, self.buffer_size)

    def size(self):
        return self.count

    def push_from_rollout(self, b, prev_obs, n, obs_scale, seed, counter):
        """n transitions sampled (Philox, keyed by (seed, counter)) from one collected launch group `b` (RolloutCollector.ready():
        [T, N, .] obs / actions / rew / done, [N][5] rows) into the ring -- ONE launch of libmrsim's mrsim_replay_push instead
        of ~20 indexing kernels.  prev_obs: [N, 5] observations the group's first actions were computed from."""
        import ctypes as C
        from . import _lib
        obs_T, act_T, rew_T, done_T = b["obs"], b["actions"], b["rew"], b["done"]
        T, N = int(obs_T.shape[0]), int(obs_T.shape[1])
        n = min(int(n), self.buffer_size)
        sc = (C.c_float * 5)(*([1.0] * 5 if obs_scale is None else [float(x) for x in obs_scale]))
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        strm = C.c_void_p(torch.cuda.current_stream(self.s.device).cuda_stream)
        _lib.check(_lib.lib().mrsim_replay_push(N, T, p(obs_T), p(act_T), p(rew_T), p(done_T.view(torch.uint8)), p(prev_obs), sc, n,
                                                p(self.s), p(self.a), p(self.r), p(self.t), p(self.s2), self.buffer_size, self.head,
                                                int(seed), int(counter), strm), "mrsim_replay_push")
        self.head = (self.head + n) % self.buffer_size
        self.count = min(self.count + n, self.buffer_size)

    def add_step(self, obs_prev, actions, rew, done_u8, obs_next, final_obs, final_ret, obs_scale, ended=None):
        """One lockstep env step's bookkeeping as ONE launch of libmrsim's mrsim_replay_add_step: add() of the n transitions
        (s = obs_prev, s2 = final_obs where done else obs_next, both x obs_scale), obs_prev := obs_next in place, and
        ended[2] += {sum of final_ret over the finished envs, their number}.  [n][5] rows; final_obs / final_ret may be None
        (no auto-reset)."""
        import ctypes as C
        from . import _lib
        n = int(obs_prev.shape[0])
        sc = (C.c_float * 5)(*([1.0] * 5 if obs_scale is None else [float(x) for x in obs_scale]))
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        strm = C.c_void_p(torch.cuda.current_stream(self.s.device).cuda_stream)
        _lib.check(_lib.lib().mrsim_replay_add_step(n, p(obs_prev), p(actions), p(rew), p(done_u8), p(obs_next), p(final_obs),
                                                    p(final_ret), sc, p(self.s), p(self.a), p(self.r), p(self.t), p(self.s2),
                                                    self.buffer_size, self.head, p(obs_prev), p(ended), strm),
                   "mrsim_replay_add_step")
        B = min(n, self.buffer_size)
        self.head = (self.head + B) % self.buffer_size
        self.count = min(self.count + B, self.buffer_size)

    def sink(self, obs_scale=None, ended=None):
        """The ring as a _lib.MrsimReplaySink for MRVecEnv.step(actor=..., replay=...): the step kernel stores its transitions
        at the CURRENT head; call advance(n) after the launch.  ended: optional [2] device tensor (+= finished episodes' return
        sum, count)."""
        from . import _lib
        sk = getattr(self, "_sink", None)
        if sk is None:
            sk = self._sink = _lib.MrsimReplaySink(self.s.data_ptr(), self.a.data_ptr(), self.r.data_ptr(), self.t.data_ptr(),
                                                   self.s2.data_ptr(), None, self.buffer_size, 0)
        sc = [1.0] * 5 if obs_scale is None else [float(x) for x in obs_scale]
        for j in range(5):
            sk.obs_scale[j] = sc[j]
        sk.head = self.head
        sk.ended2 = None if ended is None else ended.data_ptr()
        return sk

    def advance(self, n):
        """n transitions were appended at the head by a kernel (sink())."""
        B = min(int(n), self.buffer_size)
        self.head = (self.head + B) % self.buffer_size
        self.count = min(self.count + B, self.buffer_size)

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
            self.x_prev = torch.where(mask.reshape(-1, *([1] * (self.x_prev.dim() - 1))), torch.zeros_like(self.x_prev), self.x_prev)
            # (no boolean-mask assignment: that form reads the mask back to size its index -- a host sync per step)


def _uniform_(layer, lim, bias_too=True):
    nn.init.uniform_(layer.weight, -lim, lim)
    if layer.bias is not None:
        if bias_too:
            nn.init.uniform_(layer.bias, -lim, lim)
        else:
            nn.init.zeros_(layer.bias)


def _tflearn_fc_(layer):
    """tflearn.fully_connected defaults: weights_init='truncated_normal' (stddev 0.02, cut at two sigma), bias_init='zeros'"""
    nn.init.trunc_normal_(layer.weight, mean=0.0, std=0.02, a=-0.04, b=0.04)
    if layer.bias is not None:
        nn.init.zeros_(layer.bias)


def _tflearn_bn_(bn):
    """tflearn batch_normalization defaults: gamma ~ N(1, 0.002), beta = 0, moving mean 0 / variance 1, epsilon 1e-5"""
    nn.init.normal_(bn.weight, mean=1.0, std=0.002)
    nn.init.zeros_(bn.bias)
    bn.running_mean.zero_(); bn.running_var.fill_(1.0)
    bn.eps = 1e-5


class Actor(nn.Module):
    def __init__(self, state_dim=5, action_dim=2, action_bound=(20.0, 2 * math.pi), init="tflearn"):
        super().__init__()
        self.fc1, self.bn1 = nn.Linear(state_dim, 64), nn.BatchNorm1d(64)
        self.fc2, self.bn2 = nn.Linear(64, 64), nn.BatchNorm1d(64)
        self.out = nn.Linear(64, action_dim)
        if init == "tflearn":
            with torch.no_grad():
                _tflearn_fc_(self.fc1); _tflearn_fc_(self.fc2); _tflearn_bn_(self.bn1); _tflearn_bn_(self.bn2)
            _uniform_(self.out, 3e-3, bias_too=False)  # "Final layer weights are init to Uniform[-3e-3, 3e-3]" (:131-134)
        elif init == "torch":
            _uniform_(self.out, 3e-3)
        else:
            raise ValueError("init must be 'tflearn' or 'torch'")
        self.register_buffer("action_bound", torch.as_tensor(action_bound, dtype=torch.float32))

    def forward(self, s):
        h = torch.relu(self.bn1(self.fc1(s)))
        h = torch.relu(self.bn2(self.fc2(h)))
        return torch.tanh(self.out(h)) * self.action_bound  # scaled_out (:136-137)


class Critic(nn.Module):
    def __init__(self, state_dim=5, action_dim=2, init="tflearn"):
        super().__init__()
        self.fc1, self.bn1 = nn.Linear(state_dim, 64), nn.BatchNorm1d(64)
        self.t1 = nn.Linear(64, 32, bias=False)   # t1.b exists in the reference but is never used (:218-219)
        self.t2 = nn.Linear(action_dim, 32)
        self.out = nn.Linear(32, 1)
        if init == "tflearn":
            with torch.no_grad():
                _tflearn_fc_(self.fc1); _tflearn_fc_(self.t1); _tflearn_fc_(self.t2); _tflearn_bn_(self.bn1)
            _uniform_(self.out, 3e-3, bias_too=False)   # :222-223
        elif init == "torch":
            _uniform_(self.out, 3e-3)
        else:
            raise ValueError("init must be 'tflearn' or 'torch'")

    def forward(self, s, a):
        h = torch.relu(self.bn1(self.fc1(s)))
        h = torch.relu(self.t1(h) + self.t2(a))
        return self.out(h)


@torch.no_grad()
def soft_update(target, online, tau, copy_buffers=True):
    """target <- tau * online + (1 - tau) * target over the trainable variables (RL/MR_ddpg.py:101-104,190-193: weights,
    biases, batch-norm gamma / beta).  copy_buffers: also copy the batch-norm statistics (they are not trainable variables
    in the reference and never change there; with bn_mode="train" the target must follow the online network's)."""
    tp, op = list(target.parameters()), list(online.parameters())
    torch._foreach_mul_(tp, 1.0 - tau)
    torch._foreach_add_(tp, op, alpha=tau)
    if copy_buffers:
        for bt, bo in zip(target.buffers(), online.buffers()):
            bt.copy_(bo)


def _mean_returns(ret_rows, len_rows):
    """[E] mean return of the episodes that ended in each launch group, from the [E, N] final_ret / final_len rows"""
    ended = (len_rows > 0).float()
    return (ret_rows * ended).sum(1) / ended.sum(1).clamp(min=1.0)


class DDPG:
    """The training loop of RL/MR_ddpg.py:251-323 for N envs in lockstep."""

    def __init__(self, env, gamma=0.99, tau=0.001, actor_lr=1e-3, critic_lr=1e-2, min_batch=64, buffer_size=10000,
                 seed=0, obs_scale=None, device_actor=False, refresh_every=1, bn_mode="reference", init="tflearn",
                 target_init="reference", sample="auto", fused=False):
        """device_actor=True: the behaviour policy of train() is evaluated INSIDE the env's step kernel
        (mr_rl_amd.actor.DeviceActor: folded eval-mode network + OU noise in libmrsim.so) instead of as eager PyTorch
        between two launches; its parameters are re-uploaded from the learner's actor every `refresh_every` updates
        (sync_policy() does it on demand).
        bn_mode / init / target_init: see the module docstring ("reference" = what the script does).
        sample: "without_replacement" = random.sample's law (RL/MR_ddpg.py:37-44) drawn on the device with fixed shapes
        (uniform keys + top-k, graph-capturable), "with_replacement" = randint (rings too large for a top-k per update),
        "auto" = the former up to 65 536 slots.
        fused=True: update() is ONE hand-written kernel of libmrsim.so (mrsim_ddpg_update: both networks' forward and
        backward passes, both Adam steps and both soft updates in a single launch; needs bn_mode="reference")."""
        self.env, self.gamma, self.tau, self.min_batch = env, gamma, tau, min_batch
        if bn_mode not in ("reference", "train"):
            raise ValueError("bn_mode must be 'reference' or 'train'")
        if target_init not in ("reference", "copy"):
            raise ValueError("target_init must be 'reference' or 'copy'")
        self.bn_mode = bn_mode
        dev = env.device
        torch.manual_seed(seed)
        bound = torch.as_tensor(env.action_space.high, dtype=torch.float32)  # RL/MR_ddpg.py:345
        self.actor, self.actor_t = Actor(5, 2, bound, init=init).to(dev), Actor(5, 2, bound, init=init).to(dev)
        self.critic, self.critic_t = Critic(init=init).to(dev), Critic(init=init).to(dev)
        if target_init == "copy":
            self.actor_t.load_state_dict(self.actor.state_dict())
            self.critic_t.load_state_dict(self.critic.state_dict())
        else:   # "Initialize target network weights": one soft update of independently initialised targets (:255-257)
            soft_update(self.actor_t, self.actor, tau, copy_buffers=False)
            soft_update(self.critic_t, self.critic, tau, copy_buffers=False)
        self._set_mode(training=False)
        capt = dev.type == "cuda"
        self.opt_a = torch.optim.Adam(self.actor.parameters(), lr=actor_lr, capturable=capt, foreach=True)
        self.opt_c = torch.optim.Adam(self.critic.parameters(), lr=critic_lr, capturable=capt, foreach=True)
        self.actor_lr, self.critic_lr = actor_lr, critic_lr
        self.buffer = ReplayBuffer(buffer_size, device=dev)
        if sample == "auto":
            sample = "without_replacement" if buffer_size <= 65536 else "with_replacement"
        if sample not in ("without_replacement", "with_replacement"):
            raise ValueError("sample must be 'auto', 'without_replacement' or 'with_replacement'")
        self.sample_mode = sample
        self.noise = OUNoise((env.num_envs, 2), device=dev, seed=seed)
        # optional fixed observation scaling (the reference feeds raw observations; obs are O(100))
        self.obs_scale = None if obs_scale is None else torch.as_tensor(obs_scale, dtype=torch.float32, device=dev)
        self.device_actor, self.refresh_every, self._updates = None, max(1, int(refresh_every)), 0
        self._graph = None
        self._count_t = torch.zeros((), dtype=torch.float32, device=dev)    # ring fill, as the captured sampler reads it
        self.last_losses = None
        self.fused = None
        self.fused_upload = True    # update(): the policy upload rides in the update's launch (False: a launch of its own; tests)
        if fused:
            if bn_mode != "reference":
                raise ValueError("fused=True needs bn_mode='reference'")
            from .learner import FusedLearner
            self.fused = FusedLearner(self)
        if device_actor:
            from .actor import DeviceActor
            self.actor.eval()
            self.device_actor = DeviceActor.from_module(self.actor, obs_scale=obs_scale, device=dev, ou=True,
                                                        theta=self.noise.theta, sigma=self.noise.sigma, dt=self.noise.dt,
                                                        reset_on_done=True)   # as train() does with its own OUNoise
            self._set_mode(training=False)

    # ------------------------------------------------------------------------------------------------ batch-norm mode
    def _set_mode(self, training):
        """bn_mode="reference": every network stays in inference mode for good (tflearn's training mode is never switched
        on in the script); bn_mode="train": the online networks use batch statistics inside update()."""
        on = bool(training) and self.bn_mode == "train"
        self.actor.train(on); self.critic.train(on)
        self.actor_t.eval(); self.critic_t.eval()

    def _prep(self, obs):
        return obs if self.obs_scale is None else obs * self.obs_scale

    @torch.no_grad()
    def act(self, obs, explore=True):
        self._set_mode(training=False)
        a = self.actor(self._prep(obs))
        return a + self.noise() if explore else a

    # ------------------------------------------------------------------------------------------------ the update
    def _sample(self, n):
        """n ring indices on the device with fixed shapes (graph-capturable); reads the fill count from _count_t"""
        buf = self.buffer
        if self.sample_mode == "without_replacement":
            keys = torch.rand(buf.buffer_size, device=buf.s.device)
            keys = keys + (torch.arange(buf.buffer_size, device=buf.s.device) >= self._count_t).float() * 2.0  # empty slots last
            return torch.topk(keys, n, largest=False, sorted=False).indices
        return (torch.rand(n, device=buf.s.device) * self._count_t).long().clamp_(max=buf.buffer_size - 1)

    def _update_body(self, batch=None):
        """sample -> critic target -> critic step -> actor step -> two soft updates (RL/MR_ddpg.py:288-305); no host sync"""
        buf = self.buffer
        if batch is None:
            idx = self._sample(self.min_batch)
            s, a, r, t, s2 = buf.s[idx], buf.a[idx], buf.r[idx], buf.t[idx], buf.s2[idx]
        else:
            s, a, r, t, s2 = batch
        with torch.no_grad():
            y = r[:, None] + self.gamma * self.critic_t(s2, self.actor_t(s2)) * (1.0 - t[:, None])  # :290-294
        self._set_mode(training=True)
        loss_c = torch.mean((y - self.critic(s, a)) ** 2)                                            # :297
        self.opt_c.zero_grad(set_to_none=True); loss_c.backward(); self.opt_c.step()
        loss_a = -self.critic(s, self.actor(s)).mean()  # ascent along dQ/da (:300-302)
        self.opt_a.zero_grad(set_to_none=True); loss_a.backward(); self.opt_a.step()
        self._set_mode(training=False)
        cb = self.bn_mode == "train"
        soft_update(self.actor_t, self.actor, self.tau, copy_buffers=cb)                              # :305-306
        soft_update(self.critic_t, self.critic, self.tau, copy_buffers=cb)
        return loss_c.detach(), loss_a.detach()

    def update(self, batch=None):
        """One learner update.  Returns (critic loss, actor loss) as device TENSORS -- nothing is synchronised -- or None
        while the ring holds fewer than min_batch transitions (RL/MR_ddpg.py:283-286).  batch: optional explicit
        (s, a, r, done, s2) instead of a ring sample (tests)."""
        if batch is None and self.buffer.size() < self.min_batch:
            return None
        uploaded = False
        if self.fused is not None:
            # one launch: the rows are drawn in the kernel, and when the behaviour policy is due for its refresh the same launch
            # folds and packs the new actor into the policy's block (no mrsim_actor_pack_device launch: 9 us of a 62 us iteration)
            pol = self.device_actor
            uploaded = (self.fused_upload and pol is not None and len(pol.blobs) == 1
                        and (self._updates + 1) % self.refresh_every == 0)
            self.last_losses = self.fused.update(batch, pack_into=pol if uploaded else None)
        else:
            self._count_t.fill_(float(self.buffer.size()))
            self.last_losses = self._update_body(batch)
        self._after_update(uploaded)
        return self.last_losses

    def _after_update(self, uploaded=False):
        self._updates += 1
        if self.device_actor is not None and self._updates % self.refresh_every == 0 and not uploaded:
            self.sync_policy()

    def sync_policy(self, policy=None, slot=None):
        """Upload the learner's current actor into the behaviour policy's parameter block(s) (fold batch norm, pack, copy:
        all on the device, ordered on the current stream -- no host synchronisation)."""
        pol = self.device_actor if policy is None else policy
        if pol is not None:
            if self.fused is not None:
                pol.load_from_learner(self.fused, slot=slot)     # one launch of the library's device-side fold + pack
            else:
                pol.load_module_device(self.actor, slot=slot)

    def capture_update(self):
        """Capture one whole update as a hipGraph (torch.cuda.CUDAGraph is the capture plumbing; Adam runs with
        capturable=True, the sampler reads the ring's fill count from a device word).  update_graphed() replays it."""
        if self.fused is not None:
            return None     # the fused learner is one launch already
        if self._graph is not None:
            return self._graph
        if self.buffer.size() < self.min_batch:
            raise RuntimeError("capture_update: the ring holds fewer than min_batch transitions")
        dev = self.buffer.s.device
        self._count_t.fill_(float(self.buffer.size()))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up on a side stream, as torch's graph recipe asks (allocator, Adam state)
            for _ in range(3):
                self._update_body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._graph_losses = self._update_body()
        return self._graph

    def update_graphed(self, n=1):
        """n learner updates, each ONE graph replay (fused learner: all n in ONE launch).  Returns the loss tensors of the last
        one (device, no sync).  Does not refresh a device-side behaviour policy: call sync_policy() when the envs should see the
        new parameters."""
        if self.fused is not None:                # n updates in ONE launch: the learner holds one compute unit for the burst
            if self.buffer.size() < self.min_batch:
                return None
            self.last_losses = self.fused.update(None, n=n)
            self._updates += n
            return self.last_losses
        if self.buffer.size() < self.min_batch:
            return None                       # RL/MR_ddpg.py:283-286: no update while the ring fills
        if self._graph is None:
            self.capture_update()
        self._count_t.fill_(float(self.buffer.size()))
        for _ in range(n):
            self._graph.replay()
        self._updates += n
        self.last_losses = self._graph_losses
        return self.last_losses

    def train_collected(self, episodes, updates_per_episode=4, sample=4096, streams=2, math="f32", graphed=True,
                        on_episode=None, stats=None, warm_episodes=0, learner_cus=0):
        """The loop of RL/MR_ddpg.py:251-323 at collection speed: every episode of all N envs is ONE fused launch group of
        the rollout kernel with this agent's actor (+ OU noise) as its in-kernel policy (RolloutCollector(policy=DeviceActor)),
        `sample` of its N x 51 transitions go into the replay ring, `updates_per_episode` learner updates follow (graph
        replays / fused launches: no host sync anywhere in the loop), and the new parameters are folded, packed and uploaded
        ON THE DEVICE before the next episode starts.  The learner's updates of episode k run while the envs collect
        episode k + 1 with the parameters of the end of episode k - 1 (the collector double-buffers; one episode of policy lag,
        as any actor/learner split has); graphed=False runs the eager update instead.
        The reference does one update per transition (RL/MR_ddpg.py:288-305); with N lockstep envs that ratio is
        updates_per_episode / (N x 51) -- say which one you ran.  Needs an env config with auto_reset=True.
        Returns the mean return of every episode (device tensors gathered once at the end: one host sync).
        stats: optional dict that receives the wall-clock seconds of episodes [warm_episodes, episodes) (the device is
        synchronised at both ends of that span -- a measurement aid, two host syncs).
        learner_cus: 0 = everything shares the device (the learner's launches find a compute unit only at a launch boundary and
        displace collection blocks: they do NOT hide behind the collection); k >= 1 = the device is partitioned
        (mr_rl_amd.partition.CuPartition): k compute units of every XCC belong to the learner's stream (updates, replay push,
        parameter upload, the bookkeeping copies), all others to `streams` collection streams (8 or 16 sub-shards then: the
        collection no longer fits one resident round and the blocks that do not fit must be able to run beside the other
        sub-shards' next launch group).  The learner then costs the collection its units (3 % for k = 1), not its time."""
        env = self.env
        if not learner_cus:
            return self._train_collected(episodes, updates_per_episode, sample, streams, math, graphed, on_episode, stats,
                                         warm_episodes, None)
        from .partition import CuPartition
        key = (int(learner_cus), int(streams))
        part = getattr(self, "partition", None)
        if getattr(self, "_partition_key", None) != key or part is None or part.closed:
            self.close()                              # the previous partition, with everything that refers to its streams
            self.partition, self._partition_key = CuPartition(env.device, per_xcc=int(learner_cus), collection_streams=int(streams)), key
        part = self.partition
        outer = torch.cuda.current_stream(env.device)
        part.learner_stream.wait_stream(outer)
        with torch.cuda.stream(part.learner_stream):
            out = self._train_collected(episodes, updates_per_episode, sample, streams, math, graphed, on_episode, stats,
                                        warm_episodes, part.collection_streams)
        outer.wait_stream(part.learner_stream)
        return out

    def close(self):
        """Release what train_collected(learner_cus=k) keeps beyond its return: the collector (its ExternalStream wrappers and the
        events recorded on them), the in-kernel policy block, the update graph captured on the learner's stream, the allocator's
        cached blocks -- and only then the CU-masked streams themselves (CuPartition.close()).  Destroying a stream that a live
        wrapper, event or cached block still names is a use-after-destroy at teardown.  Safe to call repeatedly; the agent can
        train again afterwards (a new partition is built on demand)."""
        part = getattr(self, "partition", None)
        if part is None:
            return
        col = getattr(self, "collector", None)
        if col is not None:
            col.join()
        torch.cuda.synchronize(self.env.device)
        self.collector = None                         # built on the partition's streams
        del col
        self._graph = None                            # captured while the learner's stream was current
        self.partition, self._partition_key = None, None
        part.close()

    def _train_collected(self, episodes, updates_per_episode, sample, streams, math, graphed, on_episode, stats, warm_episodes,
                         stream_list):
        from .actor import DeviceActor
        from .collector import RolloutCollector
        env = self.env
        if not env.cfg.auto_reset:
            raise ValueError("train_collected needs MRConfig(auto_reset=True)")
        scale = None if self.obs_scale is None else self.obs_scale.cpu().numpy()
        self._set_mode(training=False)
        pol = DeviceActor.from_module(self.actor, obs_scale=scale, device=env.device, ou=True, theta=self.noise.theta,
                                      sigma=self.noise.sigma, dt=self.noise.dt, reset_on_done=True, math=math, slots=2)
        RB = 32                                   # episodes per block of returns (reduced block by block: bounded memory)
        # Three rotating transition sets: the set of episode k - 1 stays untouched until episode k's replay push has read the
        # observations its first actions were computed from (the last row of episode k - 1) -- no copy of that row.  Episode
        # returns / lengths land in [RB, N] blocks written by the launches themselves (returns_interval).
        col = RolloutCollector(env.num_envs, cfg=env.cfg, device=env.device, seed=env.seed_value, env_id0=env.env_id0,
                               goal_table=env.goal_table, streams=streams, policy=pol, stream_list=stream_list, depth=3,
                               returns_interval=RB)
        prev_obs = col.reset().clone()                      # the observation the first action of the episode is computed from
        T, N = col.T, col.N
        gen = torch.Generator(device=env.device)
        gen.manual_seed(12345)
        means = []
        cur = torch.cuda.current_stream(env.device)
        native_push = env.device.type == "cuda" and not col.env._soa      # mrsim_replay_push reads [N][5] observation rows
        # Two episodes in flight, two parameter blocks: episode k reads block k % 2.  When episode k is ready its block is free:
        # the learner's parameters as of then are uploaded into it and episode k + 2 starts behind that upload; the learner's
        # burst on episode k's transitions runs beside episodes k + 1 / k + 2 (policy lag: up to three episodes, as in any
        # actor / learner split).  Nothing on this path waits on the host.
        col.collect()
        if episodes > 1:
            col.collect()
        import time
        t_start = None
        for k in range(episodes):
            if stats is not None and k == warm_episodes:
                torch.cuda.synchronize(env.device)
                t_start = time.perf_counter()
            b = col.ready(k)                                 # the current stream waits for episode k
            if k > 0 and k + 2 < episodes:
                # Block k % 2 was last read by episode k, which has finished: upload the parameters the learner has produced so far
                # (its launches of the previous iterations precede this one on the current stream)
                self.sync_policy(pol, slot=k % 2)
            obs_T = b["obs"]
            n_s = min(int(sample), T * N)
            if native_push:
                self.buffer.push_from_rollout(b, prev_obs, n_s, scale, 12345, k)
            else:
                ti = torch.randint(0, T, (n_s,), device=env.device, generator=gen)
                ei = torch.randint(0, N, (n_s,), device=env.device, generator=gen)
                s = torch.where((ti == 0)[:, None], prev_obs[ei], obs_T[(ti - 1).clamp(min=0), ei])
                # s2 of a terminal transition is the next episode's reset observation here; its target is r alone (1 - done = 0)
                self.buffer.add(self._prep(s), b["actions"][ti, ei], b["rew"][ti, ei], b["done"][ti, ei].float(),
                                self._prep(obs_T[ti, ei]))
            prev_obs = obs_T[T - 1]                          # a view: its set is released one episode late (below)
            if k > 0:
                col.release(k - 1)
            if k + 2 < episodes:
                col.collect(after=cur)                       # episode k + 2 starts behind the upload and the release alone;
            blk, row = (k // RB) % 2, k % RB                 # the learner burst below runs beside the collection
            if row == RB - 1 or k == episodes - 1:
                means.append(_mean_returns(col.ret_blocks[blk][:row + 1], col.len_blocks[blk][:row + 1]))
                col.free_returns_block(blk)
            if updates_per_episode > 0:
                if graphed and env.device.type == "cuda":
                    self.update_graphed(updates_per_episode)
                else:
                    for _ in range(updates_per_episode):
                        self.update()
            if on_episode is not None:
                on_episode(k, _mean_returns(b["final_ret"][None], b["final_len"][None])[0])
        col.join()
        if stats is not None and t_start is not None:
            torch.cuda.synchronize(env.device)
            stats.update({"seconds": time.perf_counter() - t_start, "episodes_timed": episodes - warm_episodes,
                          "env_steps_timed": (episodes - warm_episodes) * T * N, "updates_timed": (episodes - warm_episodes) * updates_per_episode})
        col.check_status()
        self.sync_policy(pol)
        self.collector, self.device_actor = col, pol
        return [float(x) for x in torch.cat(means).cpu()] if means else []

    def train(self, total_steps, updates_per_step=1, log_every=0, warmup_quirk=False, observe=None, fused_bookkeeping="auto"):
        """Runs `total_steps` lockstep env steps (N transitions each); returns per-episode returns seen.
        warmup_quirk=True reproduces RL/MR_ddpg.py:283-286,307: while the ring holds fewer than min_batch transitions
        after a step, `state = next_state` is skipped -- the policy keeps seeing (and the ring keeps storing as `state`)
        the observation the episode was reset to.  observe: optional callback(step, obs_fed_to_the_policy) (tests).
        fused_bookkeeping: with the in-kernel actor on a HIP device, the step's replay add / state hand-over / finished-episode
        sums need no PyTorch kernels (~20 per step before): "auto" / True = the step kernel writes the transitions into the ring
        itself (MrsimStepIO.replay), "add_step" = one launch of mrsim_replay_add_step after the step, False = the PyTorch
        statements (the three are compared bit for bit in tests/test_gpu_round5.py); True demands the conditions."""
        env = self.env
        can_fuse = (self.device_actor is not None and env.device.type == "cuda" and env.cfg.auto_reset and not warmup_quirk
                    and env._actions_out is not None and not getattr(env, "_soa", False))
        if fused_bookkeeping is True and not can_fuse:
            raise ValueError("fused_bookkeeping=True needs DDPG(device_actor=True), auto_reset, [N][5] observation rows, "
                             "MRVecEnv(track_actions=True), a HIP device and warmup_quirk=False")
        if fused_bookkeeping and can_fuse:
            return self._train_fused_bookkeeping(total_steps, updates_per_step, log_every, observe,
                                                 in_step_kernel=fused_bookkeeping != "add_step")
        obs = env.reset().clone()
        # per-step (sum of the returns of the episodes that ended, their number) stay on the device: the loop never waits for the
        # host (the reference reads every reward on the host, RL/MR_ddpg.py:278-311); read back once per log line and at the end
        ended = []
        if self.device_actor is not None and env._actions_out is None:
            raise ValueError("DDPG(device_actor=True) needs MRVecEnv(track_actions=True): the replay ring stores the applied actions")
        for k in range(total_steps):
            if observe is not None:
                observe(k, obs)
            if self.device_actor is not None:   # policy + exploration noise + MR_Env.step in ONE kernel
                if warmup_quirk:
                    raise ValueError("warmup_quirk needs the eager policy: the in-kernel actor reads the env's own observation")
                obs2, rew, done, info = env.step(actor=self.device_actor)
                a = env._actions_out.clone()
            else:
                a = self.act(obs).float().contiguous()
                obs2, rew, done, info = env.step(a)
            # with auto_reset the returned obs of a done env is the reset obs; the transition's s2 is final_obs
            s2 = torch.where(done[:, None], info["final_obs"], obs2) if env.cfg.auto_reset else obs2
            self.buffer.add(self._prep(obs), a, rew, done.float(), self._prep(s2))
            if env.cfg.auto_reset:
                df = done.float()
                ended.append(torch.stack(((info["final_ret"] * df).sum(), df.sum())))
                if self.device_actor is None:
                    self.noise.reset(done)      # (the device actor zeroes its OU state in-kernel: reset_on_done)
            warming = warmup_quirk and self.buffer.size() < self.min_batch
            if not warming:
                obs = obs2.clone()                  # :307 (skipped by the `continue` of :283-286 while the ring fills)
            elif env.cfg.auto_reset:
                obs = torch.where(done[:, None], obs2, obs)   # `if done: break` -> the next episode's state = env.reset()
            for _ in range(updates_per_step):
                self.update()
            if log_every and (k + 1) % log_every == 0 and ended:
                tail = torch.stack(ended[-log_every:]).sum(0).tolist()
                if tail[1] > 0:
                    print(f"step {k + 1}: mean return of the episodes finished in the last {log_every} steps {tail[0] / tail[1]:.2f}")
        if not ended:
            return []
        rows = torch.stack(ended).cpu().tolist()
        return [sm / cnt for sm, cnt in rows if cnt > 0]          # mean return of the episodes that ended at each such step

    def _train_fused_bookkeeping(self, total_steps, updates_per_step, log_every, observe, in_step_kernel=True):
        """train() without PyTorch kernels between the launches: the step kernel (policy + noise + MR_Env.step) stores the
        transitions in the ring itself and sums the finished episodes' returns (in_step_kernel), or mrsim_replay_add_step does
        after it (replay_buffer.add, `state = next_state`, the returns)."""
        env = self.env
        obs = env.reset().reshape(env.num_envs, 5)
        if not in_step_kernel:
            obs = obs.clone()
        ended = torch.zeros((max(1, total_steps), 2), dtype=torch.float32, device=env.device)
        scale = None if self.obs_scale is None else [float(x) for x in self.obs_scale.cpu()]
        for k in range(total_steps):
            if observe is not None:
                observe(k, obs)     # (in_step_kernel: the env's own observation buffer -- what the kernel's policy is evaluated on)
            if in_step_kernel:
                env.step(actor=self.device_actor, replay=self.buffer.sink(scale, ended[k]))
                self.buffer.advance(env.num_envs)
            else:
                env.step(actor=self.device_actor)
                self.buffer.add_step(obs, env._actions_out, env.rew, env._done_u8, env._obs, env._final_obs, env.final_ret, scale,
                                     ended=ended[k])
            for _ in range(updates_per_step):
                self.update()
            if log_every and (k + 1) % log_every == 0:
                tail = ended[max(0, k + 1 - log_every):k + 1].sum(0).tolist()
                if tail[1] > 0:
                    print(f"step {k + 1}: mean return of the episodes finished in the last {log_every} steps {tail[0] / tail[1]:.2f}")
        rows = ended[:total_steps].cpu().tolist()
        return [sm / cnt for sm, cnt in rows if cnt > 0]
