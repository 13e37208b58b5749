"""DeviceActor: the DDPG actor of RL/MR_ddpg.py:80-160 (+ OUNoise :59-78) as an on-device policy source of the env kernels.

The collection loop of the reference (RL/MR_ddpg.py:270-311) is
    action = actor.predict(state) + actor_noise();  next_state, reward, done, _ = env.step(action)
With N envs in lockstep the actor is a [N,5] -> [N,2] network evaluation between two env kernels; in eager PyTorch that
costs 30x the env step.  Here the network is evaluated by libmrsim.so itself:

    actor = DeviceActor.from_module(agent.actor, obs_scale=..., device=...)      # fold BN, pack, upload (4744 floats)
    env.step(actor=actor)                    # fused: one kernel = policy + MR_Env.step
    env.rollout(T, actor=actor, ...)         # fused: T steps of the collection loop in one launch
    a = actor.forward(env); env.step(a)      # gym-loop form: actor kernel, then the step kernel (same bits)
    RolloutCollector(..., policy=actor)      # the DDPG collection workload on sub-shard streams

All host arithmetic on the weights (batch-norm folding, packing into the MFMA operand layout) is done by the library
(mrsim_actor_fold_bn_host / mrsim_actor_pack_host); there is no PyTorch or CPU evaluation path in here.
"""
import ctypes as C

import numpy as np

from . import _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def fold_bn(w, b, gamma, beta, mean, var, eps):
    """tflearn batch_normalization at inference folded into the preceding fully_connected layer."""
    w, b, gamma, beta, mean, var = (_f32(x) for x in (w, b, gamma, beta, mean, var))
    rows, cols = w.shape
    w_out, b_out = np.empty_like(w), np.empty_like(b)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(_lib.lib().mrsim_actor_fold_bn_host(rows, cols, p(w), p(b), p(gamma), p(beta), p(mean), p(var), float(eps),
                                                   p(w_out), p(b_out)), "mrsim_actor_fold_bn_host")
    return w_out, b_out


def fold_actor(module, obs_scale=None):
    """Inference-form weights of an mr_rl_amd.ddpg.Actor (the PyTorch twin of ActorNetwork): eval-mode batch norm folded
    into fc1 / fc2.  Returns a dict of float32 numpy arrays: w1 [64,5], b1, w2 [64,64], b2, w3 [2,64], b3, obs_scale [5],
    action_bound [2]."""
    def g(t):
        return t.detach().cpu().numpy()
    w1, b1 = fold_bn(g(module.fc1.weight), g(module.fc1.bias), g(module.bn1.weight), g(module.bn1.bias),
                     g(module.bn1.running_mean), g(module.bn1.running_var), module.bn1.eps)
    w2, b2 = fold_bn(g(module.fc2.weight), g(module.fc2.bias), g(module.bn2.weight), g(module.bn2.bias),
                     g(module.bn2.running_mean), g(module.bn2.running_var), module.bn2.eps)
    return {"w1": w1, "b1": b1, "w2": w2, "b2": b2, "w3": _f32(g(module.out.weight)), "b3": _f32(g(module.out.bias)),
            "obs_scale": _f32(np.ones(5) if obs_scale is None else obs_scale), "action_bound": _f32(g(module.action_bound))}


def pack_weights(w):
    """The library's packed parameter block (host numpy, MRSIM_ACTOR_BLOB_FLOATS floats) of an inference-form dict."""
    H = _lib.ACTOR_HIDDEN
    arrs = {k: _f32(w[k]) for k in ("w1", "b1", "w2", "b2", "w3", "b3")}
    assert arrs["w1"].shape == (H, 5) and arrs["w2"].shape == (H, H) and arrs["w3"].shape == (2, H)
    assert arrs["b1"].shape == (H,) and arrs["b2"].shape == (H,) and arrs["b3"].shape == (2,)
    W = _lib.MrsimActorWeights()
    for k, a in arrs.items():
        setattr(W, k, a.ctypes.data)
    for i, v in enumerate(_f32(w.get("obs_scale", np.ones(5))).reshape(5)):
        W.obs_scale[i] = float(v)
    for i, v in enumerate(_f32(w["action_bound"]).reshape(2)):
        W.action_bound[i] = float(v)
    blob = np.zeros(_lib.ACTOR_BLOB_FLOATS, dtype=np.float32)
    _lib.check(_lib.lib().mrsim_actor_pack_host(C.byref(W), blob.ctypes.data_as(C.c_void_p)), "mrsim_actor_pack_host")
    return blob


class DeviceActor:
    """The actor's parameters in HBM (packed block) + one OUNoise process per env.

    ou: False = actor.predict alone; True = + Ornstein-Uhlenbeck noise (theta, sigma, dt: the reference's defaults).
    reset_on_done: False = the reference (RL/MR_ddpg.py never resets actor_noise); True = x_prev := 0 at the first step of
    every episode (what mr_rl_amd.ddpg.DDPG.train does with its own OUNoise).
    The OU state tensor [num_envs, 2] is created on first use for the env count it is used with."""

    MATH = {"f32": _lib.ACTOR_F32, "bf16x3": _lib.ACTOR_BF16X3, "bf16": _lib.ACTOR_BF16}

    def __init__(self, weights, device="cuda", ou=True, theta=0.15, sigma=0.3, dt=1e-2, reset_on_done=False, math="f32"):
        """math: arithmetic of the two hidden layers.  "f32" (default) = exact f32 MFMA, bit-for-bit the documented fmaf chain;
        "bf16x3" = every f32 operand as three bf16 terms, six bf16 MFMAs with f32 accumulation: f32-class accuracy (within
        5e-6 of the action bound of the f32 result; tests/test_gpu_actor.py), about 1.5 x the collection rate; "bf16" = plain
        bf16 operands, ordinary bf16 inference (the action within 4e-5 of its bound of the f32 result with the reference's
        U[-3e-3, 3e-3] output layer, up to 3e-2 at output gains of 20-40 x), about 3.4 x: exploration-grade collection."""
        import torch
        if math not in self.MATH:
            raise ValueError("math must be 'f32', 'bf16x3' or 'bf16'")
        self.math = math
        self.device = torch.device(device)
        self.ou, self.theta, self.sigma, self.dt = bool(ou), float(theta), float(sigma), float(dt)
        self.reset_on_done = bool(reset_on_done)
        self.blob = torch.empty(_lib.ACTOR_BLOB_FLOATS, dtype=torch.float32, device=self.device)
        self._ou = {}          # env count -> [n, 2] OU state (prepared launches hold raw pointers into these: never reallocated)
        self.ou_state = None   # the one used last
        self.weights = None
        self.load(weights, sync=False)

    @classmethod
    def from_module(cls, module, obs_scale=None, **kw):
        return cls(fold_actor(module, obs_scale), **kw)

    def load(self, weights, sync=True):
        """(Re)upload the parameters, e.g. after a learner update.  Kernels stage the block into LDS when their workgroups
        start, so a launch that is still running (on any stream: RolloutCollector's sub-shard streams) while the block is
        overwritten would mix old and new parameters: sync=True (default) waits for the device first; pass sync=False only
        when no launch that reads the block can be in flight (e.g. right after RolloutCollector.join() + a stream wait)."""
        import torch
        self.weights = {k: _f32(v) for k, v in weights.items()}
        host = torch.from_numpy(pack_weights(self.weights))
        if sync:
            torch.cuda.synchronize(self.device)
        self.blob.copy_(host, non_blocking=False)

    def load_module(self, module, obs_scale=None, sync=True):
        self.load(fold_actor(module, self.weights["obs_scale"] if obs_scale is None else obs_scale), sync=sync)

    def ou_tensor(self, n):
        import torch
        if not self.ou:
            return None
        if n not in self._ou:
            self._ou[n] = torch.zeros((n, 2), dtype=torch.float32, device=self.device)
        self.ou_state = self._ou[n]
        return self.ou_state

    def reset_noise(self):
        for t in self._ou.values():
            t.zero_()

    def struct(self, n, first=0, count=None):
        """MrsimActor for the envs [first, first + count) of an n-env set (the OU state pointer advanced to `first`)."""
        ou = self.ou_tensor(n)
        return _lib.MrsimActor(self.blob.data_ptr(), None if ou is None else ou.data_ptr() + first * 8, self.theta,
                               self.sigma, self.dt, int(self.reset_on_done), self.MATH[self.math], 0)

    def forward(self, env, obs=None, out=None):
        """actions[N,2] = actor.predict(obs) + actor_noise() as a kernel of its own, for env's NEXT step (same RNG words
        as the fused forms: feed the result to env.step()).  obs defaults to the env's current observation buffer."""
        import torch
        n = env.num_envs
        if out is None:
            out = torch.empty((n, 2), dtype=torch.float32, device=env.device)
        src = env._obs if obs is None else obs
        a = self.struct(n)
        rc = env._L.mrsim_actor_forward(C.byref(env._params), n, env.env_id0, C.byref(a), C.byref(env._st), env._p(src),
                                        env._p(out), env.seed_value, env.step_idx, env._stream())
        _lib.check(rc, "mrsim_actor_forward")
        return out
