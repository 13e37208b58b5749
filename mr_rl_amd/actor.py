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

    def __init__(self, weights, device="cuda", ou=True, theta=0.15, sigma=0.3, dt=1e-2, reset_on_done=False, math="f32",
                 slots=1):
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
        # `slots` parameter blocks: a learner uploads the next parameters into the block no launch reads while the envs still
        # collect with the current one (DDPG.train_collected: episode k reads block k % slots); load() fills all of them
        self.blobs = [torch.empty(_lib.ACTOR_BLOB_FLOATS, dtype=torch.float32, device=self.device) for _ in range(max(1, int(slots)))]
        self.blob = self.blobs[0]
        self._perm = None      # device gather indices of the packed layout (load_module_device)
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
        for b in self.blobs:
            b.copy_(host, non_blocking=False)

    def load_module(self, module, obs_scale=None, sync=True):
        self.load(fold_actor(module, self.weights["obs_scale"] if obs_scale is None else obs_scale), sync=sync)

    # ------------------------------------------------------------------------------------------------------------
    # device-side upload: fold batch norm, pack into the kernels' operand layout and write a parameter block with torch ops on
    # the CURRENT stream -- no host copy of the weights, no device-wide synchronisation.  Same bits as load_module() (tested).
    def _layout(self):
        """Gather indices of the packed block, derived once from the library's own packer (mrsim_actor_pack_host) by packing
        index codes instead of weights: the layout stays private to the library."""
        import torch
        if self._perm is not None:
            return self._perm
        H = _lib.ACTOR_HIDDEN
        sizes = [("w1", H * 5), ("b1", H), ("w2", H * H), ("b2", H), ("w3", 2 * H), ("b3", 2), ("action_bound", 2)]
        off, codes, base = {}, {}, 1                       # code 0 = a padding zero of the block
        for k, n in sizes:
            off[k] = base
            codes[k] = np.arange(base, base + n, dtype=np.float32)
            base += n
        shp = {"w1": (H, 5), "w2": (H, H), "w3": (2, H)}
        w = {k: codes[k].reshape(shp.get(k, (-1,))) for k, _ in sizes}
        w["obs_scale"] = np.ones(5, dtype=np.float32)
        blob = pack_weights(w)
        F32 = _lib.ACTOR_BLOB_FLOATS - (H * H * 3) // 2    # the bf16 x 3 section is the tail of the block: 3 bf16 terms per weight
        perm_f32 = np.rint(blob[:F32]).astype(np.int64)    # index codes < 2^24 are exact in float32
        # bf16 section: term 0 of a weight w2[i] sits where the packer puts bf16(w2[i]); integers <= 256 are exact in bf16, so two
        # passes with the base-64 digits of i recover it.  Terms 1 and 2 of the same weight follow at +512 and +1024 halfwords.
        digs = []
        for d in (lambda i: i // 64, lambda i: i % 64):
            w2c = dict(w)
            w2c["w2"] = d(np.arange(H * H)).astype(np.float32).reshape(H, H)
            hw = pack_weights(w2c)[F32:].view(np.uint16).astype(np.uint32) << 16
            digs.append(np.rint(hw.view(np.float32)).astype(np.int64))
        idx = digs[0] * 64 + digs[1]                       # per halfword; meaningful at term-0 positions
        hw_n = idx.size
        pos = np.arange(hw_n)
        part = (pos // 512) % 3
        idx0 = idx[pos - part * 512]                       # the weight whose term `part` lives at this halfword
        dev = self.device
        self._perm = {"f32": torch.from_numpy(perm_f32).to(dev), "bf_idx": torch.from_numpy(idx0).to(dev),
                      "bf_part": torch.from_numpy(part.astype(np.int64)).to(dev), "off": off, "n_f32": F32}
        return self._perm

    def load_module_device(self, module, slot=None):
        """(Re)write parameter block `slot` (default: all) from an mr_rl_amd.ddpg.Actor on the device, ordered on the current
        stream.  The caller orders it against the launches that read the block (RolloutCollector.collect(after=stream))."""
        import torch
        L = self._layout()
        dev = self.device
        f64 = torch.float64

        def fold(fc, bn):   # mrsim_actor_fold_bn_host's arithmetic: double, rounded once
            eps = torch.tensor(bn.eps, dtype=torch.float32, device=dev).to(f64)
            g = bn.weight.detach().to(f64) / torch.sqrt(bn.running_var.to(f64) + eps)
            w = (fc.weight.detach().to(f64) * g[:, None]).float()
            b = ((fc.bias.detach().to(f64) - bn.running_mean.to(f64)) * g + bn.bias.detach().to(f64)).float()
            return w, b
        w1, b1 = fold(module.fc1, module.bn1)
        w2, b2 = fold(module.fc2, module.bn2)
        scale = torch.as_tensor(self.weights["obs_scale"], dtype=torch.float32, device=dev)
        flat = torch.cat([torch.zeros(1, device=dev), (w1 * scale[None, :]).reshape(-1), b1, w2.reshape(-1), b2,
                          module.out.weight.detach().float().reshape(-1), module.out.bias.detach().float(),
                          module.action_bound.float()])
        f32 = flat[L["f32"]]
        w2g = w2.reshape(-1)[L["bf_idx"]]
        t0 = w2g.to(torch.bfloat16)
        r1 = w2g - t0.float()
        t1 = r1.to(torch.bfloat16)
        t2 = (r1 - t1.float()).to(torch.bfloat16)
        hw = torch.where(L["bf_part"] == 0, t0, torch.where(L["bf_part"] == 1, t1, t2))
        block = torch.cat([f32, hw.view(torch.float32)])
        for i, b in enumerate(self.blobs):
            if slot is None or i == slot:
                b.copy_(block)

    def load_from_learner(self, fused, slot=None):
        """(Re)write parameter block `slot` (default: all) from a FusedLearner's online actor: ONE launch of the library's
        device-side fold + pack (mrsim_actor_pack_device), ordered on the current stream."""
        import torch
        L = _lib.lib()
        scale = (C.c_float * 5)(*[float(x) for x in self.weights["obs_scale"]])
        bound = (C.c_float * 2)(*[float(x) for x in self.weights["action_bound"]])
        strm = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for i, b in enumerate(self.blobs):
            if slot is None or i == slot:
                _lib.check(L.mrsim_actor_pack_device(C.c_void_p(fused.online.data_ptr()), C.c_void_p(fused.bn_stats.data_ptr()),
                                                     float(fused.struct.bn_eps), scale, bound, C.c_void_p(b.data_ptr()), strm),
                           "mrsim_actor_pack_device")

    def ou_tensor(self, n):
        import torch
        if not self.ou:
            return None
        if n not in self._ou:
            self._ou[n] = torch.zeros((n, 2), dtype=torch.float32, device=self.device)
            # The zero fill runs on the CURRENT stream and the kernels that read and write this state run on others (the
            # collector's sub-shard streams): wait for it here, once per env count, so that no launch on any stream can see the
            # buffer before the fill has landed (an unordered cross-stream initialisation).
            torch.cuda.current_stream(self.device).synchronize()
        self.ou_state = self._ou[n]
        return self.ou_state

    def state_dict(self):
        """parameters (inference form) + every OU state tensor: an actor-in-the-loop run resumes bit for bit together with
        MRVecEnv.state_dict()"""
        return {"weights": {k: v.copy() for k, v in self.weights.items()}, "blob": self.blob.detach().cpu().clone(),
                "ou": {int(n): t.detach().cpu().clone() for n, t in self._ou.items()}}

    def load_state_dict(self, sd):
        import torch
        self.weights = {k: _f32(v) for k, v in sd["weights"].items()}
        torch.cuda.synchronize(self.device)
        for b in self.blobs:
            b.copy_(sd["blob"])
        for n, t in sd["ou"].items():
            self.ou_tensor(int(n)).copy_(t)
        torch.cuda.synchronize(self.device)

    def reset_noise(self):
        for t in self._ou.values():
            t.zero_()

    def struct(self, n, first=0, count=None, slot=0):
        """MrsimActor for the envs [first, first + count) of an n-env set (the OU state pointer advanced to `first`), reading
        parameter block `slot`."""
        ou = self.ou_tensor(n)
        return _lib.MrsimActor(self.blobs[slot % len(self.blobs)].data_ptr(), None if ou is None else ou.data_ptr() + first * 8, self.theta,
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
