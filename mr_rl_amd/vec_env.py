"""MRVecEnv: N micro-robot environments advanced in lockstep by the HIP step kernel.

Host-side mirror of the reference's gym API (class MR_Env, MR_env.py:21-229) for a
batch: same method names, argument meaning and defaults, tensors instead of scalars.
All compute happens in libmrsim.so on the GPU; torch is only the owner of device
memory and streams.  There is no CPU path: constructing an env without a HIP device
raises.
"""
import ctypes as C

import numpy as np

from . import _lib
from .config import MRConfig
from .spaces import make_box


def _torch():
    import torch
    return torch


class SimulatorView:
    """`env.simulator.{noise_var,a0,is_mismatched,state_prime}` as the reference exposes them
    (MR_env.py:179-183, utils.py:54)."""

    def __init__(self, env):
        self._env = env

    noise_var = property(lambda s: s._env.cfg.noise_var)
    a0 = property(lambda s: s._env.cfg.a0)
    is_mismatched = property(lambda s: s._env.cfg.is_mismatched)
    state_prime = property(lambda s: s._env.state_prime)


class MRVecEnv:
    """N independent MR_Env instances on one GPU.

    reset(mask=None, init=None, noise_var=None, a0=None, is_mismatched=None) -> obs[N,5]
    step(actions[N,2] | None) -> (obs[N,5] f32, rew[N] f32, done[N] bool, info)

    env_id0 / seed make the RNG stream a function of the GLOBAL env id, so a shard of a
    larger job reproduces exactly the trajectories the unsharded job would produce.
    """

    metadata = {"render.modes": []}

    def __init__(self, num_envs, cfg=None, device="cuda", seed=None, env_id0=0, goal_table=None,
                 track_state_prime=False, track_actions=False, track_attempts=False):
        torch = _torch()
        self._L = _lib.lib()  # raises ImportError if the HIP extension is not built
        if not torch.cuda.is_available() or self._L.mrsim_device_count() <= 0:
            raise RuntimeError("MRVecEnv needs a HIP device (MI355X); mr_rl_amd has no CPU fallback")
        self.cfg = cfg if cfg is not None else MRConfig()
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MRVecEnv: device must be a cuda(HIP) device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.seed_value = int(self.cfg.seed if seed is None else seed)
        self.env_id0 = int(env_id0)
        self.step_idx = 0  # global RNG step index; every reset()/step() call consumes one
        n, dev = self.num_envs, self.device

        # spaces (MR_env.py:34-45)
        self.action_space = make_box(self.cfg.action_low, self.cfg.action_high)
        self.observation_space = make_box(self.cfg.obs_low, self.cfg.obs_high)
        self.init_space = make_box(self.cfg.init_low, self.cfg.init_high, seed=self.seed_value)
        self.max_timesteps = self.cfg.max_timesteps
        self.min_dist2goal = self.cfg.min_dist2goal

        # goal / reference-trajectory table [K][T][2]; None = MR_Env.init_goal = (0,0) (MR_env.py:57)
        self.goal_table = None
        self._gK = self._gT = 1
        if goal_table is not None:
            g = torch.as_tensor(goal_table, dtype=torch.float32, device=dev).contiguous()
            if g.dim() == 2:
                g = g.unsqueeze(0)
            assert g.dim() == 3 and g.shape[2] == 2, "goal_table must be [K][T][2]"
            self.goal_table, self._gK, self._gT = g, int(g.shape[0]), int(g.shape[1])
        self.init_goal = np.zeros(2)

        # persistent state (include/mrsim.h: MrsimState)
        self.pos = torch.zeros((n, 2), dtype=torch.float64, device=dev)   # == last_pos
        self.aux = torch.zeros((n, 4), dtype=torch.float32, device=dev)   # f0x, f0y, h_abs/dt, counter(bits)
        self.aux[:, 2] = 1.0
        self.ep_ret = torch.zeros(n, dtype=torch.float32, device=dev)
        # step outputs
        self._soa = self.cfg.obs_layout == "soa"
        oshape = (5, n) if self._soa else (n, 5)
        self._obs = torch.zeros(oshape, dtype=torch.float32, device=dev)
        self._final_obs = torch.zeros(oshape, dtype=torch.float32, device=dev)
        self.rew = torch.zeros(n, dtype=torch.float32, device=dev)
        self._done_u8 = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.final_ret = torch.zeros(n, dtype=torch.float32, device=dev)
        self.final_len = torch.zeros(n, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._state_prime = torch.zeros((n, 2), dtype=torch.float32, device=dev) if track_state_prime else None
        self._actions_out = torch.zeros((n, 2), dtype=torch.float32, device=dev) if track_actions else None
        # rk_step attempts of the latest step() per env (MrsimStepIO.attempts; = (integrator.nfev after - before) / 6)
        self.attempts = torch.zeros(n, dtype=torch.int32, device=dev) if track_attempts else None
        self.last_action = None
        self._prev_mismatched = False  # what Simulator.is_mismatched was before the latest reset (MR_env.py:181-183)
        self._params = None
        self._refresh_params()
        self._st = _lib.MrsimState(self.pos.data_ptr(), self.aux.data_ptr(), self.ep_ret.data_ptr())
        self.simulator = SimulatorView(self)

    # ------------------------------------------------------------------ helpers
    def _refresh_params(self):
        self._params = self.cfg.to_params(self._gK, self._gT)
        self._params_version = getattr(self, "_params_version", 0) + 1  # prepared launches hold the old block
        sb = getattr(self, "_step_base", None)
        self._params.step_base = None if sb is None else sb.data_ptr()

    def enable_device_step_base(self):
        """Move the RNG step counter into HBM (MrsimParams.step_base): the step_idx argument of every
        later call becomes an OFFSET from that device word.  Needed to replay a captured hipGraph with
        fresh noise (kernel arguments are frozen at capture)."""
        torch = _torch()
        if getattr(self, "_step_base", None) is None:
            self._step_base = torch.full((1,), self.step_idx, dtype=torch.int64, device=self.device)
            self.step_idx = 0
            self._refresh_params()
        return self._step_base

    def advance_step_base(self, delta):
        rc = self._L.mrsim_advance_step_base(self._p(self._step_base), int(delta), self._stream())
        _lib.check(rc, "mrsim_advance_step_base")

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _view(self, t):
        return t.t() if self._soa else t

    @property
    def obs(self):
        return self._view(self._obs)

    @property
    def final_obs(self):
        return self._view(self._final_obs)

    @property
    def done(self):
        return self._done_u8.view(_torch().bool)

    @property
    def last_pos(self):
        """[N,2] fp64 positions (MR_Env.last_pos, MR_env.py:91; read by utils.run_sim)."""
        return self.pos

    @property
    def counter(self):
        return self.aux[:, 3].view(_torch().int32)

    @property
    def state_prime(self):
        if self._state_prime is None:
            raise AttributeError("construct MRVecEnv(track_state_prime=True) to record Simulator.state_prime")
        return self._state_prime

    def seed(self, seed=None):
        """keras-rl era callers use env.seed(n) (old/MR_dqn_keras_rl.py:19)."""
        if seed is not None:
            self.seed_value = int(seed)
            self.init_space.seed(self.seed_value)   # MR_Env.reset(init=None) samples from it (MR_env.py:172-173)
        return [self.seed_value]

    # ------------------------------------------------------------------ gym API
    def reset(self, mask=None, init=None, noise_var=None, a0=None, is_mismatched=None):
        """MR_Env.reset for the masked envs (all when mask is None).  kwargs left at None keep the
        current cfg (the reference's defaults noise_var=1, a0=1, is_mismatched=False are MRConfig's)."""
        torch = _torch()
        if noise_var is not None:
            self.cfg.noise_var = float(noise_var)
        if a0 is not None:
            self.cfg.a0 = float(a0)
        ctor_mis = self._prev_mismatched
        if is_mismatched is not None:
            self.cfg.is_mismatched = bool(is_mismatched)
        self._prev_mismatched = self.cfg.is_mismatched
        self._refresh_params()
        mask_t = None
        if mask is not None:
            mask_t = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            assert mask_t.shape == (self.num_envs,)
        init_t = None
        if init is not None:
            init_t = torch.as_tensor(init, dtype=torch.float64, device=self.device)
            if init_t.dim() == 1:
                init_t = init_t.expand(self.num_envs, 2)
            init_t = init_t.contiguous()
            assert init_t.shape == (self.num_envs, 2)
        rc = self._L.mrsim_reset(C.byref(self._params), self.num_envs, self.env_id0, C.byref(self._st),
                                 self._p(mask_t), self._p(init_t), self._p(self.goal_table), self._p(self._obs),
                                 int(ctor_mis), self.seed_value, self.step_idx, self._stream())
        _lib.check(rc, "mrsim_reset")
        self.step_idx += 1
        return self.obs

    def _step_io(self, act_t, actor=None, replay=None):
        io = _lib.MrsimStepIO(
            self._p(act_t), self._p(self._actions_out), self._p(self.goal_table), self._p(self._obs),
            self._p(self.rew), self._p(self._done_u8), self._p(self._state_prime), self._p(self._final_obs),
            self._p(self.final_ret), self._p(self.final_len), self._p(self.status), None, self._p(self.attempts))
        if actor is not None:
            self._actor_struct = actor.struct(self.num_envs)   # keeps the ctypes block alive through the call
            io.actor = C.pointer(self._actor_struct)
        if replay is not None:
            io.replay = C.pointer(replay)     # (the caller keeps the struct alive: ReplayBuffer.sink)
        return io

    def step(self, actions=None, actor=None, replay=None):
        """MR_Env.step for all envs.  actions: [N,2] float32 device tensor {f_t, alpha_t}; or actor=DeviceActor: the action
        is actor.predict(obs) + actor_noise() evaluated inside the step kernel on the env's current observation
        (RL/MR_ddpg.py:277-278 in one launch); or neither, to draw the uniform random policy in-kernel
        (cfg.policy_low/high).  replay (with actor): a _lib.MrsimReplaySink (ddpg.ReplayBuffer.sink) -- the step kernel writes
        the transitions into that ring itself (RL/MR_ddpg.py:278-282 without a launch of its own)."""
        torch = _torch()
        act_t = None
        if actions is not None and actor is not None:
            raise ValueError("step: pass actions or actor, not both")
        if replay is not None and actor is None:
            raise ValueError("step: replay needs actor (the kernel stores the observation its own policy saw)")
        if actions is not None:
            act_t = actions if (torch.is_tensor(actions) and actions.dtype == torch.float32 and
                                actions.device == self.device and actions.is_contiguous()) else \
                torch.as_tensor(actions, dtype=torch.float32, device=self.device).contiguous()
            if act_t.shape != (self.num_envs, 2):
                raise IndexError(f"actions must have shape ({self.num_envs}, 2), got {tuple(act_t.shape)}")
            self.last_action = act_t
        elif self._actions_out is not None:
            self.last_action = self._actions_out
        io = self._step_io(act_t, actor, replay)
        rc = self._L.mrsim_step(C.byref(self._params), self.num_envs, self.env_id0, C.byref(self._st), C.byref(io),
                                self.seed_value, self.step_idx, self._stream())
        _lib.check(rc, "mrsim_step")
        self.step_idx += 1
        info = {}
        if self.cfg.auto_reset:
            info = {"final_obs": self.final_obs, "final_ret": self.final_ret, "final_len": self.final_len}
        return self.obs, self.rew, self.done, info

    def _step_shard(self, first, n, act_ptr, pol_ptr, step_idx, stream):
        """One MR_Env.step of envs [first, first + n) on `stream` (a torch stream): the C entry points take a count, the
        global id of the first env and plain pointers, so a sub-shard is the same call with offset pointers.  pol_ptr: where
        the policy kernel writes this shard's actions first (None: actions come from act_ptr or are drawn in the step
        kernel).  [N][5] observations only (the [5][N] layout strides by the launch's own n)."""
        L, sp = self._L, C.c_void_p(stream.cuda_stream)

        def off(t, nbytes):
            return None if t is None else C.c_void_p(t.data_ptr() + first * nbytes)
        if pol_ptr is not None:
            _lib.check(L.mrsim_random_policy(C.byref(self._params), n, self.env_id0 + first, pol_ptr, self.seed_value,
                                             step_idx, sp), "mrsim_random_policy")
            act_ptr = pol_ptr
        st = _lib.MrsimState(off(self.pos, 16), off(self.aux, 16), off(self.ep_ret, 4))
        io = _lib.MrsimStepIO(act_ptr, off(self._actions_out, 8), self._p(self.goal_table), off(self._obs, 20),
                              off(self.rew, 4), off(self._done_u8, 1), off(self._state_prime, 8), off(self._final_obs, 20),
                              off(self.final_ret, 4), off(self.final_len, 4), self._p(self.status), None, off(self.attempts, 4))
        _lib.check(L.mrsim_step(C.byref(self._params), n, self.env_id0 + first, C.byref(st), C.byref(io), self.seed_value,
                                step_idx, sp), "mrsim_step")

    def step_timed(self, actions=None):
        """One step whose kernel duration (ms) is measured with HIP events attached to the dispatch.
        Measurement aid for bench.py; synchronises the stream."""
        torch = _torch()
        act_t = None if actions is None else torch.as_tensor(actions, dtype=torch.float32, device=self.device).contiguous()
        io = self._step_io(act_t)
        ms = C.c_float(0.0)
        rc = self._L.mrsim_step_timed(C.byref(self._params), self.num_envs, self.env_id0, C.byref(self._st),
                                      C.byref(io), self.seed_value, self.step_idx, self._stream(), C.byref(ms))
        _lib.check(rc, "mrsim_step_timed")
        self.step_idx += 1
        return ms.value

    def random_policy(self, out=None, lookahead=0):
        """actions[N,2] ~ U[policy_low, policy_high) on device (consumes no step index: it is keyed by the
        step it feeds -- the next step() by default, the one `lookahead` steps later otherwise)."""
        torch = _torch()
        if out is None:
            out = torch.empty((self.num_envs, 2), dtype=torch.float32, device=self.device)
        rc = self._L.mrsim_random_policy(C.byref(self._params), self.num_envs, self.env_id0, self._p(out),
                                         self.seed_value, self.step_idx + int(lookahead), self._stream())
        _lib.check(rc, "mrsim_random_policy")
        return out

    def random_policy_steps(self, T, out=None):
        """The exploration policy of the next T steps in one launch: actions_T[T,N,2], row t equal to what
        random_policy() returns right before step t (the policy reads no state, so a whole episode can be drawn
        ahead of its steps)."""
        torch = _torch()
        if out is None:
            out = torch.empty((int(T), self.num_envs, 2), dtype=torch.float32, device=self.device)
        if tuple(out.shape) != (int(T), self.num_envs, 2) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("random_policy_steps: out must be a contiguous float32 [T, N, 2] tensor")
        rc = self._L.mrsim_random_policy_steps(C.byref(self._params), self.num_envs, self.env_id0, self._p(out), int(T),
                                               self.seed_value, self.step_idx, self._stream())
        _lib.check(rc, "mrsim_random_policy_steps")
        return out

    def rollout(self, T, actions=None, shared_actions=False, want=("traj",), out=None, timed=False, events=None,
                carry=None, actor=None):
        """T fused steps in one launch (batched utils.run_sim / DDPG rollout).  actions: [T,N,2], or [T,2]
        with shared_actions=True, or None for the on-device random policy; a float64 array / tensor is
        passed on as fp64 (the reference's action tables are float64, main.py:14-50), anything else as fp32.
        actor: a DeviceActor instead of actions -- every step's action is actor.predict(obs) + actor_noise() on the
        observation of the previous step, evaluated in-kernel (the collection loop of RL/MR_ddpg.py:270-311).
        want: any of "traj" (fp64 positions [T,N,2]), "state_prime", "obs", "rew", "done", "actions".
        `out` lets a caller reuse the [T,...] buffers of a previous call (the returned dict).
        carry: "f32" (default, cfg.rollout_carry) rounds the carried RK45 state to its HBM format after every
        step -- bit-identical to T calls of step(); "f64" keeps it in fp64 registers for the whole launch.
        Returns a dict of [T,...] tensors.
        Measurement aids: timed=True synchronises and returns "kernel_ms"; events=_lib.EventPair() attaches the
        pair to the dispatch without synchronising (read it later with .elapsed_ms())."""
        torch = _torch()
        n, dev = self.num_envs, self.device
        act_t, act64 = None, False
        if actions is not None:
            act64 = (torch.is_tensor(actions) and actions.dtype == torch.float64) or \
                    (isinstance(actions, np.ndarray) and actions.dtype == np.float64)
            act_t = torch.as_tensor(actions, dtype=torch.float64 if act64 else torch.float32, device=dev).contiguous()
            assert act_t.shape == ((T, 2) if shared_actions else (T, n, 2))
        carry = self.cfg.rollout_carry if carry is None else carry
        if carry not in ("f32", "f64"):
            raise ValueError("carry must be 'f32' or 'f64'")
        buf = out if out is not None else {}

        def get(key, shape, dtype):
            if key not in want:
                return None
            t = buf.get("_" + key)
            if t is None or tuple(t.shape) != tuple(shape):
                t = buf["_" + key] = torch.empty(shape, dtype=dtype, device=dev)
            return t

        traj = get("traj", (T, n, 2), torch.float64)
        sp_T = get("state_prime", (T, n, 2), torch.float32)
        obs_T = get("obs", (T, 5, n) if self._soa else (T, n, 5), torch.float32)
        rew_T = get("rew", (T, n), torch.float32)
        done_T = get("done", (T, n), torch.uint8)
        acts_T = get("actions", (T, n, 2), torch.float32)
        self.launch_rollout(T, 0, n, act_t=act_t, shared_actions=shared_actions, act64=act64, traj=traj, sp_T=sp_T,
                            obs_T=obs_T, rew_T=rew_T, done_T=done_T, acts_T=acts_T, carry=carry, timed_into=buf if timed else None,
                            events=events, actor=actor)
        self.step_idx += int(T)
        if traj is not None:
            buf["traj"] = traj
        if sp_T is not None:
            buf["state_prime"] = sp_T
        if obs_T is not None:
            buf["obs"] = obs_T.transpose(1, 2) if self._soa else obs_T
        if rew_T is not None:
            buf["rew"] = rew_T
        if done_T is not None:
            buf["done"] = done_T.view(torch.bool)
        if acts_T is not None:
            buf["actions"] = acts_T
        return buf

    def launch_rollout(self, T, first, n, act_t=None, shared_actions=False, act64=False, traj=None, sp_T=None, obs_T=None,
                       rew_T=None, done_T=None, acts_T=None, final_ret=None, final_len=None, carry="f32", step_idx=None,
                       stream=None, timed_into=None, events=None, prepare_only=False, actor=None, actor_slot=0):
        """One mrsim_rollout launch over the envs [first, first + n) of this env set (a sub-shard when n < num_envs):
        state, final_* and every [T, N, ...] buffer are passed advanced to env `first`, the buffers keep their row
        length N (MrsimRolloutIO.row_stride), the RNG keys stay the GLOBAL env ids.  Does not advance step_idx (the
        caller launches all sub-shards of a rollout with the same step_idx and advances once).  mr_rl_amd.collector
        uses it to put sub-shards on different HIP streams.  Returns the prepared launch `f(step_idx, events=None)`:
        with prepare_only=True nothing is launched now and a caller that repeats the same launch every episode (same
        buffers, new step index) pays only the C call, not the construction of the argument structures."""
        N = self.num_envs
        assert 0 <= first and n >= 1 and first + n <= N
        fr = self.final_ret if final_ret is None else final_ret
        fl = self.final_len if final_len is None else final_len

        def P(t, per_env_elems=None, soa_obs=False):
            """device pointer of t advanced to env `first`; t is [T, N, k] / [T, N] (or [T, 5, N] for SoA observations),
            or [N, ...] when per_env_elems is given"""
            if t is None:
                return None
            if per_env_elems is not None:
                return t.data_ptr() + first * per_env_elems * t.element_size()
            if soa_obs:
                return t.data_ptr() + first * t.element_size()
            return t.data_ptr() + first * int(np.prod(t.shape[2:], dtype=np.int64)) * t.element_size()

        st = _lib.MrsimState(P(self.pos, 2), P(self.aux, 4), P(self.ep_ret, 1))
        a_ptr = None
        if act_t is not None:
            a_ptr = act_t.data_ptr() if shared_actions else P(act_t)
        io = _lib.MrsimRolloutIO(int(T), int(bool(shared_actions)), a_ptr, None if self.goal_table is None else
                                 self.goal_table.data_ptr(), P(traj), P(sp_T), P(obs_T, soa_obs=self._soa), P(rew_T),
                                 P(done_T), P(acts_T), P(fr, 1), P(fl, 1), self.status.data_ptr(),
                                 0 if n == N else N, int(carry == "f64"), int(act64))
        actor_struct = None
        if actor is not None:
            if act_t is not None:
                raise ValueError("launch_rollout: pass actions or actor, not both")
            actor_struct = actor.struct(N, first, n, slot=actor_slot)   # OU state advanced to env `first`; kept alive by the closure
            io.actor = C.pointer(actor_struct)
        strm = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        head = (C.byref(self._params), n, self.env_id0 + first, C.byref(st), C.byref(io), self.seed_value)
        L = self._L

        def launch(step_idx, events=None, timed_into=None):
            """the prepared launch; the ctypes structures above stay alive in this closure"""
            _keep = actor_struct  # noqa: F841
            if timed_into is not None:
                ms = C.c_float(0.0)
                _lib.check(L.mrsim_rollout_timed(*head, int(step_idx), strm, C.byref(ms)), "mrsim_rollout_timed")
                timed_into["kernel_ms"] = ms.value
            elif events is not None:
                _lib.check(L.mrsim_rollout_events(*head, int(step_idx), strm, events.start, events.stop), "mrsim_rollout_events")
            else:
                _lib.check(L.mrsim_rollout(*head, int(step_idx), strm), "mrsim_rollout")

        launch.io = io    # (a caller that rotates final_ret / final_len rows re-points them here instead of preparing a launch per row)
        if prepare_only:
            return launch
        launch(self.step_idx if step_idx is None else int(step_idx), events=events, timed_into=timed_into)
        return launch

    def capture_steps(self, G, policy="kernel", shards=1):
        """Capture G env steps into one hipGraph (torch.cuda.CUDAGraph is only the capture plumbing).
        policy: "kernel" = policy kernel writes actions to HBM, step kernel reads them (the shape of a
        real actor -> env.step(actions) loop); "overlap" = the same two kernels, but the exploration policy of step
        t + 1 -- which depends on nothing but (seed, step index, env id) -- runs on a second captured stream while the
        step kernel of step t runs (two rotating action buffers).  Measured SLOWER on MI355X / ROCm 7.2 (14.7 vs 9.1 us per
        step at N = 262 144: graph replay pays ~5 us for every cross-stream dependency), kept for the record and tested;
        "episode" = ONE policy launch draws the G rows of actions ahead of the G step kernels (exploration only: the policy
        reads no state; 8 B x G x N of HBM, 107 MB at G = 51, N = 262 144), each step kernel reads its row;
        "fused" = the step kernel draws the policy itself.
        shards > 1 (not with "overlap"; [N][5] observations): the envs are cut into that many contiguous sub-shards (at
        multiples of 256) whose G-step chains are captured on streams of their own, forked at the start of the graph and
        joined at its end -- independent envs need no per-step dependency, and one shard's loads and stores run beside
        another's arithmetic (a step kernel over all envs is three phases in lockstep: load, compute, store).  Measured at
        N = 262 144, two shards: 40.1 vs 37.8 G env-steps/s with policy="episode", 39.2 vs 40.6 G with "fused", 17.3 vs 31.5 G with
        "kernel" (four launches per step): each half-size kernel still pays the full ~6 us launch-to-drain latency, so
        the step path is bound by that latency, not by the phases -- kept as a tested option, not used by bench.py.
        Each replay advances the device step base by G, so replays draw fresh noise."""
        torch = _torch()
        self.enable_device_step_base()
        if policy not in ("kernel", "overlap", "episode", "fused"):
            raise ValueError(f"capture_steps: unknown policy mode {policy!r}")
        acts = [torch.empty((self.num_envs, 2), dtype=torch.float32, device=self.device) for _ in range(2)]
        acts_T = torch.empty((G, self.num_envs, 2), dtype=torch.float32, device=self.device) if policy == "episode" else None
        side2 = torch.cuda.Stream(device=self.device) if policy == "overlap" else None
        shards = int(shards)
        parts, part_streams = [], []
        if shards > 1:
            if policy == "overlap" or self._soa:
                raise ValueError("capture_steps: shards > 1 needs policy kernel|episode|fused and obs_layout='aos'")
            per = -(-self.num_envs // shards)
            per = -(-per // 256) * 256
            parts = [(a, min(per, self.num_envs - a)) for a in range(0, self.num_envs, per)]
            part_streams = [torch.cuda.Stream(device=self.device) for _ in parts]

        def body_sharded():
            main = torch.cuda.current_stream(self.device)
            if policy == "episode":
                self.step_idx = 0
                self.random_policy_steps(G, out=acts_T)
            fork = torch.cuda.Event()
            fork.record(main)
            for (first, n), st in zip(parts, part_streams):
                st.wait_event(fork)
                for g in range(G):
                    act = pol = None
                    if policy == "episode":
                        act = C.c_void_p(acts_T[g].data_ptr() + first * 8)
                    elif policy == "kernel":
                        pol = C.c_void_p(acts[0].data_ptr() + first * 8)
                    self._step_shard(first, n, act, pol, g, st)
                main.wait_stream(st)
            self.advance_step_base(G)
            self.step_idx = 0
            if self._actions_out is not None:
                self.last_action = self._actions_out

        def body():
            if parts:
                return body_sharded()
            self.step_idx = 0
            if policy == "overlap":
                main = torch.cuda.current_stream(self.device)
                self.random_policy(out=acts[0])
                for g in range(G):
                    if g + 1 < G:
                        side2.wait_stream(main)      # acts[(g+1)%2] was read by step g-1, which is on `main`
                        with torch.cuda.stream(side2):
                            self.random_policy(out=acts[(g + 1) % 2], lookahead=1)
                    self.step(acts[g % 2])
                    main.wait_stream(side2)          # step g+1 reads what side2 has just been asked to write
            elif policy == "episode":
                self.random_policy_steps(G, out=acts_T)
                for g in range(G):
                    self.step(acts_T[g])
            else:
                for _ in range(G):
                    if policy == "kernel":
                        self.step(self.random_policy(out=acts[0]))
                    else:
                        self.step(None)
            self.advance_step_base(G)
            self.step_idx = 0

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            body()  # warm-up (executes for real)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        self._graph_keepalive = (acts, acts_T, side2, part_streams)
        return graph

    def check_status(self):
        """Synchronising check of the device status word (bit0: an env's RK45 attempt guard tripped --
        the reference would have raised on a failed solver)."""
        s = int(self.status.item())
        if s:
            raise RuntimeError(f"mrsim device status 0x{s:x}: RK45 step-size control failed in at least one env")
        return s

    # ------------------------------------------------------------------ reference API odds and ends
    def render(self, mode="human"):
        return None  # GUI (turtle viewer, MR_viewer.py) is out of scope

    def close(self):
        return None

    def set_init_space(self, low, high):
        """MR_env.py:154-155."""
        self.cfg.init_low, self.cfg.init_high = tuple(float(x) for x in low), tuple(float(x) for x in high)
        self.init_space = make_box(self.cfg.init_low, self.cfg.init_high, seed=self.seed_value)
        self._refresh_params()

    def set_goal(self, init=None):
        """MR_env.py:157-162 -- a no-op returning init_goal."""
        return self.init_goal

    # ------------------------------------------------------------------ checkpoint
    _CFG_STATE = ("noise_var", "a0", "is_mismatched", "init_low", "init_high")  # what reset() / set_init_space change

    def state_dict(self):
        """Everything a resumed env needs to continue bit for bit: the per-env state, the outputs of the latest step /
        reset that a gym-style loop reads before its next action (obs, rew, done, final_*, status), the RNG position -- host
        step_idx plus, once capture_steps() / enable_device_step_base() moved the counter into HBM, the device word
        it lives in -- and the cfg fields reset() kwargs and set_init_space() may have changed."""
        sb = getattr(self, "_step_base", None)
        return {"pos": self.pos.clone(), "aux": self.aux.clone(), "ep_ret": self.ep_ret.clone(),
                "final_ret": self.final_ret.clone(), "final_len": self.final_len.clone(),
                "obs": self._obs.clone(), "final_obs": self._final_obs.clone(), "rew": self.rew.clone(),
                "done": self._done_u8.clone(), "status": self.status.clone(), "obs_layout": self.cfg.obs_layout,
                "step_idx": self.step_idx, "step_base": None if sb is None else int(sb.item()),
                "seed": self.seed_value, "env_id0": self.env_id0, "prev_mismatched": self._prev_mismatched,
                "cfg": {k: getattr(self.cfg, k) for k in self._CFG_STATE}}

    def load_state_dict(self, sd):
        self.pos.copy_(sd["pos"]); self.aux.copy_(sd["aux"]); self.ep_ret.copy_(sd["ep_ret"])
        if "final_ret" in sd:
            self.final_ret.copy_(sd["final_ret"]); self.final_len.copy_(sd["final_len"])
        if "obs" in sd and sd.get("obs_layout") == self.cfg.obs_layout:
            self._obs.copy_(sd["obs"]); self._final_obs.copy_(sd["final_obs"]); self.rew.copy_(sd["rew"])
            self._done_u8.copy_(sd["done"]); self.status.copy_(sd["status"])
        self.seed_value, self.env_id0 = int(sd["seed"]), int(sd["env_id0"])
        self._prev_mismatched = bool(sd["prev_mismatched"])
        import dataclasses
        self.cfg = dataclasses.replace(self.cfg, **sd.get("cfg", {}))   # the caller's MRConfig object is left alone
        self.init_space = make_box(self.cfg.init_low, self.cfg.init_high, seed=self.seed_value)
        # RNG position = step_base (device word, if this env keeps one) + step_idx; restore it in whichever form THIS
        # env uses, so a checkpoint taken behind a captured graph resumes on an eager env and vice versa
        total = int(sd["step_idx"]) + int(sd.get("step_base") or 0)
        if getattr(self, "_step_base", None) is not None:
            self._step_base.fill_(total)
            self.step_idx = 0
        else:
            self.step_idx = total
        self._refresh_params()
