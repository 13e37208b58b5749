"""MR_Env: single-environment drop-in for the reference's class (MR_env.py:21-229).

Same constructor, method names, argument meaning and defaults; numpy in / numpy out.  This is what a reference checkout
imports in place of its own MR_env.MR_Env: `utils.run_sim` (utils.py:46-54), the DDPG loop
(RL/MR_ddpg.py:270-278) and `RL/read_data.py:58-78` call reset() / step() once per Python loop iteration.

One env, one call at a time, from the host: the cost of a step is the round trip, not the arithmetic.  The env's state, its
action and every output of a step therefore live in ONE pinned, device-mapped host record (mrsim_host_alloc): the host writes
the action into it, mrsim_step (the same kernel the vectorised env launches, n = 1) reads and writes the record over the bus,
and one mrsim_stream_synchronize later the host reads {pos, obs, state_prime, rew, done, counter} straight out of it -- one
launch and one wait per step, no copy call and no device tensor on either side (round 4's facade made six host<->device
transfers per step).  It needs the GPU (there is no CPU path); for throughput use MRVecEnv / RolloutCollector.
"""
import ctypes as C

import numpy as np

from . import _lib, recorder, spaces
from .config import MRConfig
from .spaces import make_box


class _Sim:
    def __init__(self, env):
        self._e = env

    # readable AND writable, as the reference's plain attributes are (MR_env.py:179-183 assigns them; callers may too)
    def _cfg_attr(name, refresh):  # noqa: N805
        def get(s):
            return getattr(s._e.cfg, name)

        def put(s, value):
            setattr(s._e.cfg, name, type(getattr(s._e.cfg, name))(value))
            if refresh:
                s._e._refresh_params()
        return property(get, put)

    noise_var = _cfg_attr("noise_var", True)
    a0 = _cfg_attr("a0", True)
    is_mismatched = _cfg_attr("is_mismatched", True)
    state_prime = property(lambda s: s._e.state_prime)

    def get_state(self):
        return np.array(self._e.last_pos, dtype=np.float64)


class _HostRecord:
    """The pinned, device-mapped block of one env: MrsimState (pos, aux, ep_ret), the action, the outputs of mrsim_step /
    mrsim_reset and the reset's start position.  Every array starts at a 16-byte boundary (include/mrsim.h: alignment)."""
    FIELDS = (("pos", 0, np.float64, 2), ("aux", 16, np.float32, 4), ("ep_ret", 32, np.float32, 1), ("action", 48, np.float32, 2),
              ("obs", 64, np.float32, 5), ("rew", 96, np.float32, 1), ("done", 100, np.uint8, 1), ("state_prime", 112, np.float32, 2),
              ("status", 128, np.int32, 1), ("init_xy", 144, np.float64, 2), ("step_word", 160, np.int32, 1))
    SIZE = 192

    def __init__(self, L):
        self._L = L
        h, d = C.c_void_p(), C.c_void_p()
        _lib.check(L.mrsim_host_alloc(self.SIZE, C.byref(h), C.byref(d)), "mrsim_host_alloc")
        self.host, self.dev = h.value, d.value
        raw = (C.c_uint8 * self.SIZE).from_address(self.host)
        self._raw = raw
        for name, off, dt, cnt in self.FIELDS:
            setattr(self, name, np.frombuffer(raw, dtype=dt, count=cnt, offset=off))
        self.counter = np.frombuffer(raw, dtype=np.int32, count=1, offset=16 + 12)   # aux.w holds MR_Env.counter's bits
        self.aux[2] = 1.0

    def ptr(self, name):
        off = next(o for n, o, _, _ in self.FIELDS if n == name)
        return C.c_void_p(self.dev + off)

    def close(self):
        if self.host:
            for name, *_ in self.FIELDS:
                setattr(self, name, None)
            self.counter = self._raw = None
            self._L.mrsim_host_free(C.c_void_p(self.host))
            self.host = self.dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MR_Env:
    def __init__(self, type="continuous", action_dim=2, cfg=None, device="cuda", seed=0, env_id=0):
        import torch
        self.type = type
        self.action_dim = action_dim
        self._L = _lib.lib()  # raises ImportError if the HIP extension is not built
        if not torch.cuda.is_available() or self._L.mrsim_device_count() <= 0:
            raise RuntimeError("MR_Env needs a HIP device (MI355X); mr_rl_amd has no CPU fallback")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("MR_Env: device must be a cuda(HIP) device")
        self._dev_index = torch.cuda.current_device() if dev.index is None else dev.index
        self._torch = torch
        self.cfg = cfg if cfg is not None else MRConfig()
        self.seed_value = int(self.cfg.seed if seed is None else seed)
        self.env_id = int(env_id)
        self.step_idx = 0   # RNG step index: every reset() / step() consumes one (include/mrsim.h: seed / step_idx contract)
        with torch.cuda.device(self._dev_index):
            self._rec = _HostRecord(self._L)
        r = self._rec
        self._st = _lib.MrsimState(r.ptr("pos"), r.ptr("aux"), r.ptr("ep_ret"))
        self._io = _lib.MrsimStepIO(r.ptr("action"), None, None, r.ptr("obs"), r.ptr("rew"), r.ptr("done"), r.ptr("state_prime"),
                                    None, None, None, r.ptr("status"))
        self._io.done_word = r.ptr("step_word").value
        self._word_host = C.c_void_p(r.host + 160)
        self._word = 0
        self._prev_mismatched = False   # what Simulator.is_mismatched was before the latest reset (MR_env.py:181-183)
        self._params = None
        self._refresh_params()
        # spaces (MR_env.py:34-45)
        self.action_space = make_box(self.cfg.action_low, self.cfg.action_high)
        self.observation_space = make_box(self.cfg.obs_low, self.cfg.obs_high)
        self.init_space = make_box(self.cfg.init_low, self.cfg.init_high, seed=self.seed_value)
        # MR_env.py:43-45: never sampled by the reference (its only use, :158, is commented out); low > high as written there
        self.init_goal_space = spaces.Box([-31.0, -31.0], [-32.0, -32.0])
        self.goal_loc = None
        self.borders = [[-510, 510], [-510, -510], [510, -510], [510, 510]]  # MR_env.py:46-50 (drawing only)
        self.simulator = _Sim(self)
        self.test_performance = False
        self.last_pos = np.zeros(2)
        self.init_goal = np.zeros(2)
        self.last_action = np.zeros(self.action_dim)
        self.number_loop = 0
        self.counter = 0
        self.max_timesteps = self.cfg.max_timesteps
        self.min_dist2goal = self.cfg.min_dist2goal
        self.viewer = None
        self.MR_data = None
        self.name_experiment = None
        self.state_prime = None

    # ------------------------------------------------------------------ plumbing
    def _refresh_params(self):
        self._params = self.cfg.to_params()
        self._pp = C.byref(self._params)

    def _on_device(self):
        """context of the env's device when it is not the current one (one integer comparison otherwise)"""
        t = self._torch
        return t.cuda.device(self._dev_index) if t.cuda.current_device() != self._dev_index else _NULL_CTX

    def _wait(self, what, word=None):
        """word: the value the step kernel stores into the record's step_word after its last output (MrsimStepIO.done_word) -- the
        host polls the record itself (mrsim_host_wait_word: the kernel's completion signal arrives microseconds after its last
        store); after a generous timeout, and for every launch without a word, wait for the stream."""
        if word is None or self._L.mrsim_host_wait_word(self._word_host, word, 2_000_000) != _lib.OK:
            _lib.check(self._L.mrsim_stream_synchronize(None), what)
            if word is not None and int(self._rec.step_word[0]) != word:
                raise RuntimeError(f"{what}: the launch completed without storing its step word")
        if self._rec.status[0]:
            # SciPy raises from RK45.step() once the solver has failed (step size below the spacing of floats / NaN input):
            # "Attempt to step on a failed or finished solver."
            s = int(self._rec.status[0])
            self._rec.status[0] = 0
            raise RuntimeError(f"mrsim device status 0x{s:x}: RK45 step-size control failed (the reference's integrator would have failed)")

    def seed(self, seed=None):
        """old/MR_dqn_keras_rl.py:19 calls env.seed(n): reseeds the noise stream and the init_space sampler (the
        constructor's seed= argument does the same, so MR_Env(seed=s).reset() is reproducible without this call)."""
        if seed is not None:
            self.seed_value = int(seed)
            self.init_space.seed(self.seed_value)   # MR_Env.reset(init=None) samples from it (MR_env.py:172-173)
        return [self.seed_value]

    # ------------------------------------------------------------------ gym API
    def reset(self, init=None, noise_var=1, a0=1, is_mismatched=False):
        """MR_env.py:164-201 (without its two print() calls)."""
        if init is None:
            init = self.init_space.sample()  # float32, MR_env.py:173
        init = np.asarray(init, dtype=np.float64).reshape(2)
        self.cfg.noise_var, self.cfg.a0 = float(noise_var), float(a0)
        ctor_mis = self._prev_mismatched           # reset_start_pos builds the RK45 object BEFORE :183 sets is_mismatched
        self.cfg.is_mismatched = bool(is_mismatched)
        self._prev_mismatched = self.cfg.is_mismatched
        self._refresh_params()
        r = self._rec
        r.init_xy[:] = init
        with self._on_device():
            _lib.check(self._L.mrsim_reset(self._pp, 1, self.env_id, C.byref(self._st), None, r.ptr("init_xy"), None, r.ptr("obs"),
                                           int(ctor_mis), self.seed_value, self.step_idx, None), "mrsim_reset")
            self._wait("mrsim_reset")
        self.step_idx += 1
        # MR_env.py:182: a second draw from init_space at every reset; nothing reads it (the goal stays init_goal), but it
        # advances the space's sampler, so a loop of reset(init=None) visits every OTHER sample of the stream -- as the reference
        self.goal_loc = self.init_space.sample()
        self.last_pos = init
        self.counter = 0
        obs = r.obs.astype(np.float64)
        if self.MR_data is not None:                                   # MR_env.py:189-198
            if self.MR_data.iterations > 0:
                self.MR_data.save_experiment(self.name_experiment)
            self.MR_data.new_iter(np.array(init, dtype=np.float64), obs, np.zeros(len(self.last_action)), np.array([0]))
        return obs

    def step(self, action):
        """MR_env.py:70-98: returns (obs[5], rew, done, {}).  One launch + one wait; everything read below sits in the host record."""
        f_t, alpha_t = action[0], action[1]  # IndexError on a bad action shape, like the reference
        r = self._rec
        r.action[0] = f_t
        r.action[1] = alpha_t
        self._word = word = (self._word % 0x7FFFFFFF) + 1      # never 0, never the previous step's
        self._io.done_value = word
        with self._on_device():
            _lib.check(self._L.mrsim_step(self._pp, 1, self.env_id, C.byref(self._st), C.byref(self._io), self.seed_value,
                                          self.step_idx, None), "mrsim_step")
            self._wait("mrsim_step", word)
        self.step_idx += 1
        self.counter += 1
        self.last_pos = [float(r.pos[0]), float(r.pos[1])]
        self.last_action = np.array([f_t, alpha_t])
        self.state_prime = r.state_prime.astype(np.float64)
        rew = float(r.rew[0])
        rew = int(rew) if rew == int(rew) else rew
        obs, done = r.obs.astype(np.float64), bool(r.done[0])
        if self.MR_data is not None:
            # MR_env.py:145-147: `end` saves the experiment when an episode dies on the bounds or the step limit (the goal
            # branch does not); :94-95: every transition is recorded -- end() runs before new_transition in step()
            if done and self.MR_data.iterations > 0 and (self.counter > self.max_timesteps or
                                                         not self.observation_space.contains(obs.astype(np.float32))):
                self.MR_data.save_experiment(self.name_experiment)
            self.MR_data.new_transition(np.array(self.last_pos, dtype=np.float64), obs, self.last_action, rew)
        return obs, rew, done, dict()

    def render(self, mode="human"):
        return None

    def close(self):
        """MR_env.py:220 (the reference closes its viewer); releases the host record."""
        if getattr(self, "_rec", None) is not None:
            self._rec.close()
        return None

    def set_init_space(self, low, high):
        """MR_env.py:154-155."""
        self.cfg.init_low, self.cfg.init_high = tuple(float(x) for x in low), tuple(float(x) for x in high)
        self.init_space = make_box(self.cfg.init_low, self.cfg.init_high, seed=self.seed_value)
        self._refresh_params()

    def set_goal(self, init=None):
        return self.init_goal

    def set_save_experice(self, name="experiment_ssn_ddpg_10iter"):
        """MR_env.py:223-226: from now on reset() / step() record into an MRExperiment-layout recorder and save it to
        ./_experiments/<date-hour><name> at the reference's trigger points."""
        assert isinstance(name, str), "name must be a string"
        self.MR_data = recorder.ExperimentRecorder()
        self.name_experiment = name

    def set_test_performace(self):
        self.test_performance = True


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()
