"""MR_Env: single-environment drop-in for the reference's class (MR_env.py:21-229).

Same constructor, method names, argument meaning and defaults; numpy in / numpy out.
It is an MRVecEnv with N = 1: every step is one launch of the HIP kernel (so it needs
the GPU and is meant for plumbing / parity, not speed -- use MRVecEnv for throughput).
"""
import numpy as np

from . import recorder, spaces
from .config import MRConfig
from .vec_env import MRVecEnv


class _Sim:
    def __init__(self, env):
        self._e = env

    # readable AND writable, as the reference's plain attributes are (MR_env.py:179-183 assigns them; callers may too)
    def _cfg_attr(name, refresh):  # noqa: N805
        def get(s):
            return getattr(s._e._v.cfg, name)

        def put(s, value):
            setattr(s._e._v.cfg, name, type(getattr(s._e._v.cfg, name))(value))
            if refresh:
                s._e._v._refresh_params()
        return property(get, put)

    noise_var = _cfg_attr("noise_var", True)
    a0 = _cfg_attr("a0", True)
    is_mismatched = _cfg_attr("is_mismatched", True)
    state_prime = property(lambda s: s._e.state_prime)

    def get_state(self):
        return np.array(self._e.last_pos, dtype=np.float64)


class MR_Env:
    def __init__(self, type="continuous", action_dim=2, cfg=None, device="cuda", seed=0, env_id=0):
        self.type = type
        self.action_dim = action_dim
        self._v = MRVecEnv(1, cfg=cfg if cfg is not None else MRConfig(), device=device, seed=seed, env_id0=env_id,
                           track_state_prime=True)
        v = self._v
        self.action_space, self.observation_space, self.init_space = v.action_space, v.observation_space, v.init_space
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
        self.max_timesteps = v.cfg.max_timesteps
        self.min_dist2goal = v.cfg.min_dist2goal
        self.viewer = None
        self.MR_data = None
        self.name_experiment = None
        self.state_prime = None

    def seed(self, seed=None):
        """old/MR_dqn_keras_rl.py:19 calls env.seed(n): reseeds the noise stream and the init_space sampler (the
        constructor's seed= argument does the same, so MR_Env(seed=s).reset() is reproducible without this call)."""
        out = self._v.seed(seed)
        self.init_space = self._v.init_space
        return out

    def reset(self, init=None, noise_var=1, a0=1, is_mismatched=False):
        """MR_env.py:164-201 (without its two print() calls)."""
        if init is None:
            init = self.init_space.sample()  # float32, MR_env.py:173
        init = np.asarray(init, dtype=np.float64).reshape(2)
        obs = self._v.reset(init=init[None, :], noise_var=noise_var, a0=a0, is_mismatched=is_mismatched)
        # MR_env.py:182: a second draw from init_space at every reset; nothing reads it (the goal stays init_goal), but it
        # advances the space's sampler, so a loop of reset(init=None) visits every OTHER sample of the stream -- as the reference
        self.goal_loc = self.init_space.sample()
        self.last_pos = init
        self.counter = 0
        obs = obs[0].double().cpu().numpy()
        if self.MR_data is not None:                                   # MR_env.py:189-198
            if self.MR_data.iterations > 0:
                self.MR_data.save_experiment(self.name_experiment)
            self.MR_data.new_iter(np.array(init, dtype=np.float64), obs, np.zeros(len(self.last_action)), np.array([0]))
        return obs

    def step(self, action):
        """MR_env.py:70-98: returns (obs[5], rew, done, {})."""
        f_t, alpha_t = action[0], action[1]  # IndexError on a bad action shape, like the reference
        obs, rew, done, _ = self._v.step(np.array([[f_t, alpha_t]], dtype=np.float32))
        self.counter += 1
        pos = self._v.pos[0].cpu().numpy()
        self.last_pos = [float(pos[0]), float(pos[1])]
        self.last_action = np.array([f_t, alpha_t])
        self.state_prime = self._v.state_prime[0].double().cpu().numpy()
        r = float(rew[0].item())
        r = int(r) if r == int(r) else r
        obs, done = obs[0].double().cpu().numpy(), bool(done[0].item())
        if self.MR_data is not None:
            # MR_env.py:145-147: `end` saves the experiment when an episode dies on the bounds or the step limit (the goal
            # branch does not); :94-95: every transition is recorded -- end() runs before new_transition in step()
            if done and self.MR_data.iterations > 0 and (self.counter > self.max_timesteps or
                                                         not self.observation_space.contains(obs.astype(np.float32))):
                self.MR_data.save_experiment(self.name_experiment)
            self.MR_data.new_transition(np.array(self.last_pos, dtype=np.float64), obs, self.last_action, r)
        return obs, r, done, dict()

    def render(self, mode="human"):
        return None

    def close(self):
        return None

    def set_init_space(self, low, high):
        self._v.set_init_space(low, high)
        self.init_space = self._v.init_space

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
