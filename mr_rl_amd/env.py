"""MR_Env: single-environment drop-in for the reference's class (MR_env.py:21-229).

Same constructor, method names, argument meaning and defaults; numpy in / numpy out.
It is an MRVecEnv with N = 1: every step is one launch of the HIP kernel (so it needs
the GPU and is meant for plumbing / parity, not speed -- use MRVecEnv for throughput).
"""
import numpy as np

from .config import MRConfig
from .vec_env import MRVecEnv


class _Sim:
    def __init__(self, env):
        self._e = env

    noise_var = property(lambda s: s._e._v.cfg.noise_var)
    a0 = property(lambda s: s._e._v.cfg.a0)
    is_mismatched = property(lambda s: s._e._v.cfg.is_mismatched)
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
        self.init_space.seed(seed) if hasattr(self.init_space, "seed") else None
        return self._v.seed(seed)

    def reset(self, init=None, noise_var=1, a0=1, is_mismatched=False):
        """MR_env.py:164-201 (without its two print() calls)."""
        if init is None:
            init = self.init_space.sample()  # float32, MR_env.py:173
        init = np.asarray(init, dtype=np.float64).reshape(2)
        obs = self._v.reset(init=init[None, :], noise_var=noise_var, a0=a0, is_mismatched=is_mismatched)
        self.last_pos = init
        self.counter = 0
        return obs[0].double().cpu().numpy()

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
        return obs[0].double().cpu().numpy(), (int(r) if r == int(r) else r), bool(done[0].item()), dict()

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
        assert isinstance(name, str), "name must be a string"
        self.name_experiment = name  # the MRExperiment recorder (MR_data.py) is out of scope

    def set_test_performace(self):
        self.test_performance = True
