"""MRConfig: every tunable of the reference path in one place."""
import dataclasses
import math
from typing import Tuple

from . import _lib

INTEGRATORS = {"reference": _lib.INT_RK45, "rk45": _lib.INT_RK45, "euler": _lib.INT_EULER, "rk4": _lib.INT_RK4}
REWARD_MODES = {"constant10": _lib.REW_CONSTANT10, "goal": _lib.REW_GOAL}
OBS_LAYOUTS = {"aos": _lib.OBS_AOS, "soa": _lib.OBS_SOA}
NOISE_MATH = {"fast": _lib.NOISE_FAST, "spec": _lib.NOISE_SPEC}
NOISE_LAWS = {"per_stage": _lib.LAW_PER_STAGE, "collapsed": _lib.LAW_COLLAPSED}


@dataclasses.dataclass
class MRConfig:
    # Simulator (MR_simulator.py:12-13,90-91)
    time_span: float = 0.030
    number_iterations: int = 100          # rtol = time_span / number_iterations
    atol: float = 1e-4
    # MR_Env.reset kwargs (MR_env.py:164-170)
    noise_var: float = 1.0
    a0: float = 1.0
    is_mismatched: bool = False
    # MR_Env constants (MR_env.py:34-45,62-63)
    max_timesteps: int = 50
    min_dist2goal: float = 30.0
    obs_low: Tuple[float, ...] = (-5000.0, -5000.0, -5000.0, -5000.0, 0.0)
    obs_high: Tuple[float, ...] = (5000.0, 5000.0, 5000.0, 5000.0, 80000.0)
    init_low: Tuple[float, float] = (100.0, 100.0)
    init_high: Tuple[float, float] = (120.0, 120.0)
    action_low: Tuple[float, float] = (0.0, 0.0)                 # action_space (never enforced, MR_env.py:34-36)
    action_high: Tuple[float, float] = (20.0, 2 * math.pi)
    # on-device random policy range = DDPG actor range (RL/MR_ddpg.py:136-137,345)
    policy_low: Tuple[float, float] = (-20.0, -2 * math.pi)
    policy_high: Tuple[float, float] = (20.0, 2 * math.pi)
    # build extensions
    integrator: str = "reference"         # reference (SciPy RK45 semantics) | euler | rk4
    substeps: int = 1
    reward_mode: str = "constant10"       # constant10 (MR_env.py:89) | goal (calculate_reward, :118-134)
    auto_reset: bool = False
    auto_reset_env: str = "reused"        # what an auto-reset stands for: "reused" = the same MR_Env object reset at the top of
                                          # every episode (RL/MR_ddpg.py:270): the RK45 constructor inside reset runs under the
                                          # law the previous episode left behind (MR_env.py:181-183); "fresh" = a new MR_Env
                                          # per episode (utils.run_sim): nominal-law constructor.  Differs only if is_mismatched.
    obs_layout: str = "aos"               # storage of obs: [N,5] rows or [5,N] planes (returned view is [N,5])
    noise_math: str = "fast"              # Box-Muller on hardware transcendentals | "spec": bit-identical to the oracle
    noise_law: str = "collapsed"          # "collapsed" (the library default, = mrsim_default_params): the weighted stage sums
                                          # of an RK45 attempt drawn directly from their joint Gaussian -- the law of
                                          # MR_simulator.py:73-83's per-evaluation noise with fewer draws, pinned against the
                                          # reference's own samples (include/mrsim.h); "per_stage": a fresh normal at every
                                          # RHS evaluation in the reference's order (the layout of the oracle's tape replays)
    rollout_carry: str = "f32"            # fused rollout: "f32" = carried RK45 state rounded per step (bit-identical to
                                          # step()); "f64" = kept in fp64 registers for the whole launch
    seed: int = 0

    def to_params(self, goal_K=1, goal_T=1):
        p = _lib.default_params()
        p.time_span = self.time_span
        p.rtol = self.time_span / self.number_iterations
        p.atol = self.atol
        p.a0 = self.a0
        p.sigma = self.noise_var
        p.min_dist2goal = self.min_dist2goal
        for i in range(5):
            p.obs_low[i] = self.obs_low[i]
            p.obs_high[i] = self.obs_high[i]
        for i in range(2):
            p.init_low[i] = self.init_low[i]
            p.init_high[i] = self.init_high[i]
            p.act_low[i] = self.policy_low[i]
            p.act_high[i] = self.policy_high[i]
        p.mismatched = int(bool(self.is_mismatched))
        p.integrator = INTEGRATORS[self.integrator]
        p.substeps = int(self.substeps)
        p.reward_mode = REWARD_MODES[self.reward_mode]
        p.max_timesteps = int(self.max_timesteps)
        p.auto_reset = int(bool(self.auto_reset))
        if self.auto_reset_env not in ("reused", "fresh"):
            raise ValueError("auto_reset_env must be 'reused' or 'fresh'")
        p.auto_reset_fresh_env = int(self.auto_reset_env == "fresh")
        p.goal_K, p.goal_T = int(goal_K), int(goal_T)
        p.obs_layout = OBS_LAYOUTS[self.obs_layout]
        p.noise_math = NOISE_MATH[self.noise_math]
        p.noise_law = NOISE_LAWS[self.noise_law]
        return p
