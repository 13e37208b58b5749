"""Shared helpers for the parity tests (oracle side is test infrastructure)."""
import os

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    cases = {}
    for k in g.files:
        name, field = k.split("/")
        cases.setdefault(name, {})[field] = g[k]
    return cases


def orc_params_from_cfg(cfg, goal_K=1, goal_T=1):
    """OrcParams equivalent of an mr_rl_amd.MRConfig."""
    from mr_rl_amd.config import INTEGRATORS, REWARD_MODES
    return O.default_params(
        time_span=cfg.time_span, rtol=cfg.time_span / cfg.number_iterations, atol=cfg.atol, a0=cfg.a0,
        sigma=cfg.noise_var, min_dist2goal=cfg.min_dist2goal, obs_low=cfg.obs_low, obs_high=cfg.obs_high,
        init_low=cfg.init_low, init_high=cfg.init_high, mismatched=int(cfg.is_mismatched),
        integrator=INTEGRATORS[cfg.integrator], substeps=cfg.substeps, reward_mode=REWARD_MODES[cfg.reward_mode],
        max_timesteps=cfg.max_timesteps, auto_reset=int(cfg.auto_reset), goal_K=goal_K, goal_T=goal_T,
        auto_reset_fresh_env=int(cfg.auto_reset_env == "fresh"),
        noise_law=int(getattr(cfg, "noise_law", "per_stage") == "collapsed"))


def actions_figure8(T=1000):
    th = 2 * np.pi * np.arange(T) / T
    vx, vy = np.cos(th), np.cos(2 * th)
    a = np.zeros((T, 2), dtype=np.float32)
    a[:, 0] = 4.0 * np.hypot(vx, vy)
    a[:, 1] = np.arctan2(vy, vx)
    return a


def actions_ramp(T=1000, freq=4.0):
    a = np.zeros((1000, 2), dtype=np.float32)
    a[0:200, 1] = np.linspace(0, np.pi / 2, 200)
    a[200:400, 1] = np.linspace(np.pi / 2, -np.pi / 2, 200)
    a[400:600, 1] = np.linspace(-np.pi / 2, 0, 200)
    a[600:800, 1] = np.linspace(0, np.pi / 8, 200)
    a[800:, 1] = np.linspace(np.pi / 8, -np.pi, 200)
    a[:, 0] = freq
    return a[:T]


def random_actor(seed=0, bn_stats=True, out_scale=None):
    """An mr_rl_amd.ddpg.Actor (the PyTorch twin of RL/MR_ddpg.py's ActorNetwork) in eval mode with random weights; bn_stats:
    non-trivial running statistics and affine terms, as after training; out_scale: widen the U[-3e-3, 3e-3] output layer
    so that tanh leaves its linear range."""
    import torch
    from mr_rl_amd.ddpg import Actor
    g = torch.Generator().manual_seed(seed)
    m = Actor()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (1.0 / max(1, p.shape[-1])) ** 0.5)
        if out_scale is None:
            m.out.weight.copy_((torch.rand(m.out.weight.shape, generator=g) * 2 - 1) * 3e-3)
            m.out.bias.copy_((torch.rand(m.out.bias.shape, generator=g) * 2 - 1) * 3e-3)
        else:
            m.out.weight.mul_(out_scale)
        if bn_stats:
            for bn in (m.bn1, m.bn2):
                bn.running_mean.copy_(torch.randn(64, generator=g) * 0.3)
                bn.running_var.copy_(torch.rand(64, generator=g) * 2 + 0.25)
                bn.weight.copy_(torch.rand(64, generator=g) + 0.5)
                bn.bias.copy_(torch.randn(64, generator=g) * 0.2)
    return m.eval()
