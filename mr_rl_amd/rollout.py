"""Batched `run_sim` and the reference's action profiles.

`utils.run_sim(actions, init_pos, noise_var, a0, is_mismatched)` (utils.py:43-61) is the reference's
canonical open-loop driver: fresh env, reset, step through an action table ignoring `done`, return
`X, Y, alpha, time, freq` with `time = linspace(0, (T-1)/30, T)`.  Here the whole table is ONE launch of
the fused rollout kernel for `num_envs` independent noise realisations.

The action tables reproduce main.py:14-50 / main_2d.py:137-171 (idle, circles, the alpha-ramp test
profile) plus the figure-eight of SURVEY 8(d).  They are plain numpy [T,3] arrays {freq, alpha, time},
exactly the layout the reference's scripts build.
"""
import numpy as np

from .config import MRConfig
from .vec_env import MRVecEnv

DT = 0.030  # main.py:11 "assume a timestep of 30 ms"


def actions_idle(time_steps=100, dt=DT):
    """main.py:15-17: do nothing for time_steps/30 s."""
    a = np.zeros((time_steps, 3))
    a[:, 2] = np.arange(time_steps) * dt
    return a


def actions_circle(time_steps=1800, cycles=3, freq=4.0, dt=DT):
    """main.py:20-33: `cycles` circles at constant frequency, alpha = linspace(-pi, pi) per circle."""
    steps = int(time_steps / cycles)
    one = np.zeros((steps, 3))
    one[:, 0] = freq
    one[:, 1] = np.linspace(-np.pi, np.pi, steps)
    a = np.vstack([one] * cycles)
    a[:, 2] = np.arange(len(a)) * dt
    return a


def actions_circle_fm(time_steps=300, cycles=3, f_lo=0.1, f_hi=5.0, dt=DT):
    """main_2d.py:137-160: the 2-D learning set -- `cycles` circles (alpha = linspace(-pi, pi) per circle) whose frequency is
    modulated along the run, f = (cos(t / 5) + 1) / 2 * (f_hi - f_lo) + f_lo with t = linspace(0, time_steps, time_steps)
    (the script's 4.9 and 0.1)."""
    steps = int(time_steps / cycles)
    one = np.zeros((steps, 3))
    one[:, 1] = np.linspace(-np.pi, np.pi, steps)
    a = np.vstack([one] * cycles)
    t = np.linspace(0, len(a), len(a))
    a[:, 0] = (np.cos(t / 5) + 1) / 2 * (f_hi - f_lo) + f_lo
    a[:, 2] = np.arange(len(a)) * dt
    return a


def actions_ramp(freq=4.0, dt=DT):
    """main.py:39-50: the 1000-step alpha-ramp test profile."""
    T = 1000
    a = np.zeros((T, 3))
    a[0:200, 1] = np.linspace(0, np.pi / 2, 200)
    a[200:400, 1] = np.linspace(np.pi / 2, -np.pi / 2, 200)
    a[400:600, 1] = np.linspace(-np.pi / 2, 0, 200)
    a[600:800, 1] = np.linspace(0, np.pi / 8, 200)
    a[800:, 1] = np.linspace(np.pi / 8, -np.pi, 200)
    a[:, 0] = freq
    a[:, 2] = np.arange(T) * dt
    return a


def actions_figure8(T=1000, speed=4.0, dt=DT):
    """v = (cos th, cos 2th), f = speed*|v|, alpha = atan2(v_y, v_x)."""
    th = 2 * np.pi * np.arange(T) / T
    vx, vy = np.cos(th), np.cos(2 * th)
    a = np.zeros((T, 3))
    a[:, 0] = speed * np.hypot(vx, vy)
    a[:, 1] = np.arctan2(vy, vx)
    a[:, 2] = np.arange(T) * dt
    return a


def goal_table_from_actions(actions, init=(0.0, 0.0), a0=1.0, dt=DT):
    """Noise-free path of an action table under the nominal law (x += dt*a0*f*cos a, ...): a [1][T+1][2]
    reference-trajectory table for `MRVecEnv(goal_table=...)` (row t = where the robot should be after t steps)."""
    a = np.asarray(actions, dtype=np.float64)
    v = a0 * a[:, 0:1] * np.stack([np.cos(a[:, 1]), np.sin(a[:, 1])], axis=1)
    p = np.vstack([np.asarray(init, dtype=np.float64)[None, :], np.asarray(init)[None, :] + np.cumsum(dt * v, axis=0)])
    return p[None, :, :].astype(np.float32)


def run_sim(actions, init_pos=None, noise_var=1, a0=1, is_mismatched=False, num_envs=1, device="cuda", seed=0,
            return_state_prime=False, cfg=None):
    """utils.run_sim for `num_envs` independent environments in one fused launch.

    actions: [T,2] or [T,3] ({freq, alpha[, time]}; column 2 is ignored like the reference does).
    init_pos: [2] (shared), [num_envs,2], or None = init_space.sample() per env.
    Returns X, Y, alpha, time, freq as numpy arrays; X, Y are [T] when num_envs == 1 (the reference's
    shapes) and [T, num_envs] otherwise.  `done` is ignored (no reset), exactly like utils.py:51-54.
    """
    import torch
    actions = np.asarray(actions, dtype=np.float64)
    if actions.ndim != 2 or actions.shape[1] < 2:
        raise IndexError("actions must be [T, >=2] {freq, alpha}")
    T = len(actions)
    import dataclasses
    c = dataclasses.replace(cfg) if cfg is not None else MRConfig()  # the caller's cfg is not modified
    c.auto_reset = False
    env = MRVecEnv(num_envs, cfg=c, device=device, seed=seed)
    init = None
    if init_pos is not None:
        init = np.asarray(init_pos, dtype=np.float64)
        init = np.tile(init[None, :], (num_envs, 1)) if init.ndim == 1 else init
    env.reset(init=init, noise_var=noise_var, a0=a0, is_mismatched=is_mismatched)
    want = ("traj", "state_prime") if return_state_prime else ("traj",)
    # the table goes in as float64 (the reference's main.py tables are float64 linspace tables): no float32 rounding
    out = env.rollout(T, actions=torch.as_tensor(np.ascontiguousarray(actions[:, :2])), shared_actions=True, want=want)
    env.check_status()
    traj = out["traj"].cpu().numpy()
    X, Y = traj[:, :, 0], traj[:, :, 1]
    if num_envs == 1:
        X, Y = X[:, 0], Y[:, 0]
    alpha, freq = actions[:, 1], actions[:, 0]
    time = np.linspace(0, (T - 1) / 30.0, T)  # utils.py:59
    if return_state_prime:
        sp = out["state_prime"].cpu().numpy()
        return X, Y, alpha, time, freq, (sp[:, 0, :] if num_envs == 1 else sp)
    return X, Y, alpha, time, freq


def estimate_velocity(traj, time, n_filter=14):
    """Batched `LearningModule` velocity pipeline (Learning_module.py:46-59,72-93) on the GPU:
    uniform_filter1d(N) -> np.gradient(., time) -> uniform_filter1d(N/2), plus the drift D = mean(v[N:-N]).
    traj: [T,n,2] fp64 device tensor (MRVecEnv.rollout's "traj"); time: [T].  Returns (v [T,n,2], D [n,2])."""
    import ctypes as C
    import torch
    from . import _lib
    traj = traj.contiguous()
    assert traj.dtype == torch.float64 and traj.dim() == 3 and traj.shape[2] == 2 and traj.is_cuda
    T, n = int(traj.shape[0]), int(traj.shape[1])
    t = torch.as_tensor(time, dtype=torch.float64, device=traj.device).contiguous()
    assert t.shape == (T,)
    v = torch.empty_like(traj); scratch = torch.empty_like(traj)
    drift = torch.empty((n, 2), dtype=torch.float64, device=traj.device)
    P = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    rc = _lib.lib().mrsim_velocity(n, T, int(n_filter), P(traj), P(t), P(v), P(scratch), P(drift),
                                   C.c_void_p(torch.cuda.current_stream(traj.device).cuda_stream))
    _lib.check(rc, "mrsim_velocity")
    return v, drift


def estimate_a0(v, drift, freq, n_filter=14):
    """a0 = median(speed / freq) over v[N:-N] with the drift removed (Learning_module.py:96,112-123)."""
    import torch
    sp = torch.sqrt((v[..., 0] - drift[None, :, 0]) ** 2 + (v[..., 1] - drift[None, :, 1]) ** 2)
    return torch.median(sp[n_filter:-n_filter] / freq, dim=0).values
