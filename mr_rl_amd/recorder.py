"""Trajectory recorder compatible with the reference's `MRExperiment`.

`MR_data.MRExperiment` (MR_data.py:9-85) grows per-episode arrays with np.vstack on every step (O(T^2)) and
pickles its `__dict__`.  Here a whole [T, N, ...] rollout that is already resident on the GPU is cut into
episodes once, on the host, and written in the same dictionary layout, so the reference's
`MRExperiment.load_from_experiment` (`pickle.load` + `__dict__.update`, MR_data.py:76-85) and its plotting /
settling-time helpers read new runs unchanged:

  iterations            index of the last episode (starts at -1)                        MR_data.py:15
  states[it]            [steps+1, 2]  row 0 = the reset state                           :39,54
  observations[it]      [steps+1, 5]  row 0 = the reset observation                     :40,55
  actions[it]           [steps+1, 2]  row 0 = zeros  (MR_env.py:196)                    :41,56
  rewards[it]           [steps+1, 1]  row 0 = [0]    (MR_env.py:197)                    :42,57
  steps[it]             number of transitions                                            :38,53
  info, viewer, scream, obs_states_str, time_step                                        :20-24
"""
import pickle

import numpy as np


def episodes_from_rollout(obs0, traj, obs, actions, rew, done):
    """Cut ONE env's rollout into MRExperiment episodes.

    obs0 [5]: observation returned by reset() before the rollout; traj [T,2] fp64 positions after each
    step (before any auto-reset); obs [T,5] returned observations (the reset observation on a done step
    when auto_reset is on); actions [T,2]; rew [T]; done [T].
    """
    traj, obs, actions = np.asarray(traj, np.float64), np.asarray(obs, np.float64), np.asarray(actions, np.float64)
    rew, done = np.asarray(rew, np.float64), np.asarray(done, bool)
    T = len(rew)
    d = dict(iterations=-1, states={}, observations={}, actions={}, rewards={}, steps={}, info=None, viewer=None,
             scream=None, obs_states_str={}, time_step=10)
    start, reset_obs = 0, np.asarray(obs0, np.float64)
    for t in range(T + 1):
        if t == T or done[t]:
            end = t + 1 if t < T else T
            if end == start:
                break
            it = d["iterations"] = d["iterations"] + 1
            n = end - start
            step_obs = obs[start:end].copy()
            goal = reset_obs[2:4]
            if t < T and done[t]:
                # the returned obs of the done step is the NEXT episode's reset obs (auto-reset): rebuild the
                # terminal observation from the recorded position
                p = traj[t]
                step_obs[-1] = [p[0], p[1], goal[0], goal[1], np.hypot(goal[0] - p[0], goal[1] - p[1])]
            d["states"][it] = np.vstack([reset_obs[:2], traj[start:end]])
            d["observations"][it] = np.vstack([reset_obs, step_obs])
            d["actions"][it] = np.vstack([np.zeros(2), actions[start:end]])
            d["rewards"][it] = np.vstack([np.array([0.0]), rew[start:end, None]])
            d["steps"][it] = n
            if t < T:
                reset_obs = obs[t]
            start = end
    return d


def record_rollout(env, T, env_index=0, actions=None, shared_actions=False):
    """Run a fused T-step rollout on `env` and return the MRExperiment-layout dict of one env."""
    obs0 = env.obs[env_index].double().cpu().numpy()
    out = env.rollout(T, actions=actions, shared_actions=shared_actions,
                      want=("traj", "obs", "rew", "done", "actions"))
    g = lambda k: out[k][:, env_index].cpu().numpy()  # noqa: E731
    return episodes_from_rollout(obs0, g("traj"), g("obs"), g("actions"), g("rew"), g("done"))


def save_experiment(d, path):
    """Same bytes layout as MRExperiment.save_experiment (pickle protocol 2 of the instance dict)."""
    with open(path, "wb") as f:
        pickle.dump(d, f, 2)


def load_experiment(path):
    """Only for files written by save_experiment above (never for pickles shipped with the reference)."""
    with open(path, "rb") as f:
        return pickle.load(f)
