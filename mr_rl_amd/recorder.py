"""Trajectory recorder compatible with the reference's `MRExperiment`.

`MR_data.MRExperiment` (MR_data.py:9-85) grows per-episode arrays with np.vstack on every step (O(T^2)) and
pickles its `__dict__`.  Here a whole [T, N, ...] rollout that is already resident on the GPU is cut into
episodes once, on the host, and written in the same dictionary layout, so the reference's
`MRExperiment.load_from_experiment` (`pickle.load` + `__dict__.update`, MR_data.py:76-85) and its plotting /
settling-time helpers read new runs unchanged:

  iterations            index of the last episode (starts at -1)                        MR_data.py:15
  states[it]            [steps+1, 2]  float64, row 0 = the reset state                  :39,54
  observations[it]      [steps+1, 5]  float64, row 0 = the reset observation            :40,55
  actions[it]           [steps+1, 2]  float64, row 0 = zeros  (MR_env.py:196)           :41,56
  rewards[it]           [steps+1, 1]  row 0 = [0]  (MR_env.py:197); int64 under the     :42,57
                        reference's constant reward (`rew = 10`, MR_env.py:89), float64 under calculate_reward
  steps[it]             number of transitions                                            :38,53
  info, viewer, scream, obs_states_str, time_step                                        :20-24

Pinned against the reference's own recorder by tests/golden/ref_experiment.npz (tests/test_gpu_parity.py).
"""
import pickle

import numpy as np


def _empty():
    return dict(iterations=-1, states={}, observations={}, actions={}, rewards={}, steps={}, info=None, viewer=None,
                scream=None, obs_states_str={}, time_step=10)


class ExperimentRecorder:
    """The recording half of MR_data.MRExperiment (MR_data.py:9-74) behind MR_Env.set_save_experice: `new_iter` at every
    reset, `new_transition` at every step, `save_experiment` = pickle (protocol 2) of the same dictionary layout into
    ./_experiments/<YYYY-mm-dd-HH><name>.  Rows are appended to lists and stacked when read (the reference re-stacks every
    array on every step); `to_dict()` / the attributes give exactly MRExperiment.__dict__'s content."""

    def __init__(self, info=None):
        self.iterations, self.info = -1, info
        self._rows = {}   # it -> {"states": [...], "observations": [...], "actions": [...], "rewards": [...]}

    def new_iter(self, s0, obs0, a0, r0):
        self.iterations += 1
        self._rows[self.iterations] = {"states": [np.asarray(s0)], "observations": [np.asarray(obs0)],
                                       "actions": [np.asarray(a0)], "rewards": [np.asarray(r0)]}

    def new_transition(self, s, obs, a, r):
        row = self._rows[self.iterations]
        for key, v in (("states", s), ("observations", obs), ("actions", a), ("rewards", r)):
            row[key].append(np.asarray(v))

    def _stack(self, key):
        # one row: the array itself (MRExperiment.new_iter stores s0 unstacked); more: np.vstack, as new_transition does
        return {it: (rows[key][0] if len(rows[key]) == 1 else np.vstack(rows[key])) for it, rows in self._rows.items()}

    states = property(lambda self: self._stack("states"))
    observations = property(lambda self: self._stack("observations"))
    actions = property(lambda self: self._stack("actions"))
    rewards = property(lambda self: self._stack("rewards"))
    steps = property(lambda self: {it: len(rows["states"]) - 1 for it, rows in self._rows.items()})

    def to_dict(self):
        d = _empty()
        d.update(iterations=self.iterations, states=self.states, observations=self.observations, actions=self.actions,
                 rewards=self.rewards, steps=self.steps, info=self.info)
        return d

    def save_experiment(self, descr="_experiment"):
        import datetime
        import os
        os.makedirs("_experiments", exist_ok=True)
        path = os.path.join("_experiments", datetime.datetime.now().strftime("%Y-%m-%d-%H") + descr)
        save_experiment(self.to_dict(), path)
        return path


def _rewards_column(rew, integer):
    r = np.asarray(rew, np.float64)
    col = np.vstack([np.array([0.0]), r[:, None]]) if len(r) else np.array([[0.0]])
    return col.astype(np.int64) if integer else col


def episodes_from_rollout(obs0, traj, obs, actions, rew, done, goals=None, integer_rewards=None, auto_reset=None):
    """Cut ONE env's rollout into MRExperiment episodes.

    obs0 [5]: observation returned by reset() before the rollout; traj [T,2] fp64 positions after each
    step (before any auto-reset); obs [T,5] returned observations (the reset observation on a done step
    when auto_reset is on); actions [T,2]; rew [T]; done [T].
    goals: optional [T,2] goal of every step (an env with a goal table: the goal varies along the episode);
    default = the goal the reset observation shows (MR_Env.init_goal).  Only used to rebuild the terminal
    observation that an auto-reset step replaces by the next episode's reset observation.
    integer_rewards: None = int64 column when every reward is an integer (the reference's `rew = 10`).
    auto_reset: whether the rollout ran with same-step auto-reset (then the returned observation of a done step is the
    NEXT episode's reset observation and the terminal one is rebuilt); None = tell from the data.
    """
    traj, obs, actions = np.asarray(traj, np.float64), np.asarray(obs, np.float64), np.asarray(actions, np.float64)
    rew, done = np.asarray(rew, np.float64), np.asarray(done, bool)
    T = len(rew)
    if integer_rewards is None:
        integer_rewards = bool(np.all(rew == np.round(rew)))
    d = _empty()
    start, reset_obs = 0, np.asarray(obs0, np.float64)
    for t in range(T + 1):
        if t == T or done[t]:
            end = t + 1 if t < T else T
            if end == start:
                break
            it = d["iterations"] = d["iterations"] + 1
            n = end - start
            step_obs = obs[start:end].copy()
            if t < T and done[t]:
                p = traj[t]
                replaced = auto_reset
                if replaced is None:
                    replaced = not (step_obs[-1, 0] == np.float32(p[0]) and step_obs[-1, 1] == np.float32(p[1]))
                if replaced:
                    # the returned obs of the done step is the NEXT episode's reset obs (auto-reset): rebuild the
                    # terminal observation from the recorded position and the goal of THAT step
                    goal = reset_obs[2:4] if goals is None else np.asarray(goals[t], np.float64)
                    step_obs[-1] = [p[0], p[1], goal[0], goal[1], np.hypot(goal[0] - p[0], goal[1] - p[1])]
            d["states"][it] = np.vstack([reset_obs[:2], traj[start:end]])
            d["observations"][it] = np.vstack([reset_obs, step_obs])
            d["actions"][it] = np.vstack([np.zeros(2), actions[start:end]])
            d["rewards"][it] = _rewards_column(rew[start:end], integer_rewards)
            d["steps"][it] = n
            if t < T:
                reset_obs = obs[t]
            start = end
    return d


def _goals_of(env, T, env_index, counter0):
    """[T,2] goal of each step of env `env_index` when the env has a goal table ([K][Tg][2], row = counter after the
    step, clamped), else None.  Valid for one episode (no reset inside the T steps)."""
    if env.goal_table is None:
        return None
    tab = env.goal_table.cpu().numpy()
    K, Tg = tab.shape[0], tab.shape[1]
    k = (env.env_id0 + env_index) % K if K > 1 else 0
    rows = np.clip(counter0 + 1 + np.arange(T), 0, Tg - 1)
    return tab[k, rows]


def record_rollout(env, T, env_index=0, actions=None, shared_actions=False):
    """Run a fused T-step rollout on `env` and return the MRExperiment-layout dict of one env."""
    obs0 = env.obs[env_index].double().cpu().numpy()
    p0 = env.pos[env_index].cpu().numpy()  # the reset state in fp64 (the observation is float32)
    if obs0[0] == np.float32(p0[0]) and obs0[1] == np.float32(p0[1]):
        obs0[:2] = p0
        obs0[4] = np.hypot(obs0[2] - p0[0], obs0[3] - p0[1])
    counter0 = int(env.counter[env_index].item())
    out = env.rollout(T, actions=actions, shared_actions=shared_actions,
                      want=("traj", "obs", "rew", "done", "actions"))
    g = lambda k: out[k][:, env_index].cpu().numpy()  # noqa: E731
    goals = None
    if env.goal_table is not None and not env.cfg.auto_reset:
        goals = _goals_of(env, T, env_index, counter0)
    return episodes_from_rollout(obs0, g("traj"), g("obs"), g("actions"), g("rew"), g("done"), goals=goals,
                                 auto_reset=bool(env.cfg.auto_reset))


class BatchedEpisodes:
    """The episodes of ALL N envs of one resident [T, N, .] rollout, cut on the device with tensor operations (no Python loop
    over envs): episode index of every transition = exclusive cumulative sum of `done` along T; terminal observations of
    auto-reset steps rebuilt from the recorded fp64 position; reset rows = the observation the previous episode's last step
    returned.  `dict_of(i)` / `dicts(envs)` give MRExperiment-layout dictionaries (MR_data.py:27-57) identical to
    record_rollout(env_index=i); the requested envs' columns travel to the host in ONE transfer.

    Tensors (device): ep_index [T, N] int64, ep_count [N] (episodes, the unfinished tail included), ep_len [N, E] transitions
    per episode (E = max ep_count), ep_return [N, E] float64, step_obs [T, N, 5] float64 (terminal rows rebuilt), obs0 [N, 5]."""

    def __init__(self, obs0, pos0, traj, obs, actions, rew, done, auto_reset, goals=None):
        import torch
        T, N = rew.shape
        self.T, self.N, self.auto_reset = T, N, bool(auto_reset)
        f64 = torch.float64
        done = done.bool()
        o0 = obs0.to(f64).clone()
        # the reset state in fp64 where the float32 observation is the rounded position (record_rollout does the same per env)
        same = (o0[:, 0] == pos0[:, 0].float().to(f64)) & (o0[:, 1] == pos0[:, 1].float().to(f64))
        o0[:, 0] = torch.where(same, pos0[:, 0], o0[:, 0]); o0[:, 1] = torch.where(same, pos0[:, 1], o0[:, 1])
        o0[:, 4] = torch.where(same, torch.hypot(o0[:, 2] - pos0[:, 0], o0[:, 3] - pos0[:, 1]), o0[:, 4])
        self.obs0, self._same0 = o0, same
        d_i = done.long()
        self.ep_index = torch.cumsum(d_i, 0) - d_i                                  # episodes finished BEFORE step t
        tail = ~done[T - 1]
        self.ep_count = d_i.sum(0) + tail.long()
        E = int(self.ep_count.max().item()) if N else 0
        self.ep_len = torch.zeros((N, max(E, 1)), dtype=torch.int64, device=rew.device)
        self.ep_len.scatter_add_(1, self.ep_index.t().contiguous(), torch.ones((N, T), dtype=torch.int64, device=rew.device))
        self.ep_return = torch.zeros((N, max(E, 1)), dtype=f64, device=rew.device)
        self.ep_return.scatter_add_(1, self.ep_index.t().contiguous(), rew.to(f64).t().contiguous())
        so = obs.to(f64).clone()
        if self.auto_reset:
            # the returned observation of a done step is the NEXT episode's reset observation: rebuild the terminal one from the
            # recorded position and the goal (of that step when `goals` [T, N, 2] is given, else the episode's reset observation's)
            if goals is None:
                # goal shown by the reset observation of the episode each step belongs to: obs0 for episode 0, else the observation
                # returned by the previous done step -- carried forward along T
                g_reset = torch.where(done[:, :, None], obs[:, :, 2:4].to(f64), torch.full_like(obs[:, :, 2:4].to(f64), float("nan")))
                shifted = torch.cat([o0[None, :, 2:4], g_reset[:-1]], 0)            # value that applies FROM step t on
                idx = torch.where(torch.isnan(shifted[:, :, 0]), torch.zeros_like(d_i), torch.arange(T, device=rew.device)[:, None].expand(T, N))
                last = torch.cummax(idx, 0).values                                   # most recent step that set a goal
                goal = torch.gather(shifted, 0, last[:, :, None].expand(T, N, 2))
            else:
                goal = goals.to(f64)
            term = torch.stack([traj[:, :, 0], traj[:, :, 1], goal[:, :, 0], goal[:, :, 1],
                                torch.hypot(goal[:, :, 0] - traj[:, :, 0], goal[:, :, 1] - traj[:, :, 1])], 2)
            so = torch.where(done[:, :, None], term, so)
        self.step_obs = so
        self._traj, self._obs, self._act, self._rew, self._done = traj, obs, actions, rew, done

    def dicts(self, envs):
        """[MRExperiment-layout dict for env i for i in envs]; one device-to-host transfer for all of them"""
        import torch
        idx = torch.as_tensor(list(envs), dtype=torch.long, device=self._rew.device)
        sel = lambda t: t.index_select(1, idx).cpu().numpy()  # noqa: E731
        traj, so, ob, act, rew, done = (sel(t) for t in (self._traj, self.step_obs, self._obs.double(), self._act.double(),
                                                          self._rew.double(), self._done))
        o0 = self.obs0.index_select(0, idx).cpu().numpy()
        same0 = self._same0.index_select(0, idx).cpu().numpy()
        out = []
        for c in range(len(idx)):
            # distances recomputed with numpy's hypot (the device's may differ in the last place): bit-equal to record_rollout
            if same0[c]:
                o0[c, 4] = np.hypot(o0[c, 2] - o0[c, 0], o0[c, 3] - o0[c, 1])
            if self.auto_reset:
                m = done[:, c]
                so[m, c, 4] = np.hypot(so[m, c, 2] - so[m, c, 0], so[m, c, 3] - so[m, c, 1])
            r = rew[:, c]
            integer = bool(np.all(r == np.round(r)))
            d = _empty()
            bounds = np.flatnonzero(done[:, c]) + 1
            starts = np.concatenate([[0], bounds])
            ends = np.concatenate([bounds, [self.T]])
            reset_obs = o0[c]
            for a, b in zip(starts, ends):
                if b == a:
                    break
                it = d["iterations"] = d["iterations"] + 1
                d["states"][it] = np.vstack([reset_obs[:2], traj[a:b, c]])
                d["observations"][it] = np.vstack([reset_obs, so[a:b, c]])
                d["actions"][it] = np.vstack([np.zeros(2), act[a:b, c]])
                d["rewards"][it] = _rewards_column(r[a:b], integer)
                d["steps"][it] = int(b - a)
                if b < self.T or done[b - 1, c]:
                    reset_obs = ob[b - 1, c]
            out.append(d)
        return out

    def dict_of(self, env_index):
        return self.dicts([env_index])[0]


def export_all(env, T, actions=None, shared_actions=False):
    """Run a fused T-step rollout on `env` and cut the episodes of ALL its envs on the device (BatchedEpisodes): the batched form
    of record_rollout -- MRExperiment's per-step np.vstack bookkeeping (MR_data.py:44-57) for N envs at once."""
    obs0, pos0 = env.obs.clone(), env.pos.clone()
    counter0 = env.counter.clone()
    out = env.rollout(T, actions=actions, shared_actions=shared_actions, want=("traj", "obs", "rew", "done", "actions"))
    goals = None
    if env.goal_table is not None and not env.cfg.auto_reset:       # one episode per env, the goal follows the table
        import torch
        K, Tg = env.goal_table.shape[0], env.goal_table.shape[1]
        k = (env.env_id0 + torch.arange(env.num_envs, device=env.device)) % K
        rows = (counter0.long()[None, :] + 1 + torch.arange(T, device=env.device)[:, None]).clamp(0, Tg - 1)
        goals = env.goal_table[k[None, :].expand(T, -1), rows]
    return BatchedEpisodes(obs0, pos0, out["traj"], out["obs"], out["actions"], out["rew"], out["done"], env.cfg.auto_reset, goals=goals)


def record_episodes(env, inits, action_tables, env_index=0, **reset_kwargs):
    """The reference's recorded loop (`env.set_save_experice(name)`; per episode `env.reset(init)` then `env.step(a)`
    until done -- MR_env.py:94-95,190-198) for explicit start positions: episode k resets every env of `env` to
    inits[k], plays action_tables[k] ([T_k, 2], shared by all envs) as ONE fused launch and keeps the transitions up to
    and including the first done of env `env_index`.  Returns one MRExperiment-layout dict holding all episodes."""
    if env.cfg.auto_reset:
        raise ValueError("record_episodes drives explicit resets: construct the env with auto_reset=False")
    d = _empty()
    for init, table in zip(inits, action_tables):
        table = np.asarray(table)
        T = len(table)
        init = np.asarray(init, dtype=np.float64)
        env.reset(init=np.tile(init[None, :], (env.num_envs, 1)), **reset_kwargs)
        one = record_rollout(env, T, env_index=env_index, actions=table[:, :2], shared_actions=True)
        it = d["iterations"] = d["iterations"] + 1
        for key in ("states", "observations", "actions", "rewards", "steps"):
            d[key][it] = one[key][0]  # the first episode of the launch: everything after its done is dropped
    return d


def save_experiment(d, path):
    """Same bytes layout as MRExperiment.save_experiment (pickle protocol 2 of the instance dict)."""
    with open(path, "wb") as f:
        pickle.dump(d, f, 2)


def load_experiment(path):
    """Only for files written by save_experiment above (never for pickles shipped with the reference)."""
    with open(path, "rb") as f:
        return pickle.load(f)
