"""CPU: MRExperiment-compatible recorder layout (mr_rl_amd/recorder.py)."""
import numpy as np

from mr_rl_amd.recorder import episodes_from_rollout, load_experiment, save_experiment


def test_layout_matches_mrexperiment(tmp_path):
    T = 7
    traj = np.stack([np.arange(1, T + 1) * 1.0, np.arange(1, T + 1) * 2.0], 1)
    obs = np.concatenate([traj, np.zeros((T, 2)), np.hypot(traj[:, :1], traj[:, 1:2])], 1)
    done = np.array([0, 0, 1, 0, 0, 0, 0], bool)
    obs[2] = [100.0, 110.0, 0, 0, np.hypot(100, 110)]          # auto-reset: returned obs = next reset obs
    acts = np.arange(2 * T, dtype=float).reshape(T, 2)
    rew = np.full(T, 10.0)
    obs0 = np.array([0.5, 0.25, 0, 0, np.hypot(0.5, 0.25)])
    d = episodes_from_rollout(obs0, traj, obs, acts, rew, done)
    assert d["iterations"] == 1 and d["steps"] == {0: 3, 1: 4}
    assert set(d) == {"iterations", "states", "observations", "actions", "rewards", "steps", "info", "viewer",
                      "scream", "obs_states_str", "time_step"}                     # MR_data.py:15-24
    assert d["states"][0].shape == (4, 2) and d["observations"][0].shape == (4, 5)
    assert d["actions"][0].shape == (4, 2) and d["rewards"][0].shape == (4, 1)
    np.testing.assert_array_equal(d["states"][0][0], obs0[:2])                       # reset row
    np.testing.assert_array_equal(d["actions"][0][0], [0, 0]); assert d["rewards"][0][0, 0] == 0   # MR_env.py:196-197
    np.testing.assert_array_equal(d["states"][0][1:], traj[:3])
    np.testing.assert_allclose(d["observations"][0][-1], [3, 6, 0, 0, np.hypot(3, 6)])  # terminal obs rebuilt
    np.testing.assert_array_equal(d["states"][1][0], [100.0, 110.0])                 # 2nd episode starts at the reset obs
    np.testing.assert_array_equal(d["states"][1][1:], traj[3:])
    p = tmp_path / "exp.pickle"
    save_experiment(d, p)
    e = load_experiment(p)

    class MRExperimentLike:  # what the reference does on load: self.__dict__.update(pickle.load(f))
        pass
    m = MRExperimentLike(); m.__dict__.update(e)
    assert m.iterations == 1 and np.array_equal(m.states[1], d["states"][1])


def test_layout_agrees_with_the_reference_recorder_golden():
    """tests/golden/ref_experiment.npz = MR_data.MRExperiment.__dict__ after three recorded reference episodes: same
    keys, same dtypes (int64 rewards under the constant reward), same [steps+1, width] shapes and reset rows."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_experiment.npz"))
    it = 1  # the 17-step goal-reach episode
    steps = int(g[f"steps/{it}"])
    states, obs = g[f"states/{it}"], g[f"observations/{it}"]
    # feed the recorder the reference's own rows as if they were a rollout of that env
    done = np.zeros(steps, bool); done[-1] = True
    d = episodes_from_rollout(obs[0], states[1:], obs[1:], g[f"actions/{it}"][1:], g[f"rewards/{it}"][1:, 0], done,
                              auto_reset=False)
    assert sorted(d.keys()) == [str(k) for k in g["keys"]]
    assert d["iterations"] == 0 and d["steps"][0] == steps == 17
    for key in ("states", "observations", "actions", "rewards"):
        want = g[f"{key}/{it}"]
        assert d[key][0].dtype == want.dtype and d[key][0].shape == want.shape, key
        np.testing.assert_array_equal(d[key][0], want)
    assert d["rewards"][0].dtype == np.int64


def test_goal_table_terminal_observation_and_float_rewards():
    T = 4
    traj = np.array([[1.0, 1.0], [2.0, 2.0], [3.0, 3.0], [4.0, 4.0]])
    goals = np.array([[10.0, 0.0], [20.0, 0.0], [30.0, 0.0], [40.0, 0.0]])
    obs = np.concatenate([traj, goals, np.hypot(goals[:, :1] - traj[:, :1], goals[:, 1:] - traj[:, 1:])], 1)
    done = np.array([0, 0, 1, 0], bool)
    obs[2] = [100.0, 110.0, 10.0, 0.0, 0.0]  # auto-reset replaced the terminal observation by the next reset row
    rew = np.array([-0.1, -0.1, 100.0, -0.1])
    d = episodes_from_rollout(np.array([0.0, 0.0, 10.0, 0.0, 10.0]), traj, obs, np.zeros((T, 2)), rew, done,
                              goals=goals, auto_reset=True)
    np.testing.assert_allclose(d["observations"][0][-1], [3.0, 3.0, 30.0, 0.0, np.hypot(27.0, 3.0)])  # goal of step 3
    assert d["rewards"][0].dtype == np.float64 and d["rewards"][0][-1, 0] == 100.0       # calculate_reward values


def test_saved_file_is_read_by_the_reference_loader(tmp_path, monkeypatch):
    """Build container only: the reference's own MRExperiment.load_from_experiment (MR_data.py:76-85) reads a file
    this recorder wrote (a file of ours -- nothing shipped with the reference is unpickled)."""
    import os
    import sys
    import pytest
    ref = os.environ.get("MRSIM_REFERENCE", "/root/reference")
    if not os.path.exists(os.path.join(ref, "MR_data.py")):
        pytest.skip("reference checkout not present (GPU box)")
    monkeypatch.setenv("MPLBACKEND", "Agg")
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.syspath_prepend(ref)
    import MR_data  # the reference, unmodified
    T = 5
    traj = np.cumsum(np.ones((T, 2)), 0)
    obs = np.concatenate([traj, np.zeros((T, 2)), np.hypot(traj[:, :1], traj[:, 1:])], 1)
    d = episodes_from_rollout(np.zeros(5), traj, obs, np.ones((T, 2)), np.full(T, 10.0), np.array([0, 0, 0, 0, 1], bool),
                              auto_reset=False)
    os.makedirs(tmp_path / "_experiments")
    save_experiment(d, tmp_path / "_experiments" / "ours")
    monkeypatch.chdir(tmp_path)
    exp = MR_data.MRExperiment()
    exp.load_from_experiment("ours")
    assert exp.iterations == 0 and exp.steps == {0: 5}
    np.testing.assert_array_equal(exp.states[0], d["states"][0])
    np.testing.assert_array_equal(exp.rewards[0], np.array([[0], [10], [10], [10], [10], [10]]))
