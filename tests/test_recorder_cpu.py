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
