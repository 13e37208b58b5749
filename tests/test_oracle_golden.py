"""CPU: the oracle (oracle/mrsim_oracle.c) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; tolerance 1e-10 absolute on positions
(observed ~6e-14: only the summation order inside numpy.dot differs)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load_cases

SIM = load_cases("ref_sim.npz")
NOISE = load_cases("ref_noise.npz")
ENV = load_cases("ref_env.npz")
TOL = 1e-10


@pytest.mark.parametrize("name", sorted(SIM))
def test_simulator_sigma0(name):
    G = SIM[name]
    p = O.default_params(a0=float(G["a0"]), sigma=0.0, mismatched=int(G["mismatched"]))
    s = O.Sim(p)
    s.reset(*G["init"], ctor_mismatched=bool(G["mismatch_at_reset"]))
    assert abs(s.e.h_abs - G["reset_h_abs"]) <= 1e-15
    np.testing.assert_allclose(s.e.f[:], G["reset_f"], atol=1e-13)
    for k, (f, a) in enumerate(G["actions"]):
        y = s.step(f, a)
        np.testing.assert_allclose(y, G["pos"][k], rtol=0, atol=TOL, err_msg=f"{name} step {k}")
        assert abs(s.e.t - G["t"][k]) <= 1e-12
        assert abs(s.e.h_abs - G["h_abs"][k]) <= 1e-12, (name, k)
        np.testing.assert_allclose(s.e.f[:], G["f"][k], rtol=0, atol=1e-12)
        np.testing.assert_allclose(s.e.state_prime[:], G["state_prime"][k], rtol=0, atol=1e-12)


def test_known_answers_from_survey():
    """SURVEY 3.2 / 3.4 inline known answers (obtained from the reference by the survey's probe)."""
    p = O.default_params(a0=1.0, sigma=0.0)
    s = O.Sim(p); s.reset(0.0, 0.0)
    assert np.allclose(s.step(4, np.pi / 4), [0.0848525559430384] * 2, rtol=0, atol=1e-14)
    assert np.allclose(s.step(4, np.pi / 4), [0.1697053696854241] * 2, rtol=0, atol=1e-14)
    assert np.allclose(s.step(4, np.pi / 4), [0.2545581834278098] * 2, rtol=0, atol=1e-14)
    s = O.Sim(p); s.reset(110.0, 115.0)
    exp = [[109.57573722028481, 114.57573722028481], [109.61072571848085, 114.65178347978943],
           [109.94535820121828, 114.56805748849635], [109.97817652156277, 114.55850715408525],
           [110.5751772069199, 114.61840702208534]]
    acts = [(20, 3.9269908169872414), (5, 1), (12.5, 6), (0, 0), (20, 0.1)]
    for a, e in zip(acts, exp):
        assert np.allclose(s.step(*a), e, rtol=0, atol=1e-11)
    pm = O.default_params(a0=1.5, sigma=0.0, mismatched=1)
    s = O.Sim(pm); s.reset(0.0, 0.0)
    assert np.allclose(s.step(4, 0.3), [0.2602120437676086, 0.03824480836719604], rtol=0, atol=1e-13)
    assert np.allclose(s.step(4, 0.3), [0.5204248781124049, 0.07648973292990942], rtol=0, atol=1e-13)


@pytest.mark.parametrize("name", sorted(NOISE))
def test_simulator_noise_tape(name):
    """sigma > 0: replay the reference's own numpy.random.normal draws ("tape").  Pins draw order,
    stage weights, the error controller under noise and the constructor's draws."""
    G = NOISE[name]
    p = O.default_params(a0=float(G["a0"]), sigma=float(G["sigma"]), mismatched=int(G["mismatched"]))
    s = O.Sim(p, noise_kind=O.NOISE_TAPE, tape=G["tape"])
    s.reset(*G["init"])
    for k, (f, a) in enumerate(G["actions"]):
        y = s.step(f, a)
        np.testing.assert_allclose(y, G["pos"][k], rtol=0, atol=TOL, err_msg=f"{name} step {k}")
        assert abs(s.e.h_abs - G["h_abs"][k]) <= 1e-12
    assert s.nz.tape_pos == s.nz.tape_len, "oracle consumed a different number of draws than the reference"


@pytest.mark.parametrize("name", [n for n in sorted(ENV) if n.startswith("g6_")])
def test_env_episode(name):
    """MR_Env.step: obs / rew / done / counter / state_prime / calculate_reward."""
    G = ENV[name]
    p = O.default_params(a0=float(G["a0"]), sigma=0.0, mismatched=int(G["mismatched"]))
    pg = O.default_params(a0=float(G["a0"]), sigma=0.0, mismatched=int(G["mismatched"]), reward_mode=O.REW_GOAL)
    v = O.VecOracle(1, p)
    obs0 = v.reset(step_idx=0, init_xy=G["init"][None, :])
    np.testing.assert_allclose(obs0[0], G["obs0"], rtol=0, atol=1e-12)
    import ctypes as C
    for k, a in enumerate(G["actions"][: len(G["obs"])]):
        obs, rew, done = v.step(a[None, :].astype(np.float64).astype(np.float32), step_idx=k + 1)
        # actions are float32 at the ABI; the golden used float64 actions -> compare loosely on positions,
        # tightly on logic
        assert int(v.envs["counter"][0]) == int(G["counter"][k])
        assert float(rew[0]) == float(G["rew"][k]) == 10.0
        assert int(done[0]) == int(G["done"][k]), (name, k)
        np.testing.assert_allclose(obs[0], G["obs"][k], rtol=0, atol=5e-5)
        o5 = (C.c_double * 5)(*G["obs"][k])
        assert O.lib().orc_calculate_reward(C.byref(pg), o5, int(G["counter"][k])) == float(G["calc_reward"][k])


@pytest.mark.parametrize("name", [n for n in sorted(ENV) if n.startswith("g6_")])
def test_env_episode_exact_actions(name):
    """Same episodes through orc_env_step with the golden's float64 actions: tight tolerance."""
    import ctypes as C
    G = ENV[name]
    p = O.default_params(a0=float(G["a0"]), sigma=0.0, mismatched=int(G["mismatched"]))
    e = O.OrcEnv(); nz = O.OrcNoise(); nz.kind = O.NOISE_NONE
    obs = (C.c_double * 5)(); rew = C.c_double(); done = C.c_uint8()
    O.lib().orc_env_reset(C.byref(p), C.byref(e), None, G["init"][0], G["init"][1], 0, C.byref(nz), 0, obs)
    np.testing.assert_allclose(obs[:], G["obs0"], rtol=0, atol=1e-12)
    for k, a in enumerate(G["actions"][: len(G["obs"])]):
        rc = O.lib().orc_env_step(C.byref(p), C.byref(e), None, a[0], a[1], C.byref(nz), 0, obs, C.byref(rew),
                                  C.byref(done), None, None, None)
        assert rc == 0
        np.testing.assert_allclose(obs[:], G["obs"][k], rtol=0, atol=TOL)
        np.testing.assert_allclose(e.state_prime[:], G["state_prime"][k], rtol=0, atol=1e-12)
        np.testing.assert_allclose(e.y[:], G["last_pos"][k], rtol=0, atol=TOL)
        assert done.value == G["done"][k] and rew.value == G["rew"][k] and e.counter == G["counter"][k]


@pytest.mark.parametrize("name", [n for n in sorted(ENV) if n.startswith("g7_")])
def test_run_sim_tuple(name):
    """utils.run_sim (utils.py:43-61): X, Y from last_pos, time = linspace(0,(T-1)/30,T); done ignored."""
    G = ENV[name]
    p = O.default_params(a0=float(G["a0"]), sigma=0.0, mismatched=int(G["mismatched"]))
    s = O.Sim(p); s.reset(*G["init"])
    T = len(G["X"])
    X = np.zeros(T); Y = np.zeros(T)
    for k in range(T):
        X[k], Y[k] = s.step(G["actions"][k, 0], G["actions"][k, 1])
    np.testing.assert_allclose(X, G["X"], rtol=0, atol=TOL)
    np.testing.assert_allclose(Y, G["Y"], rtol=0, atol=TOL)
    np.testing.assert_allclose(np.linspace(0, (T - 1) / 30.0, T), G["time"], rtol=0, atol=0)
    np.testing.assert_array_equal(G["alpha"], G["actions"][:, 1])
    np.testing.assert_array_equal(G["freq"], G["actions"][:, 0])


CIRCLE_FM = load_cases("ref_circle_fm.npz")


@pytest.mark.parametrize("name", sorted(CIRCLE_FM))
def test_run_sim_on_the_frequency_modulated_circles(name):
    """main_2d.py:137-160's learning set (three circles, f = (cos(t / 5) + 1) / 2 * 4.9 + 0.1) through utils.run_sim: the oracle
    reproduces the reference's tuple, and mr_rl_amd.rollout.actions_circle_fm() is the table the reference was driven with"""
    from mr_rl_amd.rollout import actions_circle_fm
    G = CIRCLE_FM[name]
    tab = actions_circle_fm()
    np.testing.assert_array_equal(tab[:, :2], G["actions_f64"])
    np.testing.assert_array_equal(tab[:, :2].astype(np.float32), G["actions"][:, :2].astype(np.float32))
    np.testing.assert_allclose(tab[:, 2], G["actions"][:, 2], rtol=0, atol=1e-15)
    p = O.default_params(a0=float(G["a0"]), sigma=0.0, mismatched=int(G["mismatched"]))
    s = O.Sim(p); s.reset(*G["init"])
    T = len(G["X"])
    xy = np.array([s.step(G["actions"][k, 0], G["actions"][k, 1]) for k in range(T)])
    np.testing.assert_allclose(xy[:, 0], G["X"], rtol=0, atol=TOL)
    np.testing.assert_allclose(xy[:, 1], G["Y"], rtol=0, atol=TOL)


REUSED = load_cases("ref_reused.npz")


def _replay_reused(G, fresh_env):
    """Three consecutive episodes of ONE reference MR_Env (reset(is_mismatched=...) at the top of each, RL/MR_ddpg.py:270)
    through the oracle's same-step auto-reset: the init box is the single point the golden episodes start from."""
    import ctypes as C
    sigma = float(G["sigma"])
    init = [float(v) for v in G["init"]]
    p = O.default_params(a0=float(G["a0"]), sigma=sigma, mismatched=int(G["mismatched"]), auto_reset=1,
                         init_low=init, init_high=init, auto_reset_fresh_env=int(fresh_env))
    e, nz = O.OrcEnv(), O.OrcNoise()
    tape = np.ascontiguousarray(G["tape"], dtype=np.float64)
    if sigma > 0:
        nz.kind, nz.tape, nz.tape_len, nz.tape_pos = O.NOISE_TAPE, tape.ctypes.data_as(C.POINTER(C.c_double)), len(tape), 0
    else:
        nz.kind = O.NOISE_NONE
    L = O.lib()
    obs = (C.c_double * 5)(); fobs = (C.c_double * 5)()
    rew, done, fret, flen = C.c_double(), C.c_uint8(), C.c_double(), C.c_int32()
    L.orc_env_reset(C.byref(p), C.byref(e), None, init[0], init[1], 0, C.byref(nz), 0, obs)   # a fresh env: nominal ctor
    worst, ep, resets = 0.0, 0, [(np.array(e.f[:]), e.h_abs)]
    for k, (f, a) in enumerate(G["actions"]):
        nz.step_idx = k + 1
        assert L.orc_env_step(C.byref(p), C.byref(e), None, float(f), float(a), C.byref(nz), 0, obs, C.byref(rew),
                              C.byref(done), fobs, C.byref(fret), C.byref(flen)) == 0
        assert done.value == G["done"][k], k
        pos = np.array(fobs[:2]) if done.value else np.array(e.y[:])
        worst = max(worst, float(np.abs(pos - G["pos"][k]).max()))
        if done.value:
            ep += 1
            resets.append((np.array(e.f[:]), e.h_abs))
        else:
            assert e.counter == G["counter"][k]
    return worst, resets, (nz.tape_pos, len(tape))


@pytest.mark.parametrize("name", sorted(REUSED))
def test_auto_reset_is_the_reused_env_object(name):
    """MR_env.py:181-183 assigns is_mismatched AFTER reset_start_pos has built the RK45 object: from the second episode on
    a re-used mismatched env starts with the drift (0.2, -0.1) as its stale first stage and a full-dt first step."""
    G = REUSED[name]
    worst, resets, (used, total) = _replay_reused(G, fresh_env=False)
    assert worst < TOL, worst
    for (f, h), gf, gh in zip(resets, G["reset_f"], G["reset_h_abs"]):
        np.testing.assert_allclose(f, gf, rtol=0, atol=1e-12)
        assert abs(h - gh) <= 1e-12 * max(1.0, gh)
    if float(G["sigma"]) > 0:
        # every draw of the reference consumed, in order, the reset constructors' included; the auto-reset behind the LAST
        # episode (the reference loop stopped there) asks for one more constructor's worth: 2 RHS evaluations x 2 or 3 draws
        assert used == total + 2 * (3 if int(G["mismatched"]) else 2)
    if int(G["mismatched"]) and float(G["sigma"]) == 0:
        assert np.allclose(G["reset_f"][1], [0.2, -0.1]) and np.allclose(G["reset_f"][0], [0.0, 0.0])
        # the other meaning of an auto-reset (a fresh env per episode) is a different trajectory: its first step after a
        # reset takes a 1e-6 sub-step with K0 = 0 instead of one full step that blends b1 = 9 % of the drift into the action's
        # velocity (up to ~100 under this law): dt b1 |drift - v| ~ 0.1 per episode
        worst_fresh, _, _ = _replay_reused(G, fresh_env=True)
        assert 1e-4 < worst_fresh < 0.5, worst_fresh
