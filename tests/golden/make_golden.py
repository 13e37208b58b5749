#!/usr/bin/env python3
"""Generate golden vectors for the MR_env.step()/MR_simulator hot path.

Runs ONLY in the build container (needs /root/reference, numpy, scipy).  The
reference's own files are imported unmodified from where they lie; nothing of
them is copied.  What is committed is data: inputs and the reference's outputs.

  ref_sim.npz   MR_simulator.Simulator driven directly (no stand-ins at all):
                positions + the RK45 object's carried state (t, h_abs, f) and
                Simulator.state_prime after every step.          (G1..G5)
  ref_noise.npz Simulator runs with noise_var > 0 where every value returned by
                numpy.random.normal is recorded in draw order ("tape").  The
                oracle replays the tape, which pins its noise plumbing (draw
                order, stage weights, error control, constructor draws) against
                the real reference even though MT19937 itself is not restated.  (G8)
  ref_env.npz   MR_env.MR_Env episodes and utils.run_sim output.  `gym`,
                `turtle` and `tkinter` are not installed in this image, so
                MR_env.py is imported against the minimal stand-ins below; the
                only stand-in behaviour that reaches a result is
                Box.contains (numeric bounds check) and Box.sample -- the
                out-of-bounds termination is therefore documented as
                "parity unpinned (gym absent)" in DESIGN.md.     (G6, G7)

  ref_reused.npz  ONE MR_Env object driven through three consecutive episodes with reset(is_mismatched=True) each time
                (the loop shape of RL/MR_ddpg.py:270-311): from the second episode on reset_start_pos builds the RK45
                object under the law the PREVIOUS episode left behind (MR_env.py:181-183), sigma = 0 and one taped sigma > 0 run.
  ref_sim_f64.npz  the ref_sim scenarios with the reference's own float64 action tables (not rounded to float32).
  ref_experiment.npz  MR_data.MRExperiment's dictionaries after three recorded MR_Env episodes (8f-3).
  ref_circle_fm.npz  utils.run_sim on main_2d.py:137-160's frequency-modulated circle learning set (nominal and mismatched).
  ref_increments.npz  sigma > 0 STATISTICS of the reference itself (schema 2, round 5): per-step noise increments of Simulator.step at
                sigma = 1, random actions in the actor range, nominal and mismatched law -- 1e6 env steps per law in the DDPG regime
                (start (110, 115), 10 000 restarts of 100 steps), 1e6 per law at start (8, -6) (first-attempt error_norm ~ 1) and 4e5 per
                law and start where SciPy's error controller splits every step (starts (0, 0) and (0.5, -0.2); episodes of 8 steps), with the rk_step attempts of every step (integrator.nfev):
                Delta - dt (b1 K0 + (1 - b1) V(action)) in units of dt cB sqrt(g^2 + sigma^2), stored as sorted-sample quantiles at
                fixed ranks + moments + attempts histograms (pooled, per step index, per episode).  The oracle's and the kernels'
                samples (per-stage and collapsed noise law, fast and spec Box-Muller) are tested against these (KS on the stored
                ECDF points, |std ratio - 1| < 0.005, chi-square on the attempts), not only against the formula of SURVEY 3.3.

  ref_main_runs.npz  main.py's own two noisy run_sim calls (idle set; first half circle of the learning set; noise_var 0.5, a0 1.5,
                mismatched, from the origin: main.py:9-84) repeated 4000 times each with the imported Simulator: position quantiles,
                moments and cumulative rk_step attempts at five checkpoints each.

Usage:  python tests/golden/make_golden.py   (writes next to this file)
"""
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True  # /root/reference is read-only

import numpy as np

REF = os.environ.get("MRSIM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import MR_simulator  # noqa: E402  (reference, unmodified)


# --------------------------------------------------------------------------
# action tables (inputs)
# --------------------------------------------------------------------------
def actions_straight(T=1000):
    a = np.zeros((T, 2))
    a[:, 0] = 4.0
    a[:, 1] = np.pi / 4
    return a


def actions_ramp(T=1000, freq=4.0):
    """The alpha-ramp test profile of the reference's main.py:39-50."""
    a = np.zeros((T, 2))
    a[0:200, 1] = np.linspace(0, np.pi / 2, 200)
    a[200:400, 1] = np.linspace(np.pi / 2, -np.pi / 2, 200)
    a[400:600, 1] = np.linspace(-np.pi / 2, 0, 200)
    a[600:800, 1] = np.linspace(0, np.pi / 8, 200)
    a[800:, 1] = np.linspace(np.pi / 8, -np.pi, 200)
    a[:, 0] = freq
    return a


def actions_figure8(T=1000):
    """v = (cos th, cos 2th), f = 4|v|, alpha = atan2."""
    th = 2 * np.pi * np.arange(T) / T
    vx, vy = np.cos(th), np.cos(2 * th)
    a = np.zeros((T, 2))
    a[:, 0] = 4.0 * np.hypot(vx, vy)
    a[:, 1] = np.arctan2(vy, vx)
    return a


def actions_random(T, rng, idle_frac=0.1, wide=False):
    a = np.zeros((T, 2))
    if wide:  # DDPG actor range, RL/MR_ddpg.py:136-137,345 (never clipped by the env)
        a[:, 0] = rng.uniform(-20, 20, T)
        a[:, 1] = rng.uniform(-2 * np.pi, 2 * np.pi, T)
    else:
        a[:, 0] = rng.uniform(0, 20, T)
        a[:, 1] = rng.uniform(0, 2 * np.pi, T)
    idle = rng.uniform(size=T) < idle_frac
    a[idle] = 0.0
    return a


# --------------------------------------------------------------------------
# drive the reference Simulator
# --------------------------------------------------------------------------
def f32(actions):
    """The C ABI takes float32 actions, so every action table is rounded to float32 BEFORE it is fed to
    the reference: reference, oracle and kernel then see bit-identical inputs."""
    return np.asarray(actions, dtype=np.float64).astype(np.float32).astype(np.float64)


def run_simulator(actions, init, a0, noise_var, mismatched, mismatch_at_reset=False):
    """MR_env.reset order (MR_env.py:179-183): noise_var, a0, reset_start_pos, then is_mismatched."""
    s = MR_simulator.Simulator()
    s.is_mismatched = bool(mismatch_at_reset)
    s.noise_var = noise_var
    s.a0 = a0
    s.reset_start_pos(np.asarray(init, dtype=np.float64))
    s.is_mismatched = bool(mismatched)
    T = len(actions)
    out = dict(
        pos=np.zeros((T, 2)), t=np.zeros(T), h_abs=np.zeros(T), f=np.zeros((T, 2)),
        state_prime=np.zeros((T, 2)),
        reset_h_abs=np.float64(s.integrator.h_abs), reset_f=np.array(s.integrator.f, dtype=np.float64),
        reset_state_prime=np.array(s.state_prime, dtype=np.float64),
    )
    for k, (f_t, al) in enumerate(actions):
        out["pos"][k] = s.step(f_t, al)
        out["t"][k] = s.integrator.t
        out["h_abs"][k] = s.integrator.h_abs
        out["f"][k] = s.integrator.f
        out["state_prime"][k] = s.state_prime
    return out


def gen_sim(round_f32=True):
    """round_f32=False: the reference's own float64 action tables (main.py:14-50 builds float64 linspace tables),
    written to ref_sim_f64.npz -- pins the fp64 action-table input of the rollout (MrsimRolloutIO.actions_f64)."""
    rng = np.random.default_rng(20261004)
    cases = {}

    def add(name, actions, init, a0, mismatched=False, mismatch_at_reset=False):
        actions = f32(actions) if round_f32 else np.asarray(actions, dtype=np.float64)
        r = run_simulator(actions, init, a0, 0.0, mismatched, mismatch_at_reset)
        cases[name] = dict(actions=actions, init=np.asarray(init, float), a0=a0,
                           mismatched=int(mismatched), mismatch_at_reset=int(mismatch_at_reset), **r)

    add("g1_straight", actions_straight(), [0.0, 0.0], 1.0)
    add("g2_ramp", actions_ramp(), [0.0, 0.0], 1.5)
    add("g3_figure8", actions_figure8(), [0.0, 0.0], 1.0)
    add("g4_rand_far", actions_random(2000, rng), [110.0, 115.0], 1.0)
    add("g4_rand_origin", actions_random(2000, rng), [0.0, 0.0], 1.0)
    add("g4_rand_near", actions_random(2000, rng), [0.5, -0.2], 1.0)
    add("g4_rand_wide", actions_random(2000, rng, wide=True), [104.25, 118.5], 1.0)
    add("g4_rand_neg", actions_random(2000, rng, wide=True), [-300.0, 4000.0], 0.7)
    add("g5_mis_const", np.tile([[4.0, 0.3]], (1000, 1)), [0.0, 0.0], 1.5, mismatched=True)
    add("g5_mis_ramp", actions_ramp(), [0.0, 0.0], 1.5, mismatched=True)
    add("g5_mis_rand", actions_random(2000, rng), [110.0, 115.0], 1.0, mismatched=True)
    add("g5_mis_reused", actions_random(500, rng), [3.0, -2.0], 1.0, mismatched=True, mismatch_at_reset=True)

    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}/{k}"] = np.asarray(v)
    fname = "ref_sim.npz" if round_f32 else "ref_sim_f64.npz"
    np.savez_compressed(os.path.join(HERE, fname), **flat)
    print(fname + ":", ", ".join(cases))


# --------------------------------------------------------------------------
# noisy runs with a recorded tape of the reference's own normal draws
# --------------------------------------------------------------------------
class _Tape:
    def __init__(self):
        self.vals = []
        self._orig = np.random.normal

    def __enter__(self):
        def rec(loc=0.0, scale=1.0, size=None):
            v = self._orig(loc, scale, size)
            self.vals.extend(np.atleast_1d(v).tolist())
            return v
        np.random.normal = rec
        return self

    def __exit__(self, *a):
        np.random.normal = self._orig


def gen_noise():
    rng = np.random.default_rng(77)
    flat = {}
    specs = [
        ("n_far_s1", actions_random(400, rng, wide=True), [110.0, 115.0], 1.0, 1.0, False),
        ("n_far_s05", actions_random(400, rng), [101.5, 119.0], 1.0, 0.5, False),
        ("n_origin_s05", actions_ramp()[:400], [0.0, 0.0], 1.5, 0.5, False),
        ("n_near_s1", actions_random(400, rng), [0.3, 0.1], 1.0, 1.0, False),
        ("n_mis_s05", actions_random(400, rng), [110.0, 115.0], 1.0, 0.5, True),
        ("n_mis_origin_s1", actions_ramp()[:300], [0.0, 0.0], 1.5, 1.0, True),
    ]
    for i, (name, actions, init, a0, sigma, mis) in enumerate(specs):
        actions = f32(actions)
        np.random.seed(1000 + i)
        with _Tape() as tape:
            r = run_simulator(actions, init, a0, sigma, mis)
        d = dict(actions=actions, init=np.asarray(init, float), a0=a0, sigma=sigma,
                 mismatched=int(mis), tape=np.asarray(tape.vals), **r)
        for k, v in d.items():
            flat[f"{name}/{k}"] = np.asarray(v)
        print(f"  {name}: {len(tape.vals)} draws over {len(actions)} steps "
              f"({len(tape.vals) / len(actions):.2f}/step)")
    np.savez_compressed(os.path.join(HERE, "ref_noise.npz"), **flat)
    print("ref_noise.npz written")


# --------------------------------------------------------------------------
# MR_Env / run_sim (needs stand-ins for the absent gym / turtle / tkinter)
# --------------------------------------------------------------------------
def _install_standins():
    gym = types.ModuleType("gym")

    class Env:  # gym.Env is only used as a base class (MR_env.py:21)
        pass

    class Box:  # numeric bounds check + uniform float32 sample; nothing else is used
        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    gym.Env, gym.spaces = Env, spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    for name in ("turtle", "tkinter"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def gen_env():
    _install_standins()
    import contextlib
    import io
    import MR_env  # noqa: E402 (reference, unmodified)
    import utils   # noqa: E402 (reference, unmodified) -- run_sim

    rng = np.random.default_rng(4242)
    flat = {}

    def episode(name, init, actions, a0=1.0, noise_var=0.0, mismatched=False, stop_on_done=False):
        actions = f32(actions)
        env = MR_env.MR_Env()
        with contextlib.redirect_stdout(io.StringIO()):  # reset() prints (MR_env.py:175-176)
            obs0 = env.reset(init=np.asarray(init, dtype=np.float64), noise_var=noise_var, a0=a0,
                             is_mismatched=mismatched)
        T = len(actions)
        obs = np.zeros((T, 5)); rew = np.zeros(T); done = np.zeros(T, dtype=np.uint8)
        sp = np.zeros((T, 2)); lp = np.zeros((T, 2)); cnt = np.zeros(T, dtype=np.int32)
        calc = np.zeros(T)
        n = T
        for k, a in enumerate(actions):
            with contextlib.redirect_stdout(io.StringIO()):
                o, r, d, info = env.step(a)
                calc[k] = env.calculate_reward(o)  # defined-but-uncalled reward, MR_env.py:118-134
            obs[k], rew[k], done[k] = o, r, d
            sp[k], lp[k], cnt[k] = env.state_prime, env.last_pos, env.counter
            if d and stop_on_done:
                n = k + 1
                break
        for key, v in dict(init=np.asarray(init, float), actions=np.asarray(actions), a0=a0,
                           mismatched=int(mismatched), obs0=obs0, obs=obs[:n], rew=rew[:n],
                           done=done[:n], state_prime=sp[:n], last_pos=lp[:n], counter=cnt[:n],
                           calc_reward=calc[:n]).items():
            flat[f"{name}/{key}"] = np.asarray(v)
        print(f"  {name}: {n} steps, first done at "
              f"{(int(np.argmax(done[:n])) + 1) if done[:n].any() else None}")

    # G6a: DDPG-shaped episode (random wide actions, start in [100,120]^2): times out at step 51
    episode("g6_timeout", [112.5, 103.25], actions_random(60, rng, idle_frac=0.0, wide=True))
    # G6b: goal reach: start (40,0), f=20, alpha=pi -> d<30 at step 17
    episode("g6_goal", [40.0, 0.0], np.tile([[20.0, np.pi]], (25, 1)))
    # G6c: out of bounds: start just inside x=5000, drive +x
    episode("g6_oob", [4999.0, 10.0], np.tile([[20.0, 0.0]], (12, 1)))
    # G6d: mismatched episode
    episode("g6_mis", [110.0, 115.0], actions_random(55, rng), a0=1.5, mismatched=True)

    # G7: run_sim tuple (utils.py:43-61), ignores done
    for name, actions, init, a0, mis in [
        ("g7_runsim_ramp", actions_ramp()[:300], [0.0, 0.0], 1.5, False),
        ("g7_runsim_mis", actions_figure8()[:300], [0.0, 0.0], 1.0, True),
    ]:
        a3 = np.zeros((len(actions), 3)); a3[:, :2] = f32(actions); a3[:, 2] = np.arange(len(actions)) * 0.030
        with contextlib.redirect_stdout(io.StringIO()):
            X, Y, alpha, time, freq = utils.run_sim(a3, init_pos=np.asarray(init, float), noise_var=0.0,
                                                    a0=a0, is_mismatched=mis)
        for key, v in dict(actions=a3, init=np.asarray(init, float), a0=a0, mismatched=int(mis),
                           X=X, Y=Y, alpha=alpha, time=time, freq=freq).items():
            flat[f"{name}/{key}"] = np.asarray(v)
        print(f"  {name}: {len(X)} steps")
    np.savez_compressed(os.path.join(HERE, "ref_env.npz"), **flat)
    print("ref_env.npz written")


def gen_experiment():
    """ref_experiment.npz: what the reference's own recorder, MR_data.MRExperiment (MR_data.py:27-57), holds after a
    DDPG-style loop over the imported MR_Env with set_save_experice (MR_env.py:223-226; hooks :94-95,190-198):
    three sigma = 0 episodes -- reset(init) then step until done.  The reference pickles itself into
    ./_experiments/ from the second reset on (MR_env.py:190-192, MR_data.py:67-74), so the loop runs in a scratch
    directory that has one; that pickle (written by the reference's code here, from our inputs) is read back through
    the reference's own load_from_experiment as a cross-check and then discarded."""
    _install_standins()
    import contextlib
    import io
    import tempfile
    import MR_env   # noqa: E402 (reference, unmodified)
    import MR_data  # noqa: E402 (reference, unmodified)

    rng = np.random.default_rng(515)
    episodes = [
        ([112.5, 103.25], actions_random(60, rng, idle_frac=0.0, wide=True)),  # times out at step 51
        ([40.0, 0.0], np.tile([[20.0, np.pi]], (25, 1))),                      # reaches the goal at step 17
        ([108.0, 119.5], actions_random(60, rng, idle_frac=0.1)),              # times out at step 51
    ]
    flat = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "_experiments"))
        os.chdir(tmp)
        try:
            env = MR_env.MR_Env()
            env.set_save_experice("golden")
            for k, (init, actions) in enumerate(episodes):
                actions = f32(actions)
                with contextlib.redirect_stdout(io.StringIO()):
                    env.reset(init=np.asarray(init, dtype=np.float64), noise_var=0.0, a0=1.0, is_mismatched=False)
                n = 0
                for a in actions:
                    with contextlib.redirect_stdout(io.StringIO()):
                        _, _, d, _ = env.step(a)
                    n += 1
                    if d:
                        break
                flat[f"in/{k}/init"] = np.asarray(init, dtype=np.float64)
                flat[f"in/{k}/actions"] = actions[:n]
            exp = env.MR_data
            d = exp.__dict__
            files = sorted(os.listdir("_experiments"))
            if files:  # cross-check: the reference's own pickle, read by the reference's own loader
                chk = MR_data.MRExperiment()
                chk.load_from_experiment(files[-1])
                assert chk.iterations >= 1 and np.array_equal(chk.states[0], d["states"][0])
        finally:
            os.chdir(cwd)
    flat["iterations"] = np.asarray(d["iterations"])
    flat["time_step"] = np.asarray(d["time_step"])
    for key in ("states", "observations", "actions", "rewards"):
        for it, v in d[key].items():
            flat[f"{key}/{it}"] = np.asarray(v)
    for it, v in d["steps"].items():
        flat[f"steps/{it}"] = np.asarray(v)
    flat["keys"] = np.asarray(sorted(d.keys()))
    np.savez_compressed(os.path.join(HERE, "ref_experiment.npz"), **flat)
    print("ref_experiment.npz: iterations", d["iterations"], "steps", dict(d["steps"]),
          "dtypes", {k: str(np.asarray(d[k][0]).dtype) for k in ("states", "observations", "actions", "rewards")},
          "keys", sorted(d.keys()))


def gen_reused():
    """One env object, several episodes (`state = env.reset(...)` at the top of every episode, RL/MR_ddpg.py:270).  Every
    episode starts at the same float32-representable point, so a vec env whose init box is that single point replays it
    through its in-kernel auto-reset."""
    _install_standins()
    import contextlib
    import io
    import MR_env  # noqa: E402 (reference, unmodified)
    rng = np.random.default_rng(909)
    flat = {}
    init = np.array([110.5, 104.25])
    n_ep, max_steps = 3, 60
    for name, sigma, mis, a0 in (("reused_mis_s0", 0.0, True, 1.5), ("reused_mis_s05", 0.5, True, 1.0),
                                 ("reused_nom_s0", 0.0, False, 1.0)):
        actions = f32(actions_random(n_ep * max_steps, rng, idle_frac=0.05))
        env = MR_env.MR_Env()
        np.random.seed(4321)
        rows = dict(pos=[], obs=[], done=[], counter=[], episode=[], reset_f=[], reset_h_abs=[], f=[], h_abs=[])
        used = []
        with _Tape() as tape:
            k = 0
            for ep in range(n_ep):
                with contextlib.redirect_stdout(io.StringIO()):
                    env.reset(init=init.copy(), noise_var=sigma, a0=a0, is_mismatched=mis)
                rows["reset_f"].append(np.array(env.simulator.integrator.f, dtype=np.float64))
                rows["reset_h_abs"].append(float(env.simulator.integrator.h_abs))
                for j in range(max_steps):
                    a = actions[k]; k += 1
                    used.append(a)
                    with contextlib.redirect_stdout(io.StringIO()):
                        o, r, d, _ = env.step(a)
                    rows["pos"].append(np.array(env.last_pos, dtype=np.float64)); rows["obs"].append(np.array(o, dtype=np.float64))
                    rows["done"].append(int(d)); rows["counter"].append(env.counter); rows["episode"].append(ep)
                    rows["f"].append(np.array(env.simulator.integrator.f, dtype=np.float64))
                    rows["h_abs"].append(float(env.simulator.integrator.h_abs))
                    if d:
                        break
        d = dict(init=init, a0=a0, sigma=sigma, mismatched=int(mis), actions=np.asarray(used), tape=np.asarray(tape.vals),
                 **{k2: np.asarray(v) for k2, v in rows.items()})
        for key, v in d.items():
            flat[f"{name}/{key}"] = np.asarray(v)
        print(f"  {name}: {len(used)} steps over {n_ep} episodes, {len(tape.vals)} draws, reset f = {rows['reset_f']}")
    np.savez_compressed(os.path.join(HERE, "ref_reused.npz"), **flat)
    print("ref_reused.npz written")


def gen_circle_fm():
    """ref_circle_fm.npz: utils.run_sim on the frequency-modulated circle learning set of main_2d.py:137-160 (three circles of
    100 steps, alpha = linspace(-pi, pi), f = (cos(t / 5) + 1) / 2 * 4.9 + 0.1 with t = linspace(0, 300, 300)), a0 = 1.5."""
    _install_standins()
    import contextlib
    import io
    import utils   # noqa: E402 (reference, unmodified)
    time_steps, cycles = 300, 3
    steps = int(time_steps / cycles)
    circle = np.zeros((steps, 2))
    circle[:, 1] = np.linspace(-np.pi, np.pi, steps)
    learn = np.vstack([circle] * cycles)
    t = np.linspace(0, time_steps, time_steps)
    learn[:, 0] = (np.cos(t / 5) + 1) / 2 * 4.9 + 0.1
    a3 = np.zeros((time_steps, 3)); a3[:, :2] = f32(learn); a3[:, 2] = np.arange(time_steps) * 0.030
    flat = {}
    for name, mis in (("g7_circle_fm", False), ("g7_circle_fm_mis", True)):
        with contextlib.redirect_stdout(io.StringIO()):
            X, Y, alpha, time, freq = utils.run_sim(a3, init_pos=np.array([0.0, 0.0]), noise_var=0.0, a0=1.5, is_mismatched=mis)
        for key, v in dict(actions=a3, actions_f64=learn, init=np.zeros(2), a0=1.5, mismatched=int(mis), X=X, Y=Y, alpha=alpha, time=time,
                           freq=freq).items():
            flat[f"{name}/{key}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "ref_circle_fm.npz"), **flat)
    print("ref_circle_fm.npz written")


# --------------------------------------------------------------------------
# sigma > 0 statistics of the reference itself, at scale (round 5)
# --------------------------------------------------------------------------
INC_DT, INC_B1 = 0.030, 35.0 / 384
INC_CB = float(np.sqrt((500 / 1113) ** 2 + (125 / 192) ** 2 + (2187 / 6784) ** 2 + (11 / 84) ** 2))
INC_ATT_BINS = 128
# name -> (start, steps per episode, episodes, quantile points).  "far" = the DDPG regime (steps are never split, the pooled
# increments are iid); "origin" / "near" = starts where the error controller of SciPy's RK45 rejects and splits steps
# (MR_simulator.py:42-43 under the noise of :79-83): episodes are SHORT and many, statistics are kept per step index.
INC_REGIMES = {
    "far": ((110.0, 115.0), 100, 10000, 8193),
    "mid": ((8.0, -6.0), 8, 125000, 1025),      # error_norm of the first attempt ~ 1: accept / reject is a coin flip
    "origin": ((0.0, 0.0), 8, 50000, 1025),     # 20 .. 50 attempts per env step
    "near": ((0.5, -0.2), 8, 50000, 1025),
}


def inc_normalised_residual(p0, p1, k0, actions, mis, sigma, a0=1.0):
    """(Delta - dt (b1 K0 + (1 - b1) V(action))) / (dt cB sqrt(g^2 + sigma^2)) for arrays [..., 2]: the part of an env step's
    displacement that the noise (and, where the step is split, the splitting) is responsible for, in units of the unsplit
    step's noise std.  Shared by this script and the tests (tests/test_noise_law_cpu.py imports the formula's twin)."""
    f, al = actions[..., 0], actions[..., 1]
    if mis:
        a0b = a0 + (f / 4) * 0.8
        V = np.stack([a0b * f * np.cos(al + 0.1) + 0.2, a0b * f * np.sin(al - 0.15) - 0.1], -1)
        g = 0.25 * sigma * f[..., None] * np.stack([np.cos(al + 0.1), np.sin(al - 0.15)], -1)
    else:
        V = np.stack([a0 * f * np.cos(al), a0 * f * np.sin(al)], -1)
        g = np.zeros_like(V)
    return (p1 - p0 - INC_DT * (INC_B1 * k0 + (1 - INC_B1) * V)) / (INC_DT * INC_CB * np.sqrt(g * g + sigma * sigma))


def _inc_chunk(task):
    """One chunk of episodes of the imported reference Simulator (a worker process; seeds fixed per chunk, so the fixture does
    not depend on how chunks are scheduled).  Attempts of an env step = rk_step calls = (nfev after - nfev before) / 6 of the
    RK45 object that integrates it (5 stages + f_new per attempt; MR_simulator.py:42-43)."""
    start, steps, n_ep, mis, sigma, seed = task
    rng = np.random.default_rng(seed)
    np.random.seed(seed % (2 ** 32))
    resn = np.zeros((n_ep, steps, 2))
    att = np.zeros((n_ep, steps), dtype=np.int32)
    for ep in range(n_ep):
        s = MR_simulator.Simulator()
        s.noise_var = sigma
        s.a0 = 1.0
        s.reset_start_pos(np.array(start, dtype=np.float64))
        s.is_mismatched = mis          # MR_env.py:181-183: set AFTER the integrator was built
        acts = f32(actions_random(steps, rng, idle_frac=0.0, wide=True))
        for k in range(steps):
            integ = s.integrator
            k0 = np.array(integ.f, dtype=np.float64)
            p0 = np.array(integ.y, dtype=np.float64)
            n0 = integ.nfev
            p1 = np.array(s.step(acts[k, 0], acts[k, 1]), dtype=np.float64)
            dn = integ.nfev - n0
            assert dn % 6 == 0 and dn >= 6
            att[ep, k] = dn // 6
            resn[ep, k] = inc_normalised_residual(p0, p1, k0, acts[k], mis, sigma)
    return resn, att


def inc_summary(resn, att, n_q, per_step):
    """The committed summary of a sample resn [E, S, 2], att [E, S]: sorted-sample quantiles at fixed ranks (the ECDF is exact at
    those points: F(q[i]) = (rank[i] + 1) / n), central moments, attempts histograms; pooled over all env steps and -- per_step --
    for every step index; plus per-EPISODE reductions (episodes are independent, steps of one episode are not)."""
    E, S, _ = resn.shape

    def one(x, nq):          # x [m, 2]
        m = x.shape[0]
        ranks = np.round(np.linspace(0.0, 1.0, nq) * (m - 1)).astype(np.int64)
        xs = np.sort(x, axis=0)
        mu = x.mean(axis=0)
        c = x - mu
        return dict(q=xs[ranks].T.astype(np.float32), ranks=ranks, n=np.int64(m), mean=mu, var=(c ** 2).mean(axis=0),
                    m3=(c ** 3).mean(axis=0), m4=(c ** 4).mean(axis=0), cov_xy=np.float64((c[:, 0] * c[:, 1]).mean()))

    out = {}
    for k, v in one(resn.reshape(-1, 2), n_q if not per_step else 4097).items():
        out["pooled/" + k] = v
    out["pooled/att_hist"] = np.bincount(np.minimum(att.ravel(), INC_ATT_BINS - 1), minlength=INC_ATT_BINS).astype(np.int64)
    # per episode: mean square of the episode's residuals (the variance estimator whose standard error is honest under
    # within-episode dependence), the episode's summed residual and its total attempts
    ssq = (resn ** 2).mean(axis=1)                                  # [E, 2]
    out["episode/sumsq_mean"] = ssq.mean(axis=0)
    out["episode/sumsq_var"] = ssq.var(axis=0)
    out["episode/n"] = np.int64(E)
    if per_step:
        per = [one(resn[:, k], n_q) for k in range(S)]
        for key in per[0]:
            out["step/" + key] = np.stack([np.asarray(p[key]) for p in per])
        out["step/att_hist"] = np.stack([np.bincount(np.minimum(att[:, k], INC_ATT_BINS - 1), minlength=INC_ATT_BINS)
                                         for k in range(S)]).astype(np.int64)
        for k, v in one(resn.sum(axis=1), 4097).items():
            out["episode/sum_" + k] = v
        tot = att.sum(axis=1)
        out["episode/att_total_hist"] = np.bincount(np.minimum(tot, 1023), minlength=1024).astype(np.int64)
    return out


def gen_increments(workers=None):
    """ref_increments.npz (schema 2): statistics of the noise increments of the imported Simulator.step at sigma = 1, nominal and
    mismatched law, random actions in the DDPG actor range -- 1e6 env steps per law in the far regime (start (110, 115), restarts
    of 100 steps), 1e6 per law where the first attempt's error_norm is about 1 (start (8, -6), 125 000 episodes of 8 steps) and
    4e5 per law and start where every step is split into 20 .. 50 attempts (starts (0, 0) and (0.5, -0.2), 50 000 episodes of
    8 steps), with the rk_step attempts of every env step (from integrator.nfev).  Stored as sorted-sample quantiles at fixed
    ranks + moments + attempts histograms so that the fixture stays small; the tests compare the oracle's and the kernels'
    samples (per-stage and collapsed noise law) against these: KS on the stored ECDF points, |std ratio - 1| < 0.005,
    chi-square on the attempts histograms."""
    import multiprocessing as mp
    import time
    workers = workers or min(8, os.cpu_count() or 1)
    flat = {"schema": np.int64(2), "sigma": np.float64(1.0), "a0": np.float64(1.0), "att_bins": np.int64(INC_ATT_BINS)}
    with mp.Pool(workers) as pool:
        for regime, (start, steps, n_ep, n_q) in INC_REGIMES.items():
            for mis in (False, True):
                name = f"{regime}_{'mismatched' if mis else 'nominal'}_s1"
                chunk = 500 if regime == "far" else 2500
                base = 50000 + 1000 * list(INC_REGIMES).index(regime) + (500 if mis else 0)
                tasks = [(start, steps, chunk, mis, 1.0, base * 1000 + c) for c in range(n_ep // chunk)]
                t0 = time.time()
                parts = pool.map(_inc_chunk, tasks, chunksize=1)
                resn = np.concatenate([p[0] for p in parts])
                att = np.concatenate([p[1] for p in parts])
                for k, v in inc_summary(resn, att, n_q, per_step=(regime != "far")).items():
                    flat[f"{name}/{k}"] = v
                flat[f"{name}/start"] = np.asarray(start)
                flat[f"{name}/steps"] = np.int64(steps)
                flat[f"{name}/mismatched"] = np.int64(mis)
                print(f"  {name}: {resn.shape[0] * steps} env steps in {time.time() - t0:.0f} s; pooled std {resn.reshape(-1, 2).std(axis=0)}, "
                      f"mean attempts {att.mean():.4f} (max {att.max()}), nfev per step = 6 attempts + 2", flush=True)
    np.savez_compressed(os.path.join(HERE, "ref_increments.npz"), **flat)
    print("ref_increments.npz written")


# --------------------------------------------------------------------------
# the reference's own experiment (main.py:9-84) repeated: distributions of where run_sim ends up
# --------------------------------------------------------------------------
MAIN_RUNS = {   # name -> (action table, checkpoints (1-based step counts), repetitions)
    # main.py:15-17,57-60: 100 idle steps (zero action) from the origin: the drift (0.2, -0.1) of the mismatched model + noise,
    # every step split into tens of attempts (|y| stays below 1)
    "idle": (lambda: np.zeros((100, 2)), (1, 5, 20, 50, 100), 4000),
    # main.py:21-33,67-70: the first half circle of the learning set (freq 4, alpha from -pi over 300 of its 600 steps): starts in the
    # splitting regime and leaves it (a0' f = 9.2 units/s: 0.28 per step)
    "learn": (lambda: np.stack([np.full(300, 4.0), np.linspace(-np.pi, np.pi, 600)[:300]], 1), (1, 10, 50, 150, 300), 4000),
}
MAIN_Q = 513


def _main_chunk(task):
    """`n` repetitions of the reference's Simulator over one of main.py's action tables at main.py's parameters (noise_var = 0.5,
    a0 = 1.5, is_mismatched = True, start (0, 0): MR_env.reset's order -- mismatch set after the integrator is built); positions and
    cumulative rk_step attempts at the checkpoints."""
    name, n, seed = task
    table_fn, cps, _ = MAIN_RUNS[name]
    acts = f32(table_fn())
    np.random.seed(seed % (2 ** 32))
    pos = np.zeros((n, len(cps), 2))
    att = np.zeros((n, len(cps)), dtype=np.int64)
    for r in range(n):
        s = MR_simulator.Simulator()
        s.noise_var = 0.5
        s.a0 = 1.5
        s.reset_start_pos(np.array([0.0, 0.0]))
        s.is_mismatched = True
        total, ci = 0, 0
        for k in range(len(acts)):
            integ = s.integrator
            n0 = integ.nfev
            p1 = s.step(acts[k, 0], acts[k, 1])
            total += (integ.nfev - n0) // 6
            if k + 1 == cps[ci]:
                pos[r, ci] = p1
                att[r, ci] = total
                ci += 1
                if ci == len(cps):
                    break
    return pos, att


def gen_main_runs(workers=None):
    """ref_main_runs.npz: main.py's own two noisy run_sim calls (idle set, first half circle of the learning set; noise_var 0.5,
    a0 1.5, mismatched, from the origin) repeated 4000 times each with the imported Simulator: per checkpoint the sorted-sample
    quantiles of x and y, their moments and covariance, and the cumulative rk_step attempts -- what the batched run_sim of the build
    (oracle and kernels, both noise laws) is compared with in distribution (tests/increments.py: compare_checkpoints)."""
    import multiprocessing as mp
    import time
    workers = workers or min(8, os.cpu_count() or 1)
    flat = {"schema": np.int64(1), "sigma": np.float64(0.5), "a0": np.float64(1.5), "mismatched": np.int64(1)}
    with mp.Pool(workers) as pool:
        for name, (table_fn, cps, reps) in MAIN_RUNS.items():
            t0 = time.time()
            chunk = 100
            parts = pool.map(_main_chunk, [(name, chunk, 880000 + 1000 * list(MAIN_RUNS).index(name) + c) for c in range(reps // chunk)],
                             chunksize=1)
            pos = np.concatenate([p[0] for p in parts]); att = np.concatenate([p[1] for p in parts])
            ranks = np.round(np.linspace(0.0, 1.0, MAIN_Q) * (len(pos) - 1)).astype(np.int64)
            flat[f"{name}/actions"] = f32(table_fn())
            flat[f"{name}/checkpoints"] = np.asarray(cps, dtype=np.int64)
            flat[f"{name}/n"] = np.int64(len(pos))
            flat[f"{name}/ranks"] = ranks
            flat[f"{name}/q"] = np.stack([np.sort(pos[:, c, :], axis=0)[ranks].T for c in range(len(cps))]).astype(np.float64)  # [C, 2, Q]
            flat[f"{name}/mean"] = pos.mean(axis=0)
            flat[f"{name}/var"] = pos.var(axis=0)
            flat[f"{name}/cov_xy"] = np.array([np.cov(pos[:, c, 0], pos[:, c, 1])[0, 1] for c in range(len(cps))])
            flat[f"{name}/att_mean"] = att.mean(axis=0)
            flat[f"{name}/att_var"] = att.var(axis=0)
            print(f"  {name}: {len(pos)} runs x {cps[-1]} steps in {time.time() - t0:.0f} s; final mean {pos[:, -1].mean(axis=0)}, "
                  f"std {pos[:, -1].std(axis=0)}, attempts per step {att[:, -1].mean() / cps[-1]:.2f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "ref_main_runs.npz"), **flat)
    print("ref_main_runs.npz written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "increments":   # only the statistics fixture (the others are unchanged)
        gen_increments()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "main_runs":
        gen_main_runs()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "circle_fm":
        gen_circle_fm()
        sys.exit(0)
    import scipy
    print(f"numpy {np.__version__}, scipy {scipy.__version__}, reference at {REF}")
    if len(sys.argv) > 1 and sys.argv[1] == "reused":
        gen_reused()
        sys.exit(0)
    gen_sim()
    gen_sim(round_f32=False)
    gen_noise()
    gen_env()
    gen_experiment()
    gen_reused()
    gen_increments()
    gen_circle_fm()
    gen_main_runs()
