"""The COLLAPSED noise law against the per-stage law it replaces (oracle vs oracle, CPU).

Per-stage = what MR_simulator.py:73-83 does: fresh N(0, sigma) at every RHS evaluation.  Collapsed = the B- and E-weighted
stage sums of one rk_step attempt drawn directly from their joint Gaussian (oracle/mrsim_oracle.c: COL_*).  Everything a
step returns or carries must have the same distribution under both: two-sample KS on the increments, the first attempt's
error_norm, the carried integrator.f and state_prime, chi-square on the attempts per step -- >= 1e6 steps per
configuration, far from and near the origin (where the error controller splits steps), both model laws, sigma 0.5 and 1.
The family of tests is held to a family-wise 1 % level (Bonferroni), seeds fixed.
"""
import numpy as np
import pytest
from scipy import stats

from oracle import oracle as O

N_ENVS, N_STEPS = 20000, 50          # 1e6 env steps per run
CASES = [(s, start, mis) for s in (0.5, 1.0) for start in ("far", "near") for mis in (0, 1)]
STARTS = {"far": (110.0, 115.0), "near": (0.5, -0.2)}
N_TESTS = len(CASES) * 13
P_MIN = 0.01 / N_TESTS


def test_collapsed_constants_match_the_tableau():
    out = (O.C.c_double * 3)()
    O.lib().orc_collapsed_constants(out)
    B = np.array([500 / 1113, 125 / 192, -2187 / 6784, 11 / 84])
    E = np.array([71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525])
    cb = np.sqrt((B * B).sum())
    ce1 = (B * E).sum() / cb
    ce2 = np.sqrt((E * E).sum() - ce1 * ce1)
    assert np.allclose(out[:], [cb, ce1, ce2], rtol=1e-14)
    # the literals of oracle/mrsim_oracle.c and mr_rl_amd/csrc/mrsim_device.h
    assert np.allclose(out[:], [0.8641431770614779, -0.05097452091652899, 0.05594888714408681], rtol=1e-15)
    # per-step displacement std = dt sigma sqrt(sum b_i^2) = 0.868937 dt sigma (b1 included)
    assert abs(np.sqrt(cb * cb + (35 / 384) ** 2) - 0.868937) < 1e-6


from tests.increments import rhs_mean as _rhs_mean  # noqa: E402


def _run(law, sigma, start, mis, seed=11):
    p = O.default_params(sigma=sigma, mismatched=mis, noise_law=law)
    orc = O.VecOracle(N_ENVS, p, seed=seed, threads=8)
    x0, y0 = STARTS[start]
    orc.reset(0, init_xy=np.tile([[x0, y0]], (N_ENVS, 1)))
    out = {k: np.zeros((N_STEPS, N_ENVS)) for k in ("dx", "dy", "err", "fx", "fy", "spx", "spy")}
    att = np.zeros((N_STEPS, N_ENVS), dtype=np.int32)
    lo, hi = (-20.0, -2 * np.pi), (20.0, 2 * np.pi)
    dt, b1 = p.time_span, 35.0 / 384
    for t in range(N_STEPS):
        a = orc.random_policy(t + 1, lo, hi)     # DYN(0, 0) words 0,1: the same actions under both laws
        prev, f_prev = orc.envs["y"].copy(), orc.envs["f"].copy()
        orc.step(a, step_idx=t + 1)
        # The actions (identical under both laws) dominate every raw quantity; the tests look at what the NOISE did: the
        # increment minus its noise-free single-attempt value dt (b1 K0 + (1 - b1) V) (= dt S_B when the step is not split),
        # and the carried derivative / state_prime minus V
        V = _rhs_mean(a.astype(np.float64), p.a0, mis)
        d = orc.envs["y"] - prev - dt * (b1 * f_prev + (1 - b1) * V)
        out["dx"][t], out["dy"][t] = d[:, 0], d[:, 1]
        out["err"][t] = orc.envs["err_norm0"]
        out["fx"][t], out["fy"][t] = (orc.envs["f"] - V).T
        out["spx"][t], out["spy"][t] = (orc.envs["state_prime"] - V).T
        att[t] = orc.envs["n_attempts"]
    return out, att, orc.envs["y"].copy()


def _pvalues(a, att_a, b, att_b):
    """Two-sample tests on INDEPENDENT samples.  The constructor's draws (carried f, state_prime, minus V) are iid over all
    env steps: pooled.  Increments, error norms and attempt counts of one env are correlated along its trajectory near the
    origin (the scale atol + rtol |y| persists), so those are reduced to one number per env first."""
    ps = {}
    for k in ("fx", "fy", "spx", "spy"):
        ps[k] = stats.ks_2samp(a[k].ravel(), b[k].ravel()).pvalue
    for k in ("dx", "dy"):
        ps[k + "_sum"] = stats.ks_2samp(a[k].sum(axis=0), b[k].sum(axis=0)).pvalue       # noise displacement of the run
        ps[k + "_last"] = stats.ks_2samp(a[k][-1], b[k][-1]).pvalue
    ps["err_first"] = stats.ks_2samp(a["err"][0], b["err"][0]).pvalue                     # same state in every env
    ps["err_last"] = stats.ks_2samp(a["err"][-1], b["err"][-1]).pvalue
    ps["err_gmean"] = stats.ks_2samp(np.log(a["err"]).mean(axis=0), np.log(b["err"]).mean(axis=0)).pvalue
    ps["att_total"] = stats.ks_2samp(att_a.sum(axis=0), att_b.sum(axis=0)).pvalue if att_a.max() > 1 else 1.0
    # attempts of the first step (same state in every env; discrete): chi-square on the histogram
    top = int(max(att_a[0].max(), att_b[0].max()))
    ha = np.bincount(att_a[0], minlength=top + 1).astype(float)
    hb = np.bincount(att_b[0], minlength=top + 1).astype(float)
    keep = (ha + hb) >= 20
    if keep.sum() > 1:
        ps["attempts"] = stats.chi2_contingency(np.vstack([ha[keep], hb[keep]]))[1]
    else:
        ps["attempts"] = 1.0      # (practically) every env took the same number of attempts: fewer than 20 did not
    return ps


@pytest.mark.parametrize("sigma,start,mis", CASES)
def test_collapsed_law_equals_per_stage_law_in_distribution(sigma, start, mis):
    a, att_a, end_a = _run(O.LAW_PER_STAGE, sigma, start, mis)
    b, att_b, end_b = _run(O.LAW_COLLAPSED, sigma, start, mis)
    ps = _pvalues(a, att_a, b, att_b)
    ps_att = ps.pop("attempts")
    worst = min(min(ps.values()), ps_att)
    msg = f"sigma={sigma} start={start} mis={mis}: p = " + ", ".join(f"{k} {v:.3g}" for k, v in ps.items()) + \
          f", attempts {ps_att:.3g}; mean attempts {att_a.mean():.3f} / {att_b.mean():.3f}"
    print(msg)
    assert worst > P_MIN, msg
    # the collapsed law is a different draw layout, not the same numbers
    assert not np.allclose(a["dx"], b["dx"])
    if start == "far" and not mis:
        # far from the origin steps are never split and the increment law is exact: N(0, (dt sigma cB)^2)
        assert att_a.max() == 1 and att_b.max() == 1
        for r in (a, b):
            for k in ("dx", "dy"):
                z = r[k].ravel() / (0.030 * sigma * 0.8641431770614779)
                assert stats.kstest(z[::7], "norm").pvalue > 1e-3
                assert abs(z.std() - 1.0) < 4e-3
    if start == "near":
        assert att_a.mean() > 1.05 and att_b.mean() > 1.05      # the error controller is at work in this regime


# ---------------------------------------------------------------------------------------------------------------------
# sigma > 0 statistics against the REFERENCE's own samples (tests/golden/ref_increments.npz, written by make_golden.py from the
# imported MR_simulator.Simulator; tests/increments.py holds the comparison): increments AND rk_step attempts, far from the
# origin, where the first attempt's error_norm is about 1, and where every step is split into 20-50 attempts -- both noise laws
# ---------------------------------------------------------------------------------------------------------------------
from tests import increments as INC  # noqa: E402

# sample sizes of the oracle's side (envs = independent episodes, steps per episode as in the fixture)
ORACLE_ENVS = {"far": 100000, "mid": 250000, "origin": 100000, "near": 100000}


class _OracleSteps:
    def __init__(self, n, law, mis, seed, start=(110.0, 115.0)):
        p = O.default_params(sigma=1.0, mismatched=int(mis), noise_law=law)
        self.orc = O.VecOracle(n, p, seed=seed, threads=8)
        self.orc.reset(0, init_xy=np.tile([list(start)], (n, 1)))    # a fresh env: nominal-law constructor (MR_env.py:181-183)
        self.t = 0

    def pos(self):
        return self.orc.envs["y"]

    def k0(self):
        return self.orc.envs["f"]

    def attempts(self):
        return self.orc.envs["n_attempts"]

    def step(self, a):
        self.t += 1
        self.orc.step(a, step_idx=self.t)


@pytest.mark.parametrize("law", [O.LAW_PER_STAGE, O.LAW_COLLAPSED], ids=["per_stage", "collapsed"])
@pytest.mark.parametrize("mis", [False, True], ids=["nominal", "mismatched"])
@pytest.mark.parametrize("regime", INC.REGIMES)
def test_oracle_increments_and_attempts_match_the_reference_sample(regime, law, mis):
    ref = INC.reference(regime, mis)
    n, steps = ORACLE_ENVS[regime], int(ref["steps"])
    if regime == "far":
        steps = 20        # iid steps: pooled (the fixture's 100-step restarts are pooled the same way)
    st = _OracleSteps(n, law, mis, seed=21 + 7 * law, start=tuple(ref["start"]))
    resn, att = INC.draw(st, n, steps, mis, seed=5 + law)
    out = INC.compare(regime, mis, resn, att, label=f"oracle law={law}")
    assert out["tol"] <= 0.02


# ---------------------------------------------------------------------------------------------------------------------
# main.py's own noisy run_sim calls (main.py:9-84: noise_var 0.5, a0 1.5, mismatched, from the origin), repeated 4000 times by the
# imported reference (tests/golden/ref_main_runs.npz): where the robot is, in distribution, at five checkpoints of each run
# ---------------------------------------------------------------------------------------------------------------------
class _OracleMain:
    def __init__(self, n, law, seed):
        p = O.default_params(sigma=0.5, a0=1.5, mismatched=1, noise_law=law)
        self.orc = O.VecOracle(n, p, seed=seed, threads=8)
        self.orc.reset(0, init_xy=np.zeros((n, 2)))      # a fresh env, as run_sim builds one (utils.py:46): nominal-law constructor
        self.t = 0

    def pos(self):
        return self.orc.envs["y"]

    def attempts(self):
        return self.orc.envs["n_attempts"]

    def step(self, a):
        self.t += 1
        self.orc.step(a, step_idx=self.t)


@pytest.mark.parametrize("law", [O.LAW_PER_STAGE, O.LAW_COLLAPSED], ids=["per_stage", "collapsed"])
@pytest.mark.parametrize("run", ["idle", "learn"])
def test_oracle_reproduces_the_law_of_main_py_runs(run, law):
    ref = INC.main_runs(run)
    n = 12000
    pos, att = INC.run_main(_OracleMain(n, law, seed=3 + law), ref, n)
    INC.compare_checkpoints(ref, pos, att, label=f"oracle law={law} main.py {run}")
