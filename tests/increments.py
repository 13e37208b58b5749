"""sigma > 0 statistics against the REFERENCE's own samples (tests/golden/ref_increments.npz, schema 2).

The fixture is written by tests/golden/make_golden.py from the imported MR_simulator.Simulator: noise increments and rk_step
attempts of Simulator.step (MR_simulator.py:36-52 under the per-evaluation noise of :73-83) at sigma = 1 for both model laws in
four regimes -- "far" (start (110, 115): the DDPG regime, steps never split, 1e6 env steps), "mid" (start (8, -6): the first
attempt's error_norm is about 1, accept / reject is a coin flip, 1e6), "origin" and "near" (starts (0, 0), (0.5, -0.2): every
step split into 20-50 attempts, 4e5 each).  It holds sorted-sample quantiles at fixed ranks (the reference's ECDF is exact at
those points), moments and attempts histograms -- pooled, per step index and per episode.

This module draws the matching sample from a stepper (the CPU oracle or the HIP kernels, either noise law) and compares:
  * Kolmogorov-Smirnov on the stored ECDF points (a sup over a subset of the two-sample statistic: its p-value is conservative),
    on INDEPENDENT samples only: pooled where steps are iid (far), per step index and per episode sum elsewhere;
  * |std ratio - 1| < 0.005 (far, mid; the reference sample's own standard error is 0.07 % there), and
    < max(0.005, 4 standard errors) where every step is split (the standard error is taken from the per-episode mean squares,
    which is honest under the dependence of one episode's steps);
  * chi-square on the attempts-per-step histograms (per step index), KS on the attempts per episode;
  * means and the x-y correlation.
Test infrastructure only.
"""
import os

import numpy as np
from scipy import stats

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT, B1, CB = 0.030, 35.0 / 384, 0.8641431770614779
REGIMES = ("far", "mid", "origin", "near")
# family-wise 1 % over every p-value one noise law produces (2 model laws x (3 + 3 regimes x 27))
P_MIN = 0.01 / 170

_cache = {}


def reference(regime, mis):
    key = f"{regime}_{'mismatched' if mis else 'nominal'}_s1"
    if "npz" not in _cache:
        _cache["npz"] = np.load(os.path.join(GOLDEN, "ref_increments.npz"))
        assert int(_cache["npz"]["schema"]) == 2
    g = _cache["npz"]
    return {k[len(key) + 1:]: g[k] for k in g.files if k.startswith(key + "/")}


def rhs_mean(a, a0, mis):
    """noise-free part of Simulator.simulate (MR_simulator.py:76-83) for actions [n,2]"""
    f, al = a[:, 0], a[:, 1]
    if mis:
        a0b = a0 + (f / 4) * 0.8
        return np.stack([a0b * f * np.cos(al + 0.1) + 0.2, a0b * f * np.sin(al - 0.15) - 0.1], axis=1)
    return np.stack([a0 * f * np.cos(al), a0 * f * np.sin(al)], axis=1)


def normalised_residual(p0, p1, k0, a, mis, sigma=1.0, a0=1.0):
    """(Delta - dt (b1 K0 + (1 - b1) V)) / (dt cB sqrt(g^2 + sigma^2)), the twin of make_golden.py: inc_normalised_residual"""
    a64 = a.astype(np.float64)
    V = rhs_mean(a64, a0, mis)
    if mis:
        g = 0.25 * sigma * a64[:, :1] * np.stack([np.cos(a64[:, 1] + 0.1), np.sin(a64[:, 1] - 0.15)], 1)
    else:
        g = np.zeros_like(V)
    return (p1 - p0 - DT * (B1 * k0 + (1 - B1) * V)) / (DT * CB * np.sqrt(g * g + sigma * sigma))


def draw(stepper, n_envs, steps, mis, seed):
    """resn [n_envs, steps, 2], att [n_envs, steps] of `stepper`: an object with .pos() -> [n,2] fp64, .k0() -> [n,2] (the carried
    integrator.f), .attempts() -> [n] and .step(actions [n,2] float32); actions uniform in the DDPG actor range
    (RL/MR_ddpg.py:136-137,345), as in the fixture."""
    rng = np.random.default_rng(seed)
    resn = np.zeros((n_envs, steps, 2))
    att = np.zeros((n_envs, steps), dtype=np.int64)
    for k in range(steps):
        a = np.stack([rng.uniform(-20, 20, n_envs), rng.uniform(-2 * np.pi, 2 * np.pi, n_envs)], 1).astype(np.float32)
        p0, k0 = stepper.pos().copy(), stepper.k0().copy()
        stepper.step(a)
        resn[:, k] = normalised_residual(p0, stepper.pos(), k0, a, mis)
        att[:, k] = stepper.attempts()
    return resn, att


def ks_vs_quantiles(x, q, ranks, n_ref):
    """sup over the stored points of |F_x - F_ref|, and the asymptotic two-sample p-value for it"""
    xs = np.sort(np.asarray(x, dtype=np.float64))
    q = np.asarray(q, dtype=np.float64)
    f_ref_hi = (np.asarray(ranks, dtype=np.float64) + 1.0) / n_ref      # F_ref at q (right-continuous)
    f_ref_lo = np.asarray(ranks, dtype=np.float64) / n_ref              # F_ref just below q
    f_hi = np.searchsorted(xs, q, side="right") / len(xs)
    f_lo = np.searchsorted(xs, q, side="left") / len(xs)
    d = max(np.abs(f_hi - f_ref_hi).max(), np.abs(f_lo - f_ref_lo).max())
    en = np.sqrt(len(xs) * float(n_ref) / (len(xs) + float(n_ref)))
    return d, float(stats.kstwobign.sf(d * en))


def chi2_hist(ha, hb, min_count=20):
    """two-sample chi-square of two histograms over the same bins; sparse bins are merged into their neighbour from the top"""
    ha, hb = np.asarray(ha, dtype=np.float64), np.asarray(hb, dtype=np.float64)
    keep_a, keep_b, acc_a, acc_b = [], [], 0.0, 0.0
    for x, y in zip(ha[::-1], hb[::-1]):
        acc_a += x; acc_b += y
        if acc_a + acc_b >= min_count:
            keep_a.append(acc_a); keep_b.append(acc_b); acc_a = acc_b = 0.0
    if acc_a + acc_b > 0 and keep_a:
        keep_a[-1] += acc_a; keep_b[-1] += acc_b
    if len(keep_a) < 2:
        return 1.0
    return float(stats.chi2_contingency(np.vstack([keep_a, keep_b]))[1])


def ks_hist(ha, hb):
    """KS of two integer-valued samples given as histograms over the same bins"""
    ha, hb = np.asarray(ha, dtype=np.float64), np.asarray(hb, dtype=np.float64)
    na, nb = ha.sum(), hb.sum()
    d = np.abs(np.cumsum(ha) / na - np.cumsum(hb) / nb).max()
    return float(stats.kstwobign.sf(d * np.sqrt(na * nb / (na + nb))))


def compare(regime, mis, resn, att, label=""):
    """Assert that the sample (resn [E,S,2], att [E,S]) has the law of the reference's; returns a dict of what was measured."""
    ref = reference(regime, mis)
    E, S, _ = resn.shape
    per_step = regime != "far"
    assert (not per_step) or S == int(ref["steps"])      # far: iid steps, any grouping into "episodes" will do
    pooled = resn.reshape(-1, 2)
    out = {"p": {}, "std_ratio": [], "tol": 0.005}
    tag = f"{label} {regime} mis={int(mis)}"
    # ---- standard deviation (and mean) of the pooled increments
    ssq = (resn ** 2).mean(axis=1)                                   # [E,2] per-episode mean squares
    for j in range(2):
        # ratio of root mean squares about the reference's pooled mean (the means agree to ~1e-3 of a std: checked below)
        ref_ms = float(ref["episode/sumsq_mean"][j])
        ratio = np.sqrt(ssq[:, j].mean() / ref_ms)
        se = 0.5 * np.sqrt(float(ref["episode/sumsq_var"][j]) / (int(ref["episode/n"]) * ref_ms ** 2) +
                           ssq[:, j].var() / (E * ssq[:, j].mean() ** 2))
        tol = 0.005 if regime in ("far", "mid") else max(0.005, 4.0 * se)
        out["std_ratio"].append((float(ratio), float(se)))
        out["tol"] = max(out["tol"], tol)
        assert abs(ratio - 1.0) < tol, (tag, "std ratio", j, ratio, se)
        mu_se = np.sqrt(float(ref["pooled/var"][j]) * (1.0 / int(ref["pooled/n"]) + 1.0 / len(pooled)))
        # (means of one episode's steps are dependent where steps are split: 8 x the iid standard error there)
        assert abs(pooled[:, j].mean() - float(ref["pooled/mean"][j])) < (5.0 if regime == "far" else 15.0) * mu_se, (tag, "mean", j)
    rho_ref = float(ref["pooled/cov_xy"]) / np.sqrt(float(ref["pooled/var"][0]) * float(ref["pooled/var"][1]))
    rho = np.corrcoef(pooled[:, 0], pooled[:, 1])[0, 1]
    assert abs(rho - rho_ref) < 0.01, (tag, "x-y correlation", rho, rho_ref)
    # ---- distributions, on independent samples
    if not per_step:
        for j in range(2):
            out["p"][f"ks_pooled_{j}"] = ks_vs_quantiles(pooled[:, j], ref["pooled/q"][j], ref["pooled/ranks"], int(ref["pooled/n"]))[1]
        hist = np.bincount(np.minimum(att.ravel(), len(ref["pooled/att_hist"]) - 1), minlength=len(ref["pooled/att_hist"]))
        out["p"]["att_pooled"] = chi2_hist(hist, ref["pooled/att_hist"])
    else:
        nb = ref["step/att_hist"].shape[1]
        for k in range(S):
            for j in range(2):
                out["p"][f"ks_step{k}_{j}"] = ks_vs_quantiles(resn[:, k, j], ref["step/q"][k, j], ref["step/ranks"][k], int(ref["step/n"][k]))[1]
            out["p"][f"att_step{k}"] = chi2_hist(np.bincount(np.minimum(att[:, k], nb - 1), minlength=nb), ref["step/att_hist"][k])
        esum = resn.sum(axis=1)
        for j in range(2):
            out["p"][f"ks_episode_sum_{j}"] = ks_vs_quantiles(esum[:, j], ref["episode/sum_q"][j], ref["episode/sum_ranks"], int(ref["episode/sum_n"]))[1]
        nt = len(ref["episode/att_total_hist"])
        out["p"]["att_episode_total"] = ks_hist(np.bincount(np.minimum(att.sum(axis=1), nt - 1), minlength=nt), ref["episode/att_total_hist"])
    worst = min(out["p"], key=out["p"].get)
    out["mean_attempts"] = (float(att.mean()), float((ref["pooled/att_hist"] * np.arange(len(ref["pooled/att_hist"]))).sum() / ref["pooled/att_hist"].sum()))
    print(f"{tag}: std ratio " + ", ".join(f"{r:.4f} (se {s:.4f})" for r, s in out["std_ratio"]) +
          f"; mean attempts {out['mean_attempts'][0]:.3f} vs reference {out['mean_attempts'][1]:.3f}; "
          f"{len(out['p'])} tests, smallest p {out['p'][worst]:.3g} ({worst})")
    assert out["p"][worst] > P_MIN, (tag, worst, out["p"][worst])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# main.py's own noisy run_sim calls (tests/golden/ref_main_runs.npz): distributions of the position at checkpoints of the run
# ---------------------------------------------------------------------------------------------------------------------
def main_runs(name):
    """{actions [T,2] float32-valued, checkpoints [C], n, ranks, q [C,2,Q], mean [C,2], var [C,2], cov_xy [C], att_mean [C], att_var [C]}
    of one of main.py's runs ("idle", "learn") as the imported reference produced it 4000 times (make_golden.py: gen_main_runs)"""
    if "main" not in _cache:
        _cache["main"] = np.load(os.path.join(GOLDEN, "ref_main_runs.npz"))
        assert int(_cache["main"]["schema"]) == 1
    g = _cache["main"]
    return {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "/")}


def run_main(stepper, ref, n_envs):
    """drive `stepper` (all envs at the origin, main.py's parameters) through the run's action table, the same action for every env;
    -> pos [n, C, 2], cumulative attempts [n, C] at the checkpoints"""
    acts, cps = ref["actions"].astype(np.float32), [int(c) for c in ref["checkpoints"]]
    pos = np.zeros((n_envs, len(cps), 2))
    att = np.zeros((n_envs, len(cps)))
    total, ci = np.zeros(n_envs), 0
    for k in range(cps[-1]):
        stepper.step(np.tile(acts[k][None, :], (n_envs, 1)))
        total += stepper.attempts()
        if k + 1 == cps[ci]:
            pos[:, ci], att[:, ci] = stepper.pos(), total
            ci += 1
    return pos, att


def compare_checkpoints(ref, pos, att, label=""):
    """the run's position law at every checkpoint against the reference's 4000 runs: KS on the stored ECDF points per component,
    std ratio within 5 % (the reference sample's own standard error: 1.1 %), means within 5 standard errors, x-y correlation within
    0.06, mean cumulative attempts within 5 standard errors"""
    n_ref, C = int(ref["n"]), len(ref["checkpoints"])
    ps, worst_ratio = {}, 0.0
    for c in range(C):
        for j in range(2):
            x = pos[:, c, j]
            ps[f"ks_cp{c}_{j}"] = ks_vs_quantiles(x, ref["q"][c, j], ref["ranks"], n_ref)[1]
            sd_ref = float(np.sqrt(ref["var"][c, j]))
            ratio = x.std() / sd_ref
            worst_ratio = max(worst_ratio, abs(ratio - 1.0))
            assert abs(ratio - 1.0) < 0.05, (label, "std ratio", c, j, ratio)
            assert abs(x.mean() - float(ref["mean"][c, j])) < 5.0 * sd_ref * np.sqrt(1.0 / n_ref + 1.0 / len(x)), (label, "mean", c, j)
        rho_ref = float(ref["cov_xy"][c]) / np.sqrt(float(ref["var"][c, 0]) * float(ref["var"][c, 1]))
        assert abs(np.corrcoef(pos[:, c, 0], pos[:, c, 1])[0, 1] - rho_ref) < 0.06, (label, "correlation", c)
        se = np.sqrt(float(ref["att_var"][c]) / n_ref + att[:, c].var() / len(att))
        assert abs(att[:, c].mean() - float(ref["att_mean"][c])) < 5.0 * se + 1e-9, (label, "attempts", c, att[:, c].mean(), float(ref["att_mean"][c]))
    worst = min(ps, key=ps.get)
    print(f"{label}: {len(ps)} KS tests, smallest p {ps[worst]:.3g} ({worst}); worst |std ratio - 1| {worst_ratio:.4f}; attempts per step at the end "
          f"{att[:, -1].mean() / int(ref['checkpoints'][-1]):.2f} vs reference {float(ref['att_mean'][-1]) / int(ref['checkpoints'][-1]):.2f}")
    assert ps[worst] > 0.01 / (4 * len(ps)), (label, worst, ps[worst])       # family-wise 1 % over runs x laws
