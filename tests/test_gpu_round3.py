"""GPU tests added in round 3 (all through the C ABI):
  * the auto-reset as "the same env object, re-used" (MR_env.py:181-183) against golden episodes of ONE reference MR_Env;
  * MR_Env.set_save_experice wired to the recorder: the facade driven exactly as the reference was for
    tests/golden/ref_experiment.npz reproduces MRExperiment.__dict__ and writes the reference's files;
  * RolloutCollector.reset() right behind collect() (no join by the caller);
  * state_dict carries the outputs a gym loop reads before its next action; MR_Env(seed=s).reset() reproducible;
  * BASELINE config 5 as far as one GPU goes: the eight 262 144-env shards one after another == one unsharded
    2 097 152-env run (returns in the gatherer's [world, E, n_local] layout, sampled transitions), bitwise.
"""
import os

import numpy as np
import pytest
import torch

from tests.util import load_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POS_TOL = 1e-6
REUSED = load_cases("ref_reused.npz")


def _env(n, seed=0, goal_table=None, env_id0=0, **cfg_kw):
    from mr_rl_amd import MRConfig, MRVecEnv
    return MRVecEnv(n, cfg=MRConfig(**cfg_kw), seed=seed, goal_table=goal_table, env_id0=env_id0)


# ----------------------------------------------------------------------------------------------------------------------
# the fused rollout's reset cache (goal-table launches) against the in-step reset and the oracle
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("carry", ["f32", "f64"])
@pytest.mark.parametrize("mis", [False, True])
def test_reset_cache_with_episodes_of_a_few_steps(mis, carry):
    """The flag-specialised goal-table rollout keeps each lane's NEXT reset in an LDS slot and refills the empty slots of a
    wave together (mrsim_kernels.hip).  Hard case: goals inside the init box and a goal radius of 6, so that episodes last 1-10
    steps -- lanes terminate several times between two refills, in consecutive steps, right after a launch starts, and
    whole waves at once.  Three launches of 40 steps against (a) 120 single steps (the in-step reset: same bits, every
    transition, the carried state, returns, lengths) and (b) the oracle (noise_math='spec': positions to POS_TOL, the same
    dones, lengths), both laws; the fp64 carry against the oracle alone (tools/ab_rollout.py compares it bitwise with a build
    without the cache: profiles/r03/ab_reset_cache_mixed.txt)."""
    from oracle import oracle as O
    from tests.util import orc_params_from_cfg
    from mr_rl_amd import MRConfig, MRVecEnv
    n, T, L = 1500, 40, 3
    tab = np.random.default_rng(3).uniform(104, 116, (3, 52, 2)).astype(np.float32)
    cfg = lambda: MRConfig(noise_var=1.0, auto_reset=True, is_mismatched=mis, noise_math="spec", reward_mode="goal",  # noqa: E731
                           min_dist2goal=6.0, rollout_carry=carry, seed=13)
    e1 = MRVecEnv(n, cfg=cfg(), seed=13, env_id0=77, goal_table=tab)
    e2 = MRVecEnv(n, cfg=cfg(), seed=13, env_id0=77, goal_table=tab)
    orc = O.VecOracle(n, orc_params_from_cfg(cfg(), 3, 52), seed=13, env_id0=77, goal_table=tab, threads=8)
    e1.reset(); e2.reset(); orc.reset(0)
    lens = []
    t_abs = 0
    for launch in range(L):
        out = e1.rollout(T, actions=None, want=("obs", "rew", "done", "actions"))
        obs, rew, done, act = (out[q].cpu().numpy() for q in ("obs", "rew", "done", "actions"))
        for t in range(T):
            t_abs += 1
            a = orc.random_policy(t_abs, cfg().policy_low, cfg().policy_high)
            np.testing.assert_array_equal(act[t], a)
            orc.step(a, step_idx=t_abs)
            np.testing.assert_array_equal(done[t].astype(np.uint8), orc.done, err_msg=f"launch {launch} step {t}")
            np.testing.assert_allclose(obs[t][:, :2], orc.obs[:, :2], rtol=0, atol=2e-5)
            if carry == "f32":
                o2, r2, d2, _ = e2.step(None)
                np.testing.assert_array_equal(obs[t].view(np.uint32), o2.cpu().numpy().view(np.uint32))
                np.testing.assert_array_equal(rew[t], r2.cpu().numpy())
                np.testing.assert_array_equal(done[t], d2.cpu().numpy())
            lens.append(orc.final_len[orc.done.astype(bool)].copy())
        np.testing.assert_allclose(e1.pos.cpu().numpy(), orc.envs["y"], rtol=0, atol=POS_TOL)
    if carry == "f32":
        np.testing.assert_array_equal(e1.pos.cpu().numpy().view(np.uint64), e2.pos.cpu().numpy().view(np.uint64))
        np.testing.assert_array_equal(e1.aux.cpu().numpy().view(np.uint32), e2.aux.cpu().numpy().view(np.uint32))
        np.testing.assert_array_equal(e1.final_ret.cpu().numpy(), e2.final_ret.cpu().numpy())
        np.testing.assert_array_equal(e1.final_len.cpu().numpy(), e2.final_len.cpu().numpy())
    np.testing.assert_array_equal(e1.final_len.cpu().numpy(), orc.final_len)
    lens = np.concatenate(lens)
    assert len(lens) > 12 * n and np.median(lens) <= 6 and (lens == 1).sum() > 3 * n   # many resets, many one-step episodes
    e1.check_status()


@pytest.mark.parametrize("case", range(10))
def test_reset_cache_randomised_tables_and_sizes(case):
    """The goal-table rollout (reset cache, hoisted row pointer, v_med3 row clamp) on random shapes: K trajectories of T_tab rows
    (rows shorter and longer than an episode: the row index clamps), episode limits 6-50, goal radius 2-7, ragged env counts,
    env-id offsets, both laws, launches of random lengths -- every transition bitwise against single steps, dones / lengths /
    reset rows against the oracle."""
    from oracle import oracle as O
    from tests.util import orc_params_from_cfg
    from mr_rl_amd import MRConfig, MRVecEnv
    r = np.random.default_rng(1000 + case)
    n = int(r.choice([1, 63, 64, 65, 257, 777, 1500]))
    K, Ttab = int(r.choice([1, 2, 3, 5])), int(r.choice([4, 20, 52, 90]))
    maxT = int(r.choice([6, 17, 50]))
    mis = bool(r.integers(2))
    id0 = int(r.choice([0, 12345, 2**31 + 7]))
    tab = r.uniform(103, 117, (K, Ttab, 2)).astype(np.float32)
    kw = dict(noise_var=float(r.choice([0.5, 1.0, 2.0])), auto_reset=True, is_mismatched=mis, noise_math="spec", reward_mode="goal",
              min_dist2goal=float(r.uniform(2, 7)), max_timesteps=maxT, seed=50 + case)
    e1 = MRVecEnv(n, cfg=MRConfig(**kw), seed=50 + case, env_id0=id0, goal_table=tab)
    e2 = MRVecEnv(n, cfg=MRConfig(**kw), seed=50 + case, env_id0=id0, goal_table=tab)
    orc = O.VecOracle(n, orc_params_from_cfg(MRConfig(**kw), K, Ttab), seed=50 + case, env_id0=id0, goal_table=tab, threads=8)
    e1.reset(); e2.reset(); orc.reset(0)
    k, ndone = 0, 0
    for T in [int(x) for x in r.integers(1, 45, size=4)]:
        out = e1.rollout(T, actions=None, want=("obs", "rew", "done", "actions"))
        obs, rew, done = (out[q].cpu().numpy() for q in ("obs", "rew", "done"))
        for t in range(T):
            k += 1
            orc.step(orc.random_policy(k, kw.get("policy_low", MRConfig().policy_low), MRConfig().policy_high), step_idx=k)
            o2, r2, d2, _ = e2.step(None)
            np.testing.assert_array_equal(obs[t].view(np.uint32), o2.cpu().numpy().view(np.uint32), err_msg=f"T={T} t={t}")
            np.testing.assert_array_equal(rew[t], r2.cpu().numpy())
            np.testing.assert_array_equal(done[t], d2.cpu().numpy())
            np.testing.assert_array_equal(done[t].astype(np.uint8), orc.done)
            np.testing.assert_allclose(obs[t], orc.obs, rtol=0, atol=3e-5)
            ndone += int(orc.done.sum())
    np.testing.assert_array_equal(e1.pos.cpu().numpy().view(np.uint64), e2.pos.cpu().numpy().view(np.uint64))
    np.testing.assert_array_equal(e1.aux.cpu().numpy().view(np.uint32), e2.aux.cpu().numpy().view(np.uint32))
    np.testing.assert_array_equal(e1.final_len.cpu().numpy(), e2.final_len.cpu().numpy())
    np.testing.assert_array_equal(e1.final_len.cpu().numpy(), orc.final_len)
    np.testing.assert_allclose(e1.pos.cpu().numpy(), orc.envs["y"], rtol=0, atol=POS_TOL)
    assert ndone > 0 or n == 1
    e1.check_status()


def test_auto_reset_key_wraps_when_the_step_index_is_set_back():
    """The draws of an auto-reset sit at step - (length - 1) in 64-bit modular arithmetic (reset_rng / orc_env_step).  Setting
    the step index back in the middle of the episodes makes that difference negative: kernel (borrow across the two 32-bit
    words, through the reset cache and through single steps) and oracle must wrap the same way."""
    from oracle import oracle as O
    from tests.util import orc_params_from_cfg
    from mr_rl_amd import MRConfig, MRVecEnv
    n = 700
    tab = np.random.default_rng(5).uniform(104, 116, (3, 52, 2)).astype(np.float32)
    cfg = lambda: MRConfig(noise_var=1.0, auto_reset=True, noise_math="spec", reward_mode="goal", min_dist2goal=4.0, seed=3)  # noqa: E731
    e1 = MRVecEnv(n, cfg=cfg(), seed=3, goal_table=tab)
    e2 = MRVecEnv(n, cfg=cfg(), seed=3, goal_table=tab)
    orc = O.VecOracle(n, orc_params_from_cfg(cfg(), 3, 52), seed=3, goal_table=tab, threads=8)
    e1.reset(); e2.reset(); orc.reset(0)
    e1.step_idx = e2.step_idx = k = 2**32 - 3          # the low word is about to carry
    ndone = 0
    for launch, T in enumerate((9, 30, 30)):
        if launch == 1:
            e1.step_idx = e2.step_idx = k = 2            # back to the start: step - (length - 1) < 0 for every running episode
        out = e1.rollout(T, actions=None, want=("obs", "rew", "done", "actions"))
        obs, done = out["obs"].cpu().numpy(), out["done"].cpu().numpy()
        for t in range(T):
            a = orc.random_policy(k, cfg().policy_low, cfg().policy_high)
            orc.step(a, step_idx=k)
            k += 1
            o2, r2, d2, _ = e2.step(None)
            np.testing.assert_array_equal(done[t].astype(np.uint8), orc.done)
            np.testing.assert_array_equal(obs[t].view(np.uint32), o2.cpu().numpy().view(np.uint32))
            np.testing.assert_allclose(obs[t][:, :2], orc.obs[:, :2], rtol=0, atol=2e-5)   # the reset rows: same start positions
            ndone += int(orc.done.sum())
    assert ndone > 3 * n
    np.testing.assert_array_equal(e1.pos.cpu().numpy().view(np.uint64), e2.pos.cpu().numpy().view(np.uint64))
    np.testing.assert_allclose(e1.pos.cpu().numpy(), orc.envs["y"], rtol=0, atol=POS_TOL)
    e1.check_status()


@pytest.mark.parametrize("math", ["fast", "spec"])
def test_noise_increments_are_gaussian_at_full_size(math):
    """SURVEY 3.3 at BASELINE config 4's size: away from the origin one env step adds, per axis, a Gaussian of standard deviation
    dt sigma sqrt(sum B_i^2) = 0.868938 dt sigma around dt v (the five stage noises of the Dormand-Prince combination; the stale
    first stage carries the previous constructor's draw).  262 144 envs x 3 steps through the fused rollout, default
    (hardware-transcendental) and specified Box-Muller: Kolmogorov-Smirnov against N(0, 1) with the THEORETICAL scale, tail
    mass beyond 3 and 4 sigma, kurtosis, and no correlation between axes, consecutive steps or neighbouring envs."""
    import scipy.stats as st
    from mr_rl_amd import MRConfig, MRVecEnv
    n, T = 262144, 4
    env = MRVecEnv(n, cfg=MRConfig(noise_var=1.0, noise_math=math, rollout_carry="f64"), seed=101)
    env.reset(init=np.tile([[110.0, 115.0]], (n, 1)))
    acts = np.tile(np.array([[4.0, 0.5]], dtype=np.float32), (T, 1))
    traj = env.rollout(T, actions=acts, shared_actions=True, want=("traj",))["traj"].cpu().numpy()   # [T, n, 2]
    v = 4.0 * np.array([np.cos(np.float64(np.float32(0.5))), np.sin(np.float64(np.float32(0.5)))])
    sd = np.sqrt((35 / 384) ** 2 + (500 / 1113) ** 2 + (125 / 192) ** 2 + (2187 / 6784) ** 2 + (11 / 84) ** 2) * 0.03
    z = (np.diff(traj, axis=0) - 0.03 * v) / sd                      # steps 2..4, standardised with the theoretical scale
    assert z.shape == (T - 1, n, 2)
    for k in range(T - 1):
        for ax in range(2):
            x = z[k, :, ax]
            assert st.kstest(x, "norm").pvalue > 1e-4, (k, ax, st.kstest(x, "norm"))
            assert abs(x.mean()) < 4.5 / np.sqrt(n) and abs(x.std() - 1) < 4.5 / np.sqrt(2 * n)
            assert abs(st.kurtosis(x)) < 4.5 * np.sqrt(24 / n)
            for c, p in ((3.0, 2 * st.norm.sf(3.0)), (4.0, 2 * st.norm.sf(4.0))):
                got = (np.abs(x) > c).mean()
                assert abs(got - p) < 4.5 * np.sqrt(p / n), (c, got, p)
    lim = 4.5 / np.sqrt(n)
    for k in range(T - 1):
        assert abs(np.corrcoef(z[k, :, 0], z[k, :, 1])[0, 1]) < lim                        # the two axes
        assert abs(np.corrcoef(z[k, :-1, 0], z[k, 1:, 0])[0, 1]) < lim                     # neighbouring envs
    for k in range(T - 2):
        assert abs(np.corrcoef(z[k, :, 0], z[k + 1, :, 0])[0, 1]) < lim                    # consecutive steps
        assert abs(np.corrcoef(z[k, :, 1], z[k + 1, :, 1])[0, 1]) < lim
    env.check_status()


# ----------------------------------------------------------------------------------------------------------------------
# auto-reset = reset() on the SAME env object (RL/MR_ddpg.py:270; MR_env.py:181-183)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [k for k in sorted(REUSED) if float(REUSED[k]["sigma"]) == 0.0])
@pytest.mark.parametrize("path", ["rollout_f64", "rollout_f32", "steps"])
def test_auto_reset_replays_a_reused_reference_env(name, path):
    """tests/golden/ref_reused.npz: ONE reference MR_Env, three episodes, reset(is_mismatched=...) at the top of each.
    The vec env's init box is the single start point, so its in-kernel auto-reset replays the reference's resets: from the
    second episode on the RK45 object is built under the law the previous episode left behind -- under is_mismatched the
    stale first stage is the drift (0.2, -0.1).  Positions to POS_TOL through the fused rollout (both carries) and through
    single steps; auto_reset_env="fresh" (a new MR_Env per episode) is a measurably different trajectory."""
    G = REUSED[name]
    init = tuple(float(v) for v in G["init"])
    kw = dict(noise_var=0.0, a0=float(G["a0"]), is_mismatched=bool(G["mismatched"]), auto_reset=True, init_low=init,
              init_high=init)
    acts = torch.from_numpy(np.ascontiguousarray(G["actions"], dtype=np.float32)).cuda()
    T = len(acts)

    def run(**extra):
        env = _env(4, **kw, **extra)
        env.reset()      # a fresh env: nominal-law constructor for the first episode
        if path == "steps":
            pos, done = [], []
            for t in range(T):
                _, _, d, info = env.step(acts[t].expand(4, 2).contiguous())
                pos.append(torch.where(d[:, None], info["final_obs"][:, :2].double(), env.pos).cpu().numpy())
                done.append(d.cpu().numpy())
            env.check_status()
            return np.stack(pos)[:, 1], np.stack(done)[:, 1]
        out = env.rollout(T, actions=acts, shared_actions=True, want=("traj", "done"), carry=path[-3:])
        env.check_status()
        return out["traj"][:, 1].cpu().numpy(), out["done"][:, 1].cpu().numpy()

    pos, done = run()
    assert np.array_equal(done.astype(np.uint8), G["done"].astype(np.uint8))
    tol = POS_TOL if path != "steps" else 2e-5   # the step path returns the terminal position as a float32 observation
    err = np.abs(pos - G["pos"])
    assert err[G["done"] == 0].max() < POS_TOL and err.max() < tol, err.max()
    if bool(G["mismatched"]):
        pos_fresh, _ = run(auto_reset_env="fresh")
        d = np.abs(pos_fresh - G["pos"])
        assert d[:50].max() < POS_TOL and d[51:].max() > 1e-4     # identical first episode, different ones after it


# ----------------------------------------------------------------------------------------------------------------------
# MR_Env.set_save_experice -> recorder (MR_env.py:94-95,145-147,190-198,223-226)
# ----------------------------------------------------------------------------------------------------------------------
def test_facade_records_like_the_reference(tmp_path, monkeypatch):
    """The facade driven exactly as tests/golden/make_golden.py: gen_experiment drove the reference (set_save_experice,
    then per episode reset(init) and step until done): MR_data holds MRExperiment.__dict__'s content -- every key, shape,
    dtype and value -- and the reference's save triggers wrote ./_experiments/<date-hour><name>."""
    from mr_rl_amd import MR_Env, recorder
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_experiment.npz"))
    n_ep = int(g["iterations"]) + 1
    monkeypatch.chdir(tmp_path)
    env = MR_Env()
    env.set_save_experice("golden")
    for k in range(n_ep):
        env.reset(init=g[f"in/{k}/init"], noise_var=0.0, a0=1.0, is_mismatched=False)
        for a in g[f"in/{k}/actions"]:
            _, _, d, _ = env.step(a)
            if d:
                break
        assert d
    dct = env.MR_data.to_dict()
    assert sorted(dct.keys()) == [str(k) for k in g["keys"]]
    assert dct["iterations"] == int(g["iterations"]) and dct["time_step"] == int(g["time_step"])
    for it in range(n_ep):
        assert dct["steps"][it] == int(g[f"steps/{it}"])
        for key in ("states", "observations", "actions", "rewards"):
            want, got = g[f"{key}/{it}"], dct[key][it]
            assert got.shape == want.shape and got.dtype == want.dtype, (key, it, got.shape, got.dtype, want.dtype)
        np.testing.assert_allclose(dct["states"][it], g[f"states/{it}"], rtol=0, atol=POS_TOL)
        want_obs = g[f"observations/{it}"]
        tol = 2 * np.spacing(np.abs(want_obs).astype(np.float32)).astype(np.float64) + POS_TOL
        assert np.all(np.abs(dct["observations"][it] - want_obs) <= tol)
        np.testing.assert_array_equal(dct["actions"][it], g[f"actions/{it}"])
        np.testing.assert_array_equal(dct["rewards"][it], g[f"rewards/{it}"])
    # save triggers: reset() with iterations > 0 (MR_env.py:190-192) and end() on the step limit (:145-147)
    files = sorted(os.listdir(tmp_path / "_experiments"))
    assert len(files) == 1 and files[0].endswith("golden")
    saved = recorder.load_experiment(tmp_path / "_experiments" / files[0])
    assert saved["iterations"] == dct["iterations"]
    # end() saves BEFORE step() records the terminal transition (MR_env.py:88 then :94-95): the file lacks that one row
    np.testing.assert_array_equal(saved["states"][n_ep - 1], dct["states"][n_ep - 1][:-1])
    np.testing.assert_array_equal(saved["states"][0], dct["states"][0])
    # without the hook nothing is recorded and nothing is written
    plain = MR_Env()
    plain.reset(init=g["in/0/init"], noise_var=0.0)
    plain.step(g["in/0/actions"][0])
    assert plain.MR_data is None


def test_facade_seed_makes_reset_reproducible():
    from mr_rl_amd import MR_Env
    a, b, c = MR_Env(seed=5), MR_Env(seed=5), MR_Env(seed=6)
    oa, ob, oc = a.reset(noise_var=0.0), b.reset(noise_var=0.0), c.reset(noise_var=0.0)
    assert np.array_equal(oa, ob) and not np.array_equal(oa, oc)
    assert 100 <= oa[0] <= 120 and 100 <= oa[1] <= 120
    a.seed(9); b.seed(9)                                  # old/MR_dqn_keras_rl.py:19 calls env.seed explicitly: still works
    assert np.array_equal(a.reset(noise_var=0.0), b.reset(noise_var=0.0))


# ----------------------------------------------------------------------------------------------------------------------
# collector.reset() right behind collect(); checkpoint of the step outputs
# ----------------------------------------------------------------------------------------------------------------------
def test_collector_reset_right_after_collect_waits_for_the_sub_shard_streams():
    """reset() launches the reset kernel on the current stream; the sub-shard chains of the previous collect() may still be
    writing the state.  With the join inside reset() the episode after it equals the one of a single-stream env that was
    synchronised by hand."""
    from mr_rl_amd import MRConfig
    from mr_rl_amd.collector import RolloutCollector
    n, seed = 200000, 3
    col = RolloutCollector(n, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=seed, streams=3)
    ref = RolloutCollector(n, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=seed, streams=1)
    for c in (col, ref):
        c.reset()
    for rep in range(3):
        col.collect(); col.collect()
        col.reset()                      # no join() by the caller
        col.collect()
        got = {k: v.clone() for k, v in col.ready().items()}
        ref.collect(); ref.collect(); ref.join(); torch.cuda.synchronize()
        ref.reset()
        ref.collect()
        want = ref.ready()
        for key in ("obs", "rew", "done", "actions"):
            assert torch.equal(got[key], want[key]), (rep, key)
        col.join(); ref.join()
        assert torch.equal(col.env.pos, ref.env.pos)
    col.check_status()


def test_collectors_share_one_set_of_sub_shard_streams():
    """Every collector of a process launches its sub-shards on the same HIP streams (collector.py: the pool streams torch hands
    to a second collector did not run their kernels side by side); results do not depend on it: two collectors used in turn
    reproduce one collector's episodes."""
    from mr_rl_amd import MRConfig
    from mr_rl_amd.collector import RolloutCollector
    mk = lambda: RolloutCollector(3000, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=4, streams=2)  # noqa: E731
    a, b, ref = mk(), mk(), mk()
    assert all(x is y for x, y in zip(a.streams, b.streams)) and len(a.streams) == 2 and a.streams[0] is not a.streams[1]
    a.reset(); b.reset(); ref.reset()
    for k in range(4):
        ra = {q: v.clone() for q, v in a.ready(a.collect()).items() if torch.is_tensor(v)}; a.release()
        rb = {q: v.clone() for q, v in b.ready(b.collect()).items() if torch.is_tensor(v)}; b.release()
        rr = ref.ready(ref.collect())
        for q in ("obs", "rew", "done", "actions"):
            assert torch.equal(ra[q], rr[q]) and torch.equal(rb[q], rr[q]), (k, q)
        ref.release()
    a.join(); b.join(); ref.join()
    a.check_status(); b.check_status()


def test_state_dict_carries_the_step_outputs_and_leaves_the_callers_cfg_alone():
    from mr_rl_amd import MRConfig, MRVecEnv
    a = _env(1000, seed=4, noise_var=1.0, auto_reset=True, max_timesteps=5)
    a.reset()
    for _ in range(8):
        obs, rew, done, info = a.step(None)
    sd = a.state_dict()
    cfg_b = MRConfig(noise_var=0.25, auto_reset=True, max_timesteps=5)
    b = MRVecEnv(1000, cfg=cfg_b, seed=99)
    b.load_state_dict(sd)
    assert cfg_b.noise_var == 0.25 and b.cfg.noise_var == 1.0      # the caller's object is not written to
    assert torch.equal(b.obs, a.obs) and torch.equal(b.rew, a.rew) and torch.equal(b.done, a.done)
    assert torch.equal(b.final_obs, a.final_obs)
    for _ in range(6):                                           # a gym loop resumes from the restored observation
        oa, ra, da, _ = a.step(None)
        ob, rb, db, _ = b.step(None)
        assert torch.equal(oa, ob) and torch.equal(da, db)
    assert torch.equal(a.pos, b.pos)


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE config 5 on one GPU: eight shards one after another == the unsharded 2 097 152-env run
# ----------------------------------------------------------------------------------------------------------------------
def test_config5_eight_shards_equal_one_unsharded_run():
    """2 097 152 envs of the mixed trajectory set (goal reward, auto-reset) as ONE collector run, then as the eight
    262 144-env shards of BASELINE config 5 (env_id0 = r x 262 144), each through its own RolloutCollector and
    BlockReturnGatherer: the assembled [world, E, n_local] returns -- what one all_gather_into_tensor over eight ranks
    delivers -- and a sampled set of transitions must agree bitwise with the unsharded run, episode lengths included."""
    import bench
    from mr_rl_amd import MRConfig
    from mr_rl_amd.collector import BlockReturnGatherer, RolloutCollector
    world, n_local, E, seed = 8, 262144, 4, 7
    N = world * n_local
    cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=seed)
    tab = bench.mixed_goal_table(cfg, seed)              # sets reward_mode="goal", min_dist2goal=1
    rng = np.random.default_rng(1)
    # sampled envs: the first and last env of every shard + 512 random ones per shard
    sample = np.unique(np.concatenate([[r * n_local, (r + 1) * n_local - 1] for r in range(world)] +
                                      [r * n_local + rng.integers(0, n_local, 512) for r in range(world)]))
    sample_t = torch.from_numpy(sample).cuda()
    keys = ("obs", "rew", "done", "actions")

    col = RolloutCollector(N, cfg=cfg, seed=seed, goal_table=tab, streams=2, returns_interval=E)
    g = BlockReturnGatherer(col, world_size=1)
    col.reset()
    full = []
    for k in range(E):
        col.collect(); g.gather()
        out = col.ready(k)
        full.append({q: out[q].index_select(1, sample_t).clone() for q in keys})
        col.release(k)
    col.join()
    ret_full = g.latest().clone().view(E, N)             # [1, E, N] single process
    len_full = col.len_blocks[0].clone()
    col.check_status()
    assert int((len_full > 0).sum()) > 0.9 * E * N       # episodes ended in (almost) every launch group
    del col, g, out
    torch.cuda.empty_cache()

    gathered, lens = [], []
    for r in range(world):
        c = RolloutCollector(n_local, cfg=cfg, seed=seed, env_id0=r * n_local, goal_table=tab, streams=2, returns_interval=E)
        gr = BlockReturnGatherer(c, world_size=1)
        c.reset()
        loc = torch.from_numpy(sample[(sample >= r * n_local) & (sample < (r + 1) * n_local)] - r * n_local).cuda()
        cols = np.nonzero((sample >= r * n_local) & (sample < (r + 1) * n_local))[0]
        for k in range(E):
            c.collect(); gr.gather()
            o = c.ready(k)
            for q in keys:
                assert torch.equal(o[q].index_select(1, loc), full[k][q][:, cols]), (r, k, q)
            c.release(k)
        c.join()
        gathered.append(gr.latest().clone())             # [1, E, n_local]: this rank's slab of the all-gather
        lens.append(c.len_blocks[0].clone())
        c.check_status()
        del c, gr
    allg = torch.cat(gathered, dim=0)                    # [world, E, n_local], rank-major = global env order per episode
    assert allg.shape == (world, E, n_local)
    assert torch.equal(allg.permute(1, 0, 2).reshape(E, N), ret_full)
    assert torch.equal(torch.stack(lens).permute(1, 0, 2).reshape(E, N), len_full)
