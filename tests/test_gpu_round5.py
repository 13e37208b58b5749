"""GPU tests of round 5 (all through the C ABI):

  * the kernels' sigma > 0 statistics against the REFERENCE's own samples at scale (tests/golden/ref_increments.npz schema 2,
    tests/increments.py): increments and rk_step attempts of Simulator.step (MR_simulator.py:36-52,73-83) far from the origin,
    where the first attempt's error_norm is about 1, and where every step is split -- both noise laws, both Box-Muller
    flavours, both model laws;
  * MrsimStepIO.attempts (ABI 5) element-wise against the oracle's attempt counter;
  * the library default law (collapsed) is what MRConfig() / mrsim_default_params give, and the oracle follows the same law;
  * the single-env facade MR_Env on its pinned host record: goldens (test_gpu_parity / test_gpu_round3 keep theirs), reuse after
    close(), a second device-side env next to it, and that a step is ONE kernel launch with no copy call.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import increments as INC
from tests.util import load_cases, orc_params_from_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KERNEL_ENVS = {"far": 262144, "mid": 262144, "origin": 131072, "near": 131072}


class _KernelSteps:
    def __init__(self, n, law, mis, math, seed, start):
        from mr_rl_amd import MRConfig, MRVecEnv
        self.env = MRVecEnv(n, cfg=MRConfig(noise_var=1.0, is_mismatched=mis, noise_law=law, noise_math=math), seed=seed,
                            track_attempts=True)
        self.env.reset(init=np.tile([list(start)], (n, 1)), is_mismatched=mis)   # fresh env: nominal-law constructor

    def pos(self):
        return self.env.pos.cpu().numpy()

    def k0(self):
        return self.env.aux[:, :2].cpu().numpy().astype(np.float64)

    def attempts(self):
        return self.env.attempts.cpu().numpy()

    def step(self, a):
        self.env.step(a)


@pytest.mark.parametrize("law", ["collapsed", "per_stage"])
@pytest.mark.parametrize("math", ["fast", "spec"])
@pytest.mark.parametrize("mis", [False, True], ids=["nominal", "mismatched"])
@pytest.mark.parametrize("regime", INC.REGIMES)
def test_kernel_increments_and_attempts_match_the_reference_sample(regime, law, math, mis):
    ref = INC.reference(regime, mis)
    n, steps = KERNEL_ENVS[regime], int(ref["steps"])
    st = _KernelSteps(n, law, mis, math, seed=33 + (law == "per_stage"), start=tuple(ref["start"]))
    resn, att = INC.draw(st, n, 8 if regime == "far" else steps, mis, seed=77)
    st.env.check_status()
    out = INC.compare(regime, mis, resn, att, label=f"kernel {law}/{math}")
    assert out["tol"] <= 0.02


@pytest.mark.parametrize("law", ["collapsed", "per_stage"])
@pytest.mark.parametrize("start", [(0.0, 0.0), (8.0, -6.0), (110.0, 115.0)])
def test_attempts_output_equals_the_oracles_counter(law, start):
    """noise_math = "spec": normals bit-identical to the oracle's, so accept / reject decisions agree except where error_norm
    lands within rounding of 1 (the oracle's err_margin tells): the attempt counts are equal for every env whose margin is
    not tiny."""
    from mr_rl_amd import MRConfig, MRVecEnv
    n = 4096
    cfg = MRConfig(noise_var=1.0, noise_law=law, noise_math="spec")
    env = MRVecEnv(n, cfg=cfg, seed=9, track_attempts=True)
    orc = O.VecOracle(n, orc_params_from_cfg(cfg), seed=9, threads=8)
    init = np.tile([list(start)], (n, 1))
    env.reset(init=init); orc.reset(0, init_xy=init)
    rng = np.random.default_rng(2)
    alive = np.ones(n, bool)      # envs that have agreed with the oracle so far (test_gpu_parity.py: near-origin splitting)
    got = want = None
    for t in range(6):
        a = np.stack([rng.uniform(-20, 20, n), rng.uniform(-2 * np.pi, 2 * np.pi, n)], 1).astype(np.float32)
        env.step(a); orc.step(a, step_idx=t + 1)
        got, want = env.attempts.cpu().numpy(), orc.envs["n_attempts"]
        bad = alive & (np.abs(env.pos.cpu().numpy() - orc.envs["y"]).max(axis=1) > 1e-6)
        assert (orc.envs["err_margin"][bad] < 1e-6).all()          # only a decision within rounding of its threshold may differ
        alive &= ~bad
        assert np.array_equal(got[alive], want[alive]), (t, int((got[alive] != want[alive]).sum()))
    assert alive.mean() >= 0.99
    if start == (110.0, 115.0):
        assert got.max() == 1
    if start == (0.0, 0.0):
        assert got[alive].mean() > 15 and want[alive].mean() > 15          # every step is split here (the fixture: 20-50 attempts)


def test_library_default_law_is_collapsed_everywhere():
    import ctypes as C
    from mr_rl_amd import MRConfig, _lib
    p = _lib.default_params()
    assert p.noise_law == _lib.LAW_COLLAPSED
    assert MRConfig().noise_law == "collapsed" and MRConfig().to_params().noise_law == _lib.LAW_COLLAPSED
    assert orc_params_from_cfg(MRConfig()).noise_law == O.LAW_COLLAPSED
    assert C.sizeof(_lib.MrsimStepIO) == 128


# ---------------------------------------------------------------------------
# the drop-in MR_Env on its host record
# ---------------------------------------------------------------------------
def test_facade_step_is_one_launch_on_the_host_record():
    """MR_env.py:70-98 through mr_rl_amd.MR_Env: the golden timeout episode again, this time counting what a step does --
    the env's state and outputs are views of one pinned block (no torch tensor involved), and what step() returns is read
    from that block."""
    from mr_rl_amd import MR_Env
    G = load_cases("ref_env.npz")["g6_timeout"]
    env = MR_Env()
    rec = env._rec
    assert rec.pos.ctypes.data == rec.host and rec.obs.ctypes.data == rec.host + 64 and rec.SIZE == 192
    obs0 = env.reset(init=G["init"], noise_var=0.0, a0=float(G["a0"]), is_mismatched=bool(G["mismatched"]))
    np.testing.assert_allclose(obs0, G["obs0"], rtol=1e-6, atol=1e-6)
    for k, a in enumerate(G["actions"][: len(G["obs"])]):
        obs, rew, done, info = env.step(a)
        assert rew == 10 and isinstance(rew, int) and info == {} and done == bool(G["done"][k])
        np.testing.assert_allclose(env.last_pos, G["last_pos"][k], rtol=0, atol=5e-5)
        assert env.last_pos == [float(rec.pos[0]), float(rec.pos[1])] and env.counter == int(rec.counter[0])
        assert obs.dtype == np.float64 and np.array_equal(obs, rec.obs.astype(np.float64))
    env.close()
    env.close()                                   # idempotent
    # two envs side by side keep separate records and separate RNG streams
    a, b = MR_Env(seed=1, env_id=0), MR_Env(seed=1, env_id=1)
    oa, ob = a.reset(init=[110.0, 115.0]), b.reset(init=[110.0, 115.0])
    assert np.array_equal(oa, ob)
    pa, _, _, _ = a.step([10.0, 1.0]); pb, _, _, _ = b.step([10.0, 1.0])
    assert not np.array_equal(pa, pb)             # sigma = 1: different env ids draw different noise


def test_facade_drives_run_sim_and_the_ddpg_loop_shape_like_the_reference():
    """The two callers a reference checkout has, written as THEY are written, on mr_rl_amd.MR_Env:
    utils.run_sim (utils.py:43-61: reset, then step every row of the action table ignoring `done`, reading env.last_pos and
    env.state_prime) against the golden tuple the reference's own run_sim produced, and the episode loop of RL/MR_ddpg.py:270-311
    (reset(), render(), step(np.squeeze(action)) until done)."""
    from mr_rl_amd import MR_Env
    E = load_cases("ref_env.npz")
    for name in ("g7_runsim_ramp", "g7_runsim_mis"):
        G = E[name]
        actions = G["actions"]
        env = MR_Env()
        state = env.reset(init=G["init"], noise_var=0.0, a0=float(G["a0"]), is_mismatched=bool(G["mismatched"]))   # utils.py:47
        assert state.shape == (5,)
        X, Y, states_prime = [], [], []
        for i in range(len(actions)):                                                                                 # utils.py:51-54
            env.step(actions[i, :2])
            X.append(env.last_pos[0]); Y.append(env.last_pos[1]); states_prime.append(env.state_prime)
        np.testing.assert_allclose(np.array(X), G["X"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(np.array(Y), G["Y"], rtol=0, atol=5e-5)
        assert np.isfinite(np.array(states_prime)).all() and env.counter == len(actions)
    # the DDPG loop's shape at the reference's defaults (sigma = 1): 51 steps, return 510, then the next episode
    env = MR_Env(seed=11)
    for ep in range(2):
        s = env.reset()                                                          # RL/MR_ddpg.py:270
        ep_reward, j = 0, 0
        while True:
            env.render()                                                         # :274
            a = np.array([[3.0 + 0.1 * j, 0.05 * j]])
            s2, r, terminal, info = env.step(np.squeeze(a))                      # :278
            ep_reward += r; j += 1
            if terminal:
                break
        assert j == 51 and ep_reward == 510 and info == {}
    with pytest.raises(RuntimeError, match="mrsim device status"):              # a NaN action: SciPy's solver would fail (:42-43)
        env.reset(); env.step([float("nan"), 0.0])
    obs = env.reset()                                                            # the env is usable again after the failure
    assert np.isfinite(obs).all() and np.isfinite(env.step([1.0, 1.0])[0]).all()


def test_facade_equals_vec_env_of_one():
    """the facade and an MRVecEnv of one env (same seed, same env id) walk the same trajectory bit for bit at sigma = 1:
    same kernel, same arguments -- only where the buffers live differs."""
    from mr_rl_amd import MR_Env, MRConfig, MRVecEnv
    env = MR_Env(seed=4, env_id=17)
    vec = MRVecEnv(1, cfg=MRConfig(), seed=4, env_id0=17, track_state_prime=True)
    init = np.array([101.5, 118.25])
    o1 = env.reset(init=init, noise_var=1.0, a0=1.0, is_mismatched=False)
    o2 = vec.reset(init=init[None, :], noise_var=1.0, a0=1.0, is_mismatched=False)
    assert np.array_equal(o1, o2[0].double().cpu().numpy())
    rng = np.random.default_rng(0)
    for t in range(60):
        a = np.array([rng.uniform(-20, 20), rng.uniform(-6, 6)], dtype=np.float32)
        obs, rew, done, _ = env.step(a)
        vo, vr, vd, _ = vec.step(a[None, :])
        assert np.array_equal(obs, vo[0].double().cpu().numpy()) and done == bool(vd[0]) and rew == float(vr[0])
        assert env.last_pos == vec.pos[0].cpu().numpy().tolist()
        assert np.array_equal(env.state_prime, vec.state_prime[0].double().cpu().numpy())


def test_facade_step_rate_beats_the_reference_python():
    """the reference's own MR_Env.step runs at 7.2-9.5 k steps/s on one core (profiles/r03/ref_python_baseline.json);
    the drop-in must not be slower than what it replaces (bench.py's `facade` leg reports the measured rate)."""
    import time
    from mr_rl_amd import MR_Env
    env = MR_Env()
    env.reset()
    for _ in range(200):
        env.step([5.0, 1.0])
    env.reset()
    t0 = time.perf_counter()
    n = 0
    for ep in range(20):
        env.reset()
        for _ in range(51):
            env.step([5.0, 1.0]); n += 1
    rate = n / (time.perf_counter() - t0)
    print(f"facade: {rate:.0f} MR_Env.step/s (resets included)")
    assert rate > 9500.0


# ---------------------------------------------------------------------------
# the learner across compute units for large batches (RL/MR_ddpg.py:288-305; mrsim_learner.h: multi-workgroup form)
# ---------------------------------------------------------------------------
def _learner_pair(B, seed=3):
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.ddpg import DDPG
    from tests.test_gpu_round4 import _randomise
    env = MRVecEnv(256, cfg=MRConfig(auto_reset=True), seed=0)
    a, b = DDPG(env, seed=seed, min_batch=B, fused=True), DDPG(env, seed=seed, min_batch=B, fused=True)
    _randomise(a, 5); _randomise(b, 5)
    return a, b


@pytest.mark.parametrize("B", [128, 256, 1024, 4096])
def test_multi_workgroup_update_equals_the_single_workgroup_loop_bitwise(B):
    """batch / 64 workgroups, each on its tile, gradients summed in tile order by the last workgroup to arrive == ONE workgroup
    looping over the tiles (which adds them in the same order): parameters, Adam moments, targets, losses and step counts are
    equal bit for bit after three updates on given batches -- and after three more with the rows drawn in the kernel."""
    import torch
    from tests.test_gpu_round4 import _batch
    multi, single = _learner_pair(B)
    single.fused.multi_workgroup = False
    assert multi.fused.multi_workgroup and not single.fused.multi_workgroup
    for k in range(3):
        batch = _batch(B, 200 + k)
        lm, ls = multi.update(batch), single.update(batch)
        assert float(lm[0]) == float(ls[0]) and float(lm[1]) == float(ls[1]), (k, float(lm[0]), float(ls[0]))
    for name in ("online", "target", "adam_m", "adam_v", "grad"):
        assert torch.equal(getattr(multi.fused, name), getattr(single.fused, name)), name
    assert multi.fused.steps.tolist() == single.fused.steps.tolist() == [3, 3]
    # rows drawn in the kernel from a filled ring (B <= 256: without repetition; beyond: every workgroup draws its own tile's rows)
    g = torch.Generator(device="cuda").manual_seed(1)
    n = 10000
    for ag in (multi, single):
        g.manual_seed(1)
        s = torch.randn(n, 5, device="cuda", generator=g)
        ag.buffer.add(s, torch.randn(n, 2, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g),
                      (torch.rand(n, device="cuda", generator=g) < 0.02).float(), s + 0.01 * torch.randn(n, 5, device="cuda", generator=g))
        ag.fused.idx_out = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    for k in range(3):
        multi.update(); single.update()
        assert torch.equal(multi.fused.idx_out, single.fused.idx_out)
        rows = multi.fused.idx_out
        assert int(rows.min()) >= 0 and int(rows.max()) < n
        if B <= 256:
            assert len(set(rows.tolist())) == B
    for name in ("online", "target", "adam_m", "adam_v"):
        assert torch.equal(getattr(multi.fused, name), getattr(single.fused, name)), name
    torch.cuda.synchronize()
    sc = multi.fused.batch_scratch
    tiles = B // 64
    counters = sc[tiles * 7680 + B + 2 * tiles: tiles * 7680 + B + 2 * tiles + 2].view(torch.int32)
    assert counters.tolist() == [0, 0]            # the arrival tickets are back at zero between launches


@pytest.mark.parametrize("B", [1024, 4096])
def test_multi_workgroup_update_equals_the_eager_pytorch_update(B):
    """the large-batch form against the PyTorch twin on the same batches (as test_gpu_round4 does for 64 and 256)"""
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.ddpg import DDPG
    from mr_rl_amd.learner import ACTOR_LAYOUT, CRITIC_LAYOUT
    from tests.test_gpu_round4 import _batch, _randomise
    env = MRVecEnv(256, cfg=MRConfig(auto_reset=True), seed=0)
    eager, fused = DDPG(env, seed=3, min_batch=B), DDPG(env, seed=3, min_batch=B, fused=True)
    _randomise(eager, 5); _randomise(fused, 5)
    for k in range(3):
        batch = _batch(B, 100 + k)
        le, lf = eager.update(batch), fused.update(batch)
        assert abs(float(le[0]) - float(lf[0])) <= 2e-5 * max(1.0, abs(float(le[0])))
        assert abs(float(le[1]) - float(lf[1])) <= 2e-5 * max(1.0, abs(float(le[1])))
    for net_e, net_f, layout in ((eager.actor, fused.actor, ACTOR_LAYOUT), (eager.critic, fused.critic, CRITIC_LAYOUT),
                                 (eager.actor_t, fused.actor_t, ACTOR_LAYOUT), (eager.critic_t, fused.critic_t, CRITIC_LAYOUT)):
        for path, off in layout:
            pe, pf = net_e.get_parameter(path).detach(), net_f.get_parameter(path).detach()
            scale = float(pe.abs().max())
            err = (pe - pf).abs()
            assert float((err <= 5e-6 * scale + 5e-7).float().mean()) >= 0.95, (path, float(err.max()), scale)
            assert float(err.max()) <= 1e-3, (path, float(err.max()))
    assert fused.fused.steps.tolist() == [3, 3]


@pytest.mark.parametrize("ring,B", [(64, 64), (65, 64), (100, 64), (127, 64), (128, 64), (256, 256), (300, 256)])
def test_in_kernel_sampler_never_repeats_a_row_on_a_nearly_empty_ring(ring, B):
    """random.sample's law (RL/MR_ddpg.py:37-44) when the ring holds fewer than two batches -- the first updates of DDPG.train with
    64 envs draw 64 of 64: a partial Fisher-Yates shuffle, no rejection rounds to run out of."""
    import torch
    multi, _ = _learner_pair(B)
    ag = multi
    g = torch.Generator(device="cuda").manual_seed(2)
    s = torch.randn(ring, 5, device="cuda", generator=g)
    ag.buffer.add(s, torch.randn(ring, 2, device="cuda", generator=g), torch.randn(ring, device="cuda", generator=g),
                  torch.zeros(ring, device="cuda"), s.clone())
    ag.fused.idx_out = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    seen = torch.zeros(ring, device="cuda")
    for k in range(40):
        ag.update()
        rows = ag.fused.idx_out.clone()
        assert int(rows.min()) >= 0 and int(rows.max()) < ring
        assert len(set(rows.tolist())) == B, (k, sorted(rows.tolist()))
        seen[rows.long()] += 1
    if ring > B:                # every row gets its turn (uniform over the ring: 40 x B / ring expected visits)
        assert int((seen == 0).sum()) == 0
        assert float(seen.max()) <= 40 and float(seen.float().std()) < 0.45 * 40 * B / ring + 3


# ---------------------------------------------------------------------------
# main.py's own noisy runs (noise_var 0.5, a0 1.5, mismatched, from the origin) in distribution, kernels vs the reference's 4000 runs
# ---------------------------------------------------------------------------
class _KernelMain:
    def __init__(self, n, law, math, seed):
        from mr_rl_amd import MRConfig, MRVecEnv
        self.env = MRVecEnv(n, cfg=MRConfig(noise_var=0.5, a0=1.5, is_mismatched=True, noise_law=law, noise_math=math), seed=seed,
                            track_attempts=True)
        self.env.reset(init=np.zeros((n, 2)), noise_var=0.5, a0=1.5, is_mismatched=True)     # fresh env: nominal-law constructor

    def pos(self):
        return self.env.pos.cpu().numpy()

    def attempts(self):
        return self.env.attempts.cpu().numpy()

    def step(self, a):
        self.env.step(a)


@pytest.mark.parametrize("law", ["collapsed", "per_stage"])
@pytest.mark.parametrize("math", ["fast", "spec"])
@pytest.mark.parametrize("run", ["idle", "learn"])
def test_kernel_reproduces_the_law_of_main_py_runs(run, law, math):
    ref = INC.main_runs(run)
    n = 32768
    st = _KernelMain(n, law, math, seed=5)
    pos, att = INC.run_main(st, ref, n)
    st.env.check_status()
    INC.compare_checkpoints(ref, pos, att, label=f"kernel {law}/{math} main.py {run}")


# ---------------------------------------------------------------------------
# velocity post-processing, fused and cut into time chunks (Learning_module.py:46-59,72-93)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("T,n,N", [(1800, 300, 14), (600, 1000, 14), (33, 7, 14), (64, 257, 14), (95, 3, 30), (2000, 64, 31),
                                   (400, 37, 40), (1, 4, 14), (2, 4, 14), (3, 5, 2)])
def test_fused_velocity_matches_scipy_numpy_over_chunk_boundaries(T, n, N):
    """mrsim_velocity (one fused launch over trajectories x time chunks; the three-pass form for windows beyond 15 steps: N = 40)
    against uniform_filter1d -> np.gradient -> uniform_filter1d and the drift mean(v[N:-N]) computed with scipy / numpy as the
    reference does: long runs (many chunks), lengths around the chunk and window sizes, T = 1 .. 3, non-uniform time axis"""
    import torch
    from scipy.ndimage import uniform_filter1d
    from mr_rl_amd.rollout import estimate_velocity
    rng = np.random.default_rng(T * 7 + n)
    traj = np.cumsum(rng.normal(0.1, 0.05, (T, n, 2)), axis=0)
    time_axis = np.linspace(0, (T - 1) / 30.0, T) + rng.uniform(0, 1e-3, T).cumsum() * (T > 100)
    v, D = estimate_velocity(torch.as_tensor(traj, device="cuda"), time_axis, n_filter=N)
    want = np.zeros_like(traj)
    for e in range(n):
        for d in range(2):
            p = uniform_filter1d(traj[:, e, d], N, mode="nearest")
            g = np.gradient(p, time_axis) if T > 1 else np.zeros(1)
            want[:, e, d] = uniform_filter1d(g, max(int(N / 2), 1), mode="nearest")
    np.testing.assert_allclose(v.cpu().numpy(), want, rtol=1e-9, atol=1e-9)
    if T > 2 * N:
        np.testing.assert_allclose(D.cpu().numpy(), want[N:-N].mean(0), rtol=1e-9, atol=1e-12)
    else:
        assert float(D.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# The reference-shaped loop's per-step bookkeeping as one launch (mrsim_replay_add_step)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fused_learner", [False, True])
@pytest.mark.parametrize("n_envs,ring", [(256, 10000), (192, 1000), (3000, 2048)])
def test_fused_step_bookkeeping_equals_the_pytorch_statements(n_envs, ring, fused_learner):
    """DDPG.train with fused_bookkeeping -- the step kernel stores the transitions in the ring itself (MrsimStepIO.replay), or ONE
    launch of mrsim_replay_add_step does after the step (replay add + `state = next_state` + finished-episode sums) -- against the
    same loop written as PyTorch statements: the ring (all five arrays, head, fill), every network parameter after the updates
    and the device policy's block are BIT-identical; the per-step mean returns agree to float32 summation order.  Rings that
    wrap (192 envs into 1000 slots) and rings smaller than one step (3000 envs into 2048 slots: the last 2048 stay)."""
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.ddpg import DDPG
    scale = (0.01, 0.01, 0.01, 0.01, 1.0)
    cfg = MRConfig(noise_var=0.1, reward_mode="goal", auto_reset=True, init_low=(20.0, 20.0), init_high=(30.0, 30.0),
                   min_dist2goal=25.0)
    out = []
    for fb in (False, "add_step", True):
        env = MRVecEnv(n_envs, cfg=cfg, seed=3, track_actions=True)
        agent = DDPG(env, seed=5, obs_scale=scale, device_actor=True, buffer_size=ring, fused=fused_learner)
        seen = []
        rets = agent.train(70, fused_bookkeeping=fb, observe=lambda k, o: seen.append(o.clone()))
        env.check_status()
        b = agent.buffer
        out.append({"rets": rets, "ring": [t.clone() for t in (b.s, b.a, b.r, b.t, b.s2)], "head": b.head, "count": b.count,
                    "params": [p.detach().clone() for m in (agent.actor, agent.critic, agent.actor_t, agent.critic_t)
                               for p in m.parameters()] if not fused_learner else [agent.fused.online.clone(), agent.fused.target.clone()],
                    "blob": agent.device_actor.blob.clone(), "seen": seen, "obs": env.obs.clone()})
    a = out[0]
    for b in out[1:]:
        assert a["head"] == b["head"] and a["count"] == b["count"] == min(ring, 70 * n_envs)
        for x, y in zip(a["ring"], b["ring"]):
            assert torch.equal(x, y)
        for x, y in zip(a["seen"], b["seen"]):
            assert torch.equal(x, y)                       # the observation handed to the policy at every step
        for x, y in zip(a["params"], b["params"]):
            assert torch.equal(x, y)
        assert torch.equal(a["blob"], b["blob"]) and torch.equal(a["obs"], b["obs"])
        assert len(a["rets"]) == len(b["rets"]) > 0
        np.testing.assert_allclose(a["rets"], b["rets"], rtol=2e-6, atol=1e-4)


@pytest.mark.gpu
def test_replay_add_step_rejects_bad_arguments_and_demands_its_conditions():
    import ctypes as C
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv, _lib
    from mr_rl_amd.ddpg import DDPG
    L = _lib.lib()
    t = torch.zeros(64, device="cuda")
    p = C.c_void_p(t.data_ptr())
    sc = (C.c_float * 5)(1, 1, 1, 1, 1)
    assert L.mrsim_replay_add_step(1, p, p, p, p, p, None, None, sc, p, p, p, p, p, 8, 8, p, None, None) == _lib.EINVAL  # head
    assert L.mrsim_replay_add_step(1, None, p, p, p, p, None, None, sc, p, p, p, p, p, 8, 0, p, None, None) == _lib.EINVAL
    assert L.mrsim_replay_add_step(0, p, p, p, p, p, None, None, sc, p, p, p, p, p, 8, 0, p, None, None) == _lib.OK
    env = MRVecEnv(64, cfg=MRConfig(auto_reset=False), seed=0, track_actions=True)
    agent = DDPG(env, seed=0, device_actor=True)
    with pytest.raises(ValueError, match="fused_bookkeeping=True needs"):
        agent.train(2, fused_bookkeeping=True)
    # the step's own sink: needs the in-kernel actor, a valid head, all five arrays
    env2 = MRVecEnv(64, cfg=MRConfig(auto_reset=True), seed=0, track_actions=True)
    env2.reset()
    sk = agent.buffer.sink()
    with pytest.raises(ValueError, match="replay needs actor"):
        env2.step(replay=sk)
    sk.head = agent.buffer.buffer_size
    with pytest.raises(_lib.MrsimError):
        env2.step(actor=agent.device_actor, replay=sk)
    sk.head, sk.s2 = 0, None
    with pytest.raises(_lib.MrsimError):
        env2.step(actor=agent.device_actor, replay=sk)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [64, 256, 1024])
def test_policy_upload_in_the_update_launch_equals_the_separate_launch(batch):
    """MrsimDdpgLearner.actor_blob: the update's launch folds and packs the new online actor into the behaviour policy's block (the
    tail of the single-workgroup kernel; a launch behind the multi-workgroup form) -- against mrsim_actor_pack_device as a launch of
    its own after every update, and against the library's HOST packer on the final network: the block, the ring and every
    parameter bit-identical after 60 steps of DDPG.train."""
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.actor import DeviceActor
    from mr_rl_amd.ddpg import DDPG
    scale = (0.01, 0.01, 0.01, 0.01, 1.0)
    cfg = MRConfig(noise_var=0.1, reward_mode="goal", auto_reset=True, init_low=(20.0, 20.0), init_high=(30.0, 30.0),
                   min_dist2goal=25.0)
    out = []
    for upload in (True, False):
        env = MRVecEnv(512, cfg=cfg, seed=3, track_actions=True)
        agent = DDPG(env, seed=5, obs_scale=scale, device_actor=True, buffer_size=8192, fused=True, min_batch=batch)
        agent.fused_upload = upload
        w0 = agent.device_actor.blob.clone()
        agent.train(60)
        env.check_status()
        assert not torch.equal(w0, agent.device_actor.blob)
        b = agent.buffer
        out.append({"blob": agent.device_actor.blob.clone(), "online": agent.fused.online.clone(), "target": agent.fused.target.clone(),
                    "ring": [t.clone() for t in (b.s, b.a, b.r, b.t, b.s2)]})
        agent.actor.eval()
        host = DeviceActor.from_module(agent.actor, obs_scale=scale, device="cuda", ou=False)     # mrsim_actor_pack_host
        assert torch.equal(host.blob, agent.device_actor.blob)
    a, b = out
    assert torch.equal(a["blob"], b["blob"]) and torch.equal(a["online"], b["online"]) and torch.equal(a["target"], b["target"])
    for x, y in zip(a["ring"], b["ring"]):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_the_reference_shaped_loop_learns_the_one_step_goal_task():
    """DDPG.train as the script runs it (RL/MR_ddpg.py:262-311: one env step, one replay add, ONE 64-row update per iteration; the
    policy inside the step kernel is the network that update has just produced) on task "C" of tools/learning_check.py with 64 envs in
    lockstep: returns start at the untrained policy's level and reach >= 60 (of +100) on EVERY one of four seeds within 4 000
    iterations (profiles/r05/train_loop_learning.txt: 72-91 by the second to fourth tenth; DDPG drifts afterwards on some seeds, so the
    criterion is the best tenth, and that it comes after the first)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_loop_learning", os.path.join(root, "tools", "train_loop_learning.py"))
    tl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tl)
    for seed in range(4):
        agent, rets = tl.run(64, 4000, seed, fused=True)
        assert agent._updates == 4000 and len(rets) > 3000 and np.isfinite(rets).all()
        k = len(rets) // 10
        tenths = [float(np.mean(rets[i:i + k])) for i in range(0, 10 * k, k)]
        assert tenths[0] < 45.0 and max(tenths[1:]) >= 60.0 and max(tenths[1:]) > tenths[0] + 30.0, (seed, tenths)


@pytest.mark.gpu
def test_step_word_is_stored_after_the_outputs_of_a_one_workgroup_launch():
    """MrsimStepIO.done_word: a launch of one workgroup stores done_value into a word of mrsim_host_alloc memory after its outputs
    (the one-env facade polls it instead of waiting for the stream: tests of the facade above run through it); launches of more
    than one workgroup refuse it.  Here: 200 envs whose outputs live in the SAME pinned block as the word -- when the host sees the
    word, rewards and done flags of all 200 envs are there, and they equal a twin env's stepped the ordinary way."""
    import ctypes as C
    import torch
    from mr_rl_amd import MRConfig, MRVecEnv, _lib
    L = _lib.lib()
    n = 200
    cfg = MRConfig(noise_var=1.0, auto_reset=True)
    env, twin = MRVecEnv(n, cfg=cfg, seed=4), MRVecEnv(n, cfg=cfg, seed=4)
    env.reset(); twin.reset()
    h, d = C.c_void_p(), C.c_void_p()
    _lib.check(L.mrsim_host_alloc(4096, C.byref(h), C.byref(d)), "mrsim_host_alloc")
    try:
        host = np.frombuffer((C.c_uint8 * 4096).from_address(h.value), dtype=np.uint8)
        word = host[:4].view(np.int32); rew = host[16:16 + 4 * n].view(np.float32); done = host[2048:2048 + n]
        for k in range(60):
            io = env._step_io(None)
            io.rew, io.done = d.value + 16, d.value + 2048
            io.done_word, io.done_value = d.value, 100 + k
            rew[:] = -1.0
            _lib.check(L.mrsim_step(C.byref(env._params), n, env.env_id0, C.byref(env._st), C.byref(io), env.seed_value, env.step_idx,
                                    env._stream()), "mrsim_step")
            env.step_idx += 1
            assert L.mrsim_host_wait_word(h, 100 + k, 5_000_000) == _lib.OK and int(word[0]) == 100 + k
            got_rew, got_done = rew.copy(), done.copy()            # no stream wait has happened since the launch
            twin.step(None)
            assert np.array_equal(got_rew, twin.rew.cpu().numpy()) and np.array_equal(got_done, twin._done_u8.cpu().numpy())
        torch.cuda.synchronize()
        assert torch.equal(env.obs, twin.obs)
        big = MRVecEnv(300, cfg=cfg, seed=4)
        big.reset()
        io = big._step_io(None)
        io.done_word, io.done_value = d.value, 5
        assert L.mrsim_step(C.byref(big._params), 300, 0, C.byref(big._st), C.byref(io), 4, 1, big._stream()) == _lib.EINVAL
    finally:
        torch.cuda.synchronize()
        L.mrsim_host_free(h)
