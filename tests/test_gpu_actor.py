"""GPU: the DDPG actor (+ OU noise) as an on-device policy source -- SURVEY 8(f) row 2, RL/MR_ddpg.py:80-160,59-78,270-311.
All through the C ABI (mrsim_actor_forward, MrsimStepIO.actor, MrsimRolloutIO.actor).

  * the HIP actor (f32 MFMA layers, VALU output layer, specified tanh) against the fp32 PyTorch Actor.forward of
    mr_rl_amd/ddpg.py (tolerance ACTOR_TOL relative to action_bound) and against the oracle's restatement (BITWISE: an
    f32-input MFMA is a k-ordered fmaf chain and the oracle follows the same order);
  * OU noise: bitwise with noise_math="spec", radius-scaled with "fast";
  * fused step (actor inside mr_step_kernel) == actor kernel -> step kernel, bitwise;
  * fused rollout (actor inside the time loop) == T x (actor kernel -> step kernel), bitwise, ragged sizes, both laws;
  * the closed collection loop against the oracle (actor + env restated on the CPU), 4096 envs and BASELINE config 4's
    262 144 envs, auto-reset, element-wise;
  * the collector with policy=actor on sub-shard streams == one launch; argument errors.
Parity with the reference's TF1 / tflearn network itself is UNPINNED (libraries absent)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.util import orc_params_from_cfg, random_actor

pytestmark = pytest.mark.gpu
ACTOR_TOL = 1e-5   # relative to action_bound, HIP fp32 vs PyTorch fp32 (scaled observations, the twin's configuration)
BF_TOL = 5e-6      # relative to action_bound: HIP bf16x3 layer vs the oracle's emulation of the same operand splits (the
                   # order in which the hardware sums the 16 products of one MFMA is not documented; the emulation sums
                   # them exactly and rounds once)
POS_TOL = 1e-6
SCALE = [0.01] * 5


def _env(n, seed=0, env_id0=0, goal_table=None, **kw):
    from mr_rl_amd import MRConfig, MRVecEnv
    return MRVecEnv(n, cfg=MRConfig(**kw), seed=seed, env_id0=env_id0, goal_table=goal_table, track_actions=True)


def _actor(module, scale=SCALE, **kw):
    from mr_rl_amd.actor import DeviceActor
    return DeviceActor.from_module(module, obs_scale=scale, device="cuda", **kw)


def _obs(n, seed=0):
    r = np.random.default_rng(seed)
    o = np.zeros((n, 5), dtype=np.float32)
    o[:, :2] = r.uniform(-300, 300, (n, 2))
    o[:, 2:4] = r.uniform(-50, 50, (n, 2))
    o[:, 4] = np.hypot(o[:, 2] - o[:, 0], o[:, 3] - o[:, 1])
    return o


BF16_TOL = 6e-2    # relative to action_bound: plain bf16 inference vs PyTorch fp32 on nets whose output layer saturates


@pytest.mark.parametrize("n", [1, 63, 64, 255, 257, 4096])
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("math", ["f32", "bf16x3", "bf16"])
def test_actor_kernel_matches_pytorch_and_oracle(n, layout, math):
    """mrsim_actor_forward without noise: vs Actor.forward (PyTorch fp32, GPU) within ACTOR_TOL of the bound, vs the oracle
    bitwise; ragged sizes exercise the lanes-past-n handling of the wave-wide MFMA."""
    from mr_rl_amd.actor import fold_actor
    for seed, out_scale in ((0, None), (1, 30.0)):
        m = random_actor(seed, out_scale=out_scale)
        act = _actor(m, ou=False, math=math)
        env = _env(n, obs_layout=layout, noise_var=0.0)
        obs = _obs(n, seed)
        obs_t = torch.from_numpy(obs).cuda()
        src = obs_t.t().contiguous() if layout == "soa" else obs_t
        got = act.forward(env, obs=src).cpu().numpy()
        with torch.no_grad():
            want = m.cuda()(obs_t * torch.tensor(SCALE, device="cuda")).cpu().numpy()
        bound = m.action_bound.cpu().numpy()
        if math == "bf16":   # ordinary bf16 inference: 1e-4 of the bound on the freshly initialised net, percent-level where tanh saturates
            assert (np.abs(got - want) / bound).max() < (2e-4 if out_scale is None else BF16_TOL)
            orc = O.actor_forward(O.make_actor(fold_actor(m.cpu(), SCALE), math=math), obs)
            assert (np.abs(got - orc) / bound).max() < 2e-4 * (1 if out_scale is None else 30)   # the same roundings, emulated
            continue
        assert (np.abs(got - want) / bound).max() < ACTOR_TOL
        orc = O.actor_forward(O.make_actor(fold_actor(m.cpu(), SCALE), math=math), obs)
        if math == "f32":
            assert np.array_equal(got, orc), np.abs(got - orc).max()      # an f32 MFMA is the oracle's fmaf chain, bit for bit
        else:
            assert (np.abs(got - orc) / bound).max() < BF_TOL, (np.abs(got - orc) / bound).max()
            # ... and the exact-f32 arithmetic (the oracle's fmaf chains = the F32 kernel, bitwise): the split keeps f32-class accuracy
            orc32 = O.actor_forward(O.make_actor(fold_actor(m.cpu(), SCALE)), obs)
            assert (np.abs(got - orc32) / bound).max() < BF_TOL, (np.abs(got - orc32) / bound).max()


def test_actor_kernel_raw_observations_tolerance():
    """Raw (unscaled) observations of a few hundred units, the reference's own input scaling: bitwise vs the oracle;
    vs PyTorch fp32 within 3x what fp32 evaluation order is worth there (PyTorch fp32 vs PyTorch fp64)."""
    from mr_rl_amd.actor import fold_actor
    m = random_actor(5, out_scale=2.0)
    act = _actor(m, scale=None, ou=False)
    env = _env(8192, noise_var=0.0)
    obs = _obs(8192, 5)
    got = act.forward(env, obs=torch.from_numpy(obs).cuda()).cpu().numpy()
    assert np.array_equal(got, O.actor_forward(O.make_actor(fold_actor(m, None)), obs))
    with torch.no_grad():
        w32 = m(torch.from_numpy(obs)).numpy()
        w64 = random_actor(5, out_scale=2.0).double()(torch.from_numpy(obs).double()).numpy()
    bound = m.action_bound.numpy()
    cond = (np.abs(w32 - w64) / bound).max()
    assert (np.abs(got - w32) / bound).max() < max(ACTOR_TOL, 3 * cond)


@pytest.mark.parametrize("noise_math", ["spec", "fast"])
def test_actor_kernel_ou_noise(noise_math):
    """actor.predict + OUNoise over 30 calls: spec Box-Muller bitwise vs the oracle (actions and OU state), fast within the
    generator's radius-scaled bound; the OU pair uses the step's DYN(0,0) words (sharding: offset env ids)."""
    from mr_rl_amd.actor import fold_actor
    n, seed, id0 = 1000, 3, 77
    m = random_actor(2)
    act = _actor(m, ou=True)
    env = _env(n, seed=seed, env_id0=id0, noise_math=noise_math)
    A = O.make_actor(fold_actor(m, SCALE), ou=True)
    ou = np.zeros((n, 2), dtype=np.float32)
    obs = _obs(n, 1)
    obs_t = torch.from_numpy(obs).cuda()
    for t in range(30):
        env.step_idx = 100 + t
        got = act.forward(env, obs=obs_t).cpu().numpy()
        want = O.actor_policy(A, obs, ou, seed, 100 + t, env_id0=id0)
        if noise_math == "spec":
            assert np.array_equal(got, want) and np.array_equal(act.ou_state.cpu().numpy(), ou)
        else:
            assert np.abs(got - want).max() < 2e-6 * (t + 2) and np.abs(act.ou_state.cpu().numpy() - ou).max() < 2e-6 * (t + 2)
            ou[:] = act.ou_state.cpu().numpy()   # keep the comparison per step


def _gym_loop(env, act, T, rec):
    """T x (actor kernel -> step kernel): the gym-loop form of RL/MR_ddpg.py:277-278"""
    for _ in range(T):
        a = act.forward(env)
        obs, rew, done, info = env.step(a)
        rec["actions"].append(a.clone()); rec["obs"].append(obs.clone()); rec["rew"].append(rew.clone())
        rec["done"].append(done.clone())


@pytest.mark.parametrize("mis", [False, True])
@pytest.mark.parametrize("sigma", [0.0, 1.0])
@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_fused_step_equals_actor_kernel_then_step(mis, sigma, math):
    n, T = 1000, 60
    m = random_actor(4, out_scale=20.0)
    envA, envB = (_env(n, seed=9, noise_var=sigma, is_mismatched=mis, auto_reset=True) for _ in range(2))
    actA, actB = _actor(m, math=math), _actor(m, math=math)
    envA.reset(); envB.reset()
    rec = {"actions": [], "obs": [], "rew": [], "done": []}
    _gym_loop(envA, actA, T, rec)
    for t in range(T):
        obs, rew, done, info = envB.step(actor=actB)
        assert torch.equal(envB._actions_out, rec["actions"][t]), t
        assert torch.equal(obs, rec["obs"][t]) and torch.equal(rew, rec["rew"][t]) and torch.equal(done, rec["done"][t])
    assert torch.equal(envA.pos, envB.pos) and torch.equal(envA.aux, envB.aux) and torch.equal(actA.ou_state, actB.ou_state)
    assert bool(torch.stack(rec["done"]).any())     # episodes ended and restarted inside the window
    envA.check_status(); envB.check_status()


@pytest.mark.parametrize("n", [1, 63, 257, 1000])
@pytest.mark.parametrize("mis", [False, True])
@pytest.mark.parametrize("math", ["f32", "bf16x3", "bf16"])
def test_fused_rollout_equals_gym_loop_bitwise(n, mis, math):
    """mrsim_rollout with MrsimRolloutIO.actor (carry f32) == T x (mrsim_actor_forward -> mrsim_step): every action,
    observation, reward, done flag, the final state and the OU state; two launches back to back (state / OU state /
    observation re-derived from HBM at the launch boundary)."""
    T = 70
    m = random_actor(6, out_scale=25.0)
    envA, envB = (_env(n, seed=5, env_id0=11, noise_var=1.0, is_mismatched=mis, auto_reset=True) for _ in range(2))
    actA, actB = _actor(m, math=math), _actor(m, math=math)
    envA.reset(); envB.reset()
    rec = {"actions": [], "obs": [], "rew": [], "done": []}
    _gym_loop(envA, actA, T, rec)
    out1 = envB.rollout(40, want=("obs", "rew", "done", "actions"), actor=actB, carry="f32")
    o1 = {k: v.clone() for k, v in out1.items() if not k.startswith("_")}
    out2 = envB.rollout(T - 40, want=("obs", "rew", "done", "actions"), actor=actB, carry="f32")
    for key in ("actions", "obs", "rew", "done"):
        got = torch.cat([o1[key], out2[key]])
        assert torch.equal(got, torch.stack(rec[key])), key
    assert torch.equal(envA.pos, envB.pos) and torch.equal(envA.aux, envB.aux) and torch.equal(envA.ep_ret, envB.ep_ret)
    assert torch.equal(actA.ou_state, actB.ou_state)
    envB.check_status()


@pytest.mark.parametrize("n", [1, 65, 700, 1300])
@pytest.mark.parametrize("mis", [False, True])
@pytest.mark.parametrize("math", ["f32", "bf16x3", "bf16"])
def test_goal_table_actor_rollout_specialised_equals_generic_and_the_gym_loop(n, mis, math):
    """A policy collecting on a trajectory set (goal table, goal reward, fp64 carry: BASELINE config 5's workload with the actor
    as its policy) takes a flag-specialised mr_rollout_actor_fl_kernel that keeps the reset cache and 512-thread blocks.  Goals
    inside the init box and a goal radius of 6: episodes of 1-10 steps.  (a) bitwise equal to the generic actor rollout kernel
    (the same launch with one more output requested), every transition, the carried state, the OU state, over three launches;
    (b) with the fp32 carry, where a rollout equals single steps: == T x (mrsim_actor_forward -> mrsim_step)."""
    tab = np.random.default_rng(8).uniform(104, 116, (3, 52, 2)).astype(np.float32)
    kw = dict(noise_var=1.0, is_mismatched=mis, auto_reset=True, reward_mode="goal", min_dist2goal=6.0)
    m = random_actor(6, out_scale=25.0)
    envA, envB = (_env(n, seed=5, env_id0=11, goal_table=tab, **kw) for _ in range(2))
    actA, actB = _actor(m, math=math, reset_on_done=True), _actor(m, math=math, reset_on_done=True)
    envA.reset(); envB.reset()
    ndone = 0
    for T in (33, 7, 40):
        a = envA.rollout(T, want=("obs", "rew", "done", "actions"), actor=actA, carry="f64")
        b = envB.rollout(T, want=("obs", "rew", "done", "actions", "traj"), actor=actB, carry="f64")
        for key in ("actions", "obs", "rew", "done"):
            assert torch.equal(a[key], b[key]), (T, key)
        ndone += int(a["done"].sum())
    assert torch.equal(envA.pos, envB.pos) and torch.equal(envA.aux, envB.aux) and torch.equal(envA.ep_ret, envB.ep_ret)
    assert torch.equal(actA.ou_state, actB.ou_state) and torch.equal(envA.final_len, envB.final_len)
    assert ndone > (n // 2 if n > 1 else 0), ndone      # resets happened: the cache and the in-step reset were both exercised
    envA.check_status(); envB.check_status()
    # (b) fp32 carry against the gym loop (generic kernel: the specialised one is built for the fp64 carry only)
    envC, envD = (_env(n, seed=6, env_id0=3, goal_table=tab, **kw) for _ in range(2))
    actC, actD = _actor(m, math=math, reset_on_done=True), _actor(m, math=math, reset_on_done=True)
    envC.reset(); envD.reset()
    rec = {"actions": [], "obs": [], "rew": [], "done": []}
    _gym_loop(envC, actC, 30, rec)
    out = envD.rollout(30, want=("obs", "rew", "done", "actions"), actor=actD, carry="f32")
    for key in ("actions", "obs", "rew", "done"):
        assert torch.equal(out[key], torch.stack(rec[key])), key
    assert torch.equal(envC.pos, envD.pos) and torch.equal(actC.ou_state, actD.ou_state)


def _closed_loop_oracle(cfg, m, n, T, seed, gpu, id0=0, reset_on_done=False, threads=8, exact_actions=True, math="f32"):
    """The collection loop on the CPU, teacher-forced: at every step the oracle's actor sees the observation the KERNEL's
    actor saw (row t - 1 of the kernel's observations; the reset observation for t = 0) and must produce the kernel's
    action (bitwise with noise_math="spec"); the oracle's env is then stepped with the kernel's action and must land where
    the kernel's env landed.  Free-running, the two loops drift apart through the float32 rounding of the observations
    (an ulp of 110 is 7.6e-6; the feedback through the network amplifies it to ~1e-6 of position per episode): that
    comparison is made too, with a tolerance that says so."""
    from mr_rl_amd.actor import fold_actor
    orc = O.VecOracle(n, orc_params_from_cfg(cfg), seed=seed, env_id0=id0, threads=threads)
    free = O.VecOracle(n, orc_params_from_cfg(cfg), seed=seed, env_id0=id0, threads=threads)
    A = O.make_actor(fold_actor(m, SCALE), ou=True, reset_on_done=reset_on_done, math=math)
    orc.reset(0)
    obs_free = free.reset(0).astype(np.float32)
    ou, ou_free = np.zeros((n, 2), dtype=np.float32), np.zeros((n, 2), dtype=np.float32)
    bound = m.action_bound.numpy()
    worst = {"action": 0.0, "pos": 0.0, "free_pos": 0.0, "free_action": 0.0}
    obs_in = gpu["obs0"]
    for t in range(T):
        a = O.actor_policy(A, obs_in, ou, seed, t + 1, env_id0=id0, counter=orc.envs["counter"].copy(), threads=threads)
        if exact_actions:
            assert np.array_equal(a, gpu["actions"][t]), (t, np.abs(a - gpu["actions"][t]).max())
        else:
            worst["action"] = max(worst["action"], float((np.abs(a - gpu["actions"][t]) / bound).max()))
            ou[:] = gpu["actions"][t] - (a - ou)      # continue from the kernel's OU state (action = net + ou)
        o64, rew, done = orc.step(gpu["actions"][t], step_idx=t + 1)
        assert np.array_equal(done, gpu["done"][t]), t
        tol = 2 * np.spacing(np.abs(o64).astype(np.float32)) + 5 * POS_TOL
        assert (np.abs(gpu["obs"][t] - o64) <= tol).all(), t
        live = done == 0     # traj rows hold the pre-reset position of a done step; the oracle's y is post-reset there
        if live.any():
            worst["pos"] = max(worst["pos"], float(np.abs(gpu["traj"][t] - orc.envs["y"])[live].max()))
        obs_in = gpu["obs"][t]
        af = O.actor_policy(A, obs_free, ou_free, seed, t + 1, env_id0=id0, counter=free.envs["counter"].copy(), threads=threads)
        of64, _, dfree = free.step(af, step_idx=t + 1)
        obs_free = of64.astype(np.float32)
        assert np.array_equal(dfree, gpu["done"][t])
        worst["free_action"] = max(worst["free_action"], float((np.abs(af - gpu["actions"][t]) / bound).max()))
        if live.any():
            worst["free_pos"] = max(worst["free_pos"], float(np.abs(gpu["traj"][t] - free.envs["y"])[live].max()))
    return worst, ou, orc, free


@pytest.mark.parametrize("reset_on_done", [False, True])
def test_fused_rollout_against_the_oracle_closed_loop(reset_on_done):
    """4096 envs x 110 steps (two auto-resets), noise_math="spec", fp64 carry."""
    n, T, seed = 4096, 110, 21
    m = random_actor(8, out_scale=15.0)
    env = _env(n, seed=seed, noise_var=1.0, auto_reset=True, noise_math="spec")
    act = _actor(m, reset_on_done=reset_on_done)
    obs0 = env.reset().cpu().numpy().copy()
    out = env.rollout(T, want=("traj", "obs", "done", "actions"), actor=act, carry="f64")
    env.check_status()
    gpu = {"obs0": obs0, "obs": out["obs"].cpu().numpy(), "actions": out["actions"].cpu().numpy(),
           "done": out["done"].cpu().numpy().astype(np.uint8), "traj": out["traj"].cpu().numpy()}
    worst, ou, orc, free = _closed_loop_oracle(env.cfg, m, n, T, seed, gpu, reset_on_done=reset_on_done)
    assert worst["pos"] < POS_TOL, worst
    assert np.array_equal(act.ou_state.cpu().numpy(), ou)
    assert gpu["done"].sum() == 2 * n
    # free-running loops: same episodes, actions within 1e-4 of the bound, positions within 2e-5 after 110 steps
    assert worst["free_action"] < 1e-4 and worst["free_pos"] < 2e-5, worst


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_config4_full_size_actor_in_the_loop_against_the_oracle(math):
    """BASELINE config 4's 262 144 envs, one 51-step episode + the reset row, fast noise, actor + OU in the kernel,
    element-wise against the CPU restatement of the same loop (teacher-forced; the OU normals of noise_math="fast" are
    within 1e-6 + 8e-7 r of the oracle's, so actions are compared to 1e-6 of the bound instead of bitwise; bf16x3: BF_TOL)."""
    n, T, seed = 262144, 52, 7
    m = random_actor(9, out_scale=15.0)
    env = _env(n, seed=seed, noise_var=1.0, auto_reset=True)
    act = _actor(m, math=math)
    obs0 = env.reset().cpu().numpy().copy()
    out = env.rollout(T, want=("traj", "obs", "done", "actions", "rew"), actor=act, carry="f64")
    env.check_status()
    assert (out["rew"] == 10).all()
    gpu = {"obs0": obs0, "obs": out["obs"].cpu().numpy(), "actions": out["actions"].cpu().numpy(),
           "done": out["done"].cpu().numpy().astype(np.uint8), "traj": out["traj"].cpu().numpy()}
    worst, ou, orc, free = _closed_loop_oracle(env.cfg, m, n, T, seed, gpu, threads=O.lib().orc_num_threads(),
                                               exact_actions=False, math=math)
    assert worst["action"] < (1e-6 if math == "f32" else BF_TOL) and worst["pos"] < 5 * POS_TOL, worst
    assert worst["free_action"] < 1e-4 and worst["free_pos"] < 2e-5, worst
    assert (env.final_len == 51).all() and gpu["done"][50].all()


@pytest.mark.parametrize("math", ["f32", "bf16x3", "bf16"])
def test_collector_with_actor_policy_equals_single_launch(math):
    from mr_rl_amd import MRConfig
    from mr_rl_amd.collector import RolloutCollector
    n, seed = 5000, 13
    m = random_actor(10, out_scale=10.0)
    env = _env(n, seed=seed, noise_var=1.0, auto_reset=True)
    actA = _actor(m, math=math)
    env.reset()
    want = [env.rollout(51, want=("obs", "rew", "done", "actions"), actor=actA, carry="f64") for _ in range(1)]
    w = {k: v.clone() for k, v in want[0].items() if not k.startswith("_")}
    w2 = env.rollout(51, want=("obs", "rew", "done", "actions"), actor=actA, carry="f64")
    for S in (1, 2, 3):
        actB = _actor(m, math=math)
        col = RolloutCollector(n, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=seed, streams=S, policy=actB)
        col.reset()
        col.collect()
        e0 = {k: v.clone() for k, v in col.ready(0).items()}
        col.release(0)
        col.collect()
        e1 = col.ready(1)
        for key in ("obs", "rew", "done", "actions"):
            assert torch.equal(e0[key], w[key]), (S, key)
            assert torch.equal(e1[key], w2[key]), (S, key)
        col.join()
        assert torch.equal(col.env.pos, env.pos) and torch.equal(actB.ou_state, actA.ou_state)
        col.check_status()


def test_actor_argument_errors():
    from mr_rl_amd import _lib
    m = random_actor(0)
    act = _actor(m)
    env = _env(256, noise_var=1.0)
    env.reset()
    a = torch.zeros((256, 2), device="cuda")
    with pytest.raises(ValueError):
        env.step(a, actor=act)
    io = env._step_io(a, act)                              # both policy sources at the C ABI
    rc = env._L.mrsim_step(C.byref(env._params), 256, 0, C.byref(env._st), C.byref(io), 0, 1, env._stream())
    assert rc == _lib.EINVAL
    env2 = _env(256, integrator="euler", substeps=4)       # the actor drives the reference integrator only
    env2.reset()
    with pytest.raises(_lib.MrsimError):
        env2.step(actor=act)
    bad = _lib.MrsimActor(act.blob.data_ptr() + 4, None, 0.15, 0.3, 1e-2, 0)
    out = torch.zeros((256, 2), device="cuda")
    rc = env._L.mrsim_actor_forward(C.byref(env._params), 256, 0, C.byref(bad), None, env._p(env._obs), env._p(out), 0, 0,
                                    env._stream())
    assert rc == _lib.EALIGN
    rst = _lib.MrsimActor(act.blob.data_ptr(), act.ou_tensor(256).data_ptr(), 0.15, 0.3, 1e-2, 1)
    rc = env._L.mrsim_actor_forward(C.byref(env._params), 256, 0, C.byref(rst), None, env._p(env._obs), env._p(out), 0, 0,
                                    env._stream())
    assert rc == _lib.EINVAL                               # ou_reset_on_done needs the env state (MR_Env.counter)


def test_ddpg_twin_trains_with_the_device_actor():
    """mr_rl_amd.ddpg.DDPG(device_actor=True): the behaviour policy runs inside the env's step kernel and follows the
    learner (parameters re-uploaded after every update): after training, the device actor's noise-free output equals the
    PyTorch actor's eval-mode forward on the same observations to ACTOR_TOL of the bound."""
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.actor import DeviceActor
    from mr_rl_amd.ddpg import DDPG
    cfg = MRConfig(noise_var=0.1, reward_mode="goal", auto_reset=True, init_low=(20.0, 20.0), init_high=(30.0, 30.0),
                   min_dist2goal=25.0)
    env = MRVecEnv(256, cfg=cfg, seed=0, track_actions=True)
    agent = DDPG(env, seed=0, obs_scale=SCALE, device_actor=True)
    w0 = agent.device_actor.blob.clone()
    rets = agent.train(120)
    assert agent.buffer.size() == 10000 and len(rets) > 0 and all(np.isfinite(rets))
    assert not torch.equal(w0, agent.device_actor.blob)            # the learner's updates reached the device policy
    env.check_status()
    agent.actor.eval()
    quiet = DeviceActor.from_module(agent.actor, obs_scale=SCALE, device="cuda", ou=False)
    got = quiet.forward(env)
    with torch.no_grad():
        want = agent.actor(env.obs * torch.tensor(SCALE, device="cuda"))
    bound = agent.actor.action_bound
    assert ((got - want).abs() / bound).max().item() < ACTOR_TOL
    # what train() uploaded last (folded, packed and written ON THE DEVICE, DeviceActor.load_module_device) is bit for bit what the
    # library's host packer makes of the same network
    assert torch.equal(quiet.blob, agent.device_actor.blob)


def test_ddpg_twin_trains_at_collection_speed():
    """DDPG.train_collected: every episode of all envs is one fused launch group with the agent's actor in the kernel; sampled
    transitions (s from the previous row / the previous episode's last row, a = the action the kernel applied) feed the replay
    ring; the uploaded parameters follow the learner."""
    from mr_rl_amd import MRConfig, MRVecEnv
    from mr_rl_amd.ddpg import DDPG
    cfg = MRConfig(noise_var=0.1, reward_mode="goal", auto_reset=True, init_low=(20.0, 20.0), init_high=(30.0, 30.0),
                   min_dist2goal=25.0)
    env = MRVecEnv(4096, cfg=cfg, seed=0)
    agent = DDPG(env, seed=0, obs_scale=SCALE)
    rets = agent.train_collected(6, updates_per_episode=3, sample=4000)
    assert len(rets) == 6 and np.isfinite(rets).all()
    assert agent.buffer.size() == 10000 and agent._updates == 18
    # a stored transition is consistent: its action lies in the actor's range + OU noise, its state / next state are scaled obs
    assert float(agent.buffer.a[:, 0].abs().max()) < 20 + 3 and float(agent.buffer.s.abs().max()) < 60
    agent.actor.eval()
    with torch.no_grad():
        want = agent.actor(agent.collector.env.obs * torch.tensor(SCALE, device="cuda"))
    from mr_rl_amd.actor import DeviceActor
    quiet = DeviceActor.from_module(agent.actor, obs_scale=SCALE, device="cuda", ou=False)
    got = quiet.forward(agent.collector.env)
    assert ((got - want).abs() / agent.actor.action_bound).max().item() < ACTOR_TOL
    # both parameter blocks of the behaviour policy hold the learner's final network (uploaded on the device)
    for b in agent.device_actor.blobs:
        assert torch.equal(b, quiet.blob)
