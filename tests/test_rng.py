"""CPU: the RNG definition (Philox4x32-10 + specified Box-Muller) of the oracle."""
import numpy as np
import scipy.stats as st

from oracle import oracle as O


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 (cross-checked here against ATen's
    PhiloxRNGEngine.h compiled on the host: identical words)."""
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_box_muller_matches_fp64_formula():
    rng = np.random.default_rng(0)
    ua = rng.integers(0, 2**32, 4000, dtype=np.uint64)
    ub = rng.integers(0, 2**32, 4000, dtype=np.uint64)
    for a, b in zip(ua, ub):
        z0, z1 = O.box_muller(int(a), int(b))
        r = np.sqrt(-2 * np.log((a + 0.5) / 2**32)); th = 2 * np.pi * (b + 0.5) / 2**32
        assert abs(z0 - r * np.cos(th)) < 2e-5 and abs(z1 - r * np.sin(th)) < 2e-5
    # edges: all octant boundaries, extreme u
    for b in [0, 2**29 - 1, 2**29, 2**30, 2**31, 2**32 - 1, 3 << 29, (5 << 29) + 7]:
        for a in [0, 1, 2**31, 2**32 - 1]:
            z0, z1 = O.box_muller(a, b)
            assert np.isfinite(z0) and np.isfinite(z1) and abs(z0) < 7 and abs(z1) < 7


def test_normal_distribution():
    z = O.fill_normals(12345, 7, 3, 0, 250_000).astype(np.float64)  # 1e6 draws
    n = len(z)
    assert abs(z.mean()) < 5 / np.sqrt(n)
    assert abs(z.std() - 1) < 5 / np.sqrt(2 * n)
    assert abs(st.skew(z)) < 0.02 and abs(st.kurtosis(z)) < 0.03
    assert st.kstest(z[:200_000], "norm").pvalue > 1e-3
    for a, b in [(0, 1), (0, 2), (1, 3), (2, 3)]:
        assert abs(np.corrcoef(z[a::4], z[b::4])[0, 1]) < 0.01


def test_streams_are_distinct():
    a = O.normals4(1, 0, 0, O.c0(O.STREAM_DYN, 0, 0))
    assert not np.array_equal(a, O.normals4(1, 1, 0, O.c0(O.STREAM_DYN, 0, 0)))   # env
    assert not np.array_equal(a, O.normals4(1, 0, 1, O.c0(O.STREAM_DYN, 0, 0)))   # step
    assert not np.array_equal(a, O.normals4(2, 0, 0, O.c0(O.STREAM_DYN, 0, 0)))   # seed
    assert not np.array_equal(a, O.normals4(1, 0, 0, O.c0(O.STREAM_CTOR, 0, 0)))  # stream
    assert not np.array_equal(a, O.normals4(1, 0, 0, O.c0(O.STREAM_DYN, 1, 0)))   # block
    assert np.array_equal(a, O.normals4(1, 0, 0, O.c0(O.STREAM_DYN, 0, 0)))


def test_noise_increment_law():
    """away from the origin one env step adds noise with std 0.868937*dt*sigma per axis
    and mean dt*(b1*v_prev + (1-b1)*v_new)."""
    n = 20000
    p = O.default_params(a0=1.0, sigma=1.0)
    v = O.VecOracle(n, p, seed=99)
    init = np.tile([[110.0, 115.0]], (n, 1))
    v.reset(step_idx=0, init_xy=init)
    act = np.tile(np.array([[4.0, 0.5]], dtype=np.float32), (n, 1))
    v.step(act, step_idx=1)
    y1 = v.envs["y"].copy()
    v.step(act, step_idx=2)
    d = v.envs["y"] - y1
    vx, vy = 4 * np.cos(np.float64(np.float32(0.5))), 4 * np.sin(np.float64(np.float32(0.5)))
    sd = 0.868937 * 0.03
    assert abs(d[:, 0].mean() - 0.03 * vx) < 5 * sd / np.sqrt(n)
    assert abs(d[:, 1].mean() - 0.03 * vy) < 5 * sd / np.sqrt(n)
    assert abs(d[:, 0].std() / sd - 1) < 0.03 and abs(d[:, 1].std() / sd - 1) < 0.03
    assert (v.envs["n_attempts"] == 1).all()


def test_auto_reset_draws_sit_at_the_first_step_of_the_episode_that_ends():
    """the start position of an auto-reset is RESET_POS(0) at step - (length - 1) -- the step at which the
    episode that ends took its first step -- for the global env id; distinct episodes of an env get distinct positions."""
    import ctypes as C
    from mr_rl_amd import MRConfig
    from tests.util import orc_params_from_cfg
    n, id0, seed = 300, 1000, 21
    cfg = MRConfig(noise_var=1.0, auto_reset=True, reward_mode="goal", min_dist2goal=5.0, seed=seed)
    tab = np.random.default_rng(1).uniform(104, 116, (3, 52, 2)).astype(np.float32)
    p = orc_params_from_cfg(cfg, 3, 52)
    v = O.VecOracle(n, p, seed=seed, env_id0=id0, goal_table=tab, threads=4)
    v.reset(0)
    first = np.full(n, 5, dtype=np.int64)            # step index of the first step of the running episode
    seen, checked = [set() for _ in range(n)], 0
    L = O.lib()
    L.orc_sample_init.argtypes = [C.POINTER(O.OrcParams), C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_double)]
    L.orc_sample_init.restype = None
    for k in range(5, 5 + 150):
        a = v.random_policy(k, cfg.policy_low, cfg.policy_high)
        v.step(a, step_idx=k)
        for i in np.nonzero(v.done)[0]:
            assert v.final_len[i] == k - first[i] + 1
            xy = (C.c_double * 2)()
            L.orc_sample_init(C.byref(p), seed, id0 + int(i), int(first[i]), xy)
            np.testing.assert_array_equal(v.envs["y"][i], [xy[0], xy[1]])
            np.testing.assert_array_equal(v.obs[i, :2], [xy[0], xy[1]])
            assert (xy[0], xy[1]) not in seen[i]
            seen[i].add((xy[0], xy[1]))
            first[i] = k + 1
            checked += 1
    assert checked > 5 * n


def test_auto_reset_positions_are_uniform_and_uncorrelated():
    """The start positions of auto-resets (keyed by the first step of the episode that ends): uniform over the init box
    (chi-square on a 10 x 10 grid), uncorrelated between consecutive episodes of an env, between neighbouring envs in the same
    step, and with the length of the episode that ended."""
    from mr_rl_amd import MRConfig
    from tests.util import orc_params_from_cfg
    n, seed = 4096, 5
    cfg = MRConfig(noise_var=1.0, auto_reset=True, reward_mode="goal", min_dist2goal=5.0, seed=seed)
    tab = np.random.default_rng(2).uniform(104, 116, (3, 52, 2)).astype(np.float32)
    v = O.VecOracle(n, orc_params_from_cfg(cfg, 3, 52), seed=seed, goal_table=tab, threads=8)
    v.reset(0)
    pos, prev, pairs, lens, neigh = [], {}, [], [], []
    for k in range(1, 121):
        v.step(v.random_policy(k, cfg.policy_low, cfg.policy_high), step_idx=k)
        d = np.nonzero(v.done)[0]
        xy = v.envs["y"][d].copy()
        pos.append(xy); lens.append(v.final_len[d].astype(np.float64))
        both = d[:-1][np.diff(d) == 1]                              # env i and i + 1 reset in the same step
        neigh.append(np.stack([v.envs["y"][both, 0], v.envs["y"][both + 1, 0]], 1))
        for i, p in zip(d, xy):
            if i in prev:
                pairs.append((prev[i][0], p[0], prev[i][1], p[1]))
            prev[i] = p
    pos, lens, pairs, neigh = np.concatenate(pos), np.concatenate(lens), np.asarray(pairs), np.concatenate(neigh)
    assert len(pos) > 40000 and len(pairs) > 30000 and len(neigh) > 2000
    u = (pos - 100.0) / 20.0
    assert u.min() >= 0 and u.max() <= 1
    h, _, _ = np.histogram2d(u[:, 0], u[:, 1], bins=10, range=[[0, 1], [0, 1]])
    chi2 = ((h - len(u) / 100) ** 2 / (len(u) / 100)).sum()
    assert st.chi2.sf(chi2, 99) > 1e-4, chi2
    lim = lambda m: 4.5 / np.sqrt(m)  # noqa: E731
    assert abs(np.corrcoef(pairs[:, 0], pairs[:, 1])[0, 1]) < lim(len(pairs))      # x of consecutive episodes of an env
    assert abs(np.corrcoef(pairs[:, 2], pairs[:, 3])[0, 1]) < lim(len(pairs))      # y
    assert abs(np.corrcoef(pairs[:, 0], pairs[:, 3])[0, 1]) < lim(len(pairs))      # x then y
    assert abs(np.corrcoef(pos[:, 0], pos[:, 1])[0, 1]) < lim(len(pos))            # x and y of one reset
    assert abs(np.corrcoef(neigh[:, 0], neigh[:, 1])[0, 1]) < lim(len(neigh))      # neighbouring envs, same step
    assert abs(np.corrcoef(pos[:, 0], lens)[0, 1]) < lim(len(pos))                 # position vs length of the ended episode


def test_reset_blocks_of_consecutive_episodes_are_distinct():
    """include/mrsim.h: a reset consumes a step index.  An explicit reset at step s draws its start position from block
    (env, s); the auto-reset that ends an episode draws from the block of the step at which that episode took its FIRST step.  With
    the rule (first step after a reset at s uses s + 1) every episode has a block of its own; breaking it (reset and first step
    both at s) makes the auto-reset reuse the explicit reset's block: the same start position twice in a row."""
    n = 64
    p = O.default_params(sigma=0.5, auto_reset=1)
    lo, hi = (-20.0, -2 * np.pi), (20.0, 2 * np.pi)

    def starts(first_step):
        orc = O.VecOracle(n, p, seed=5)
        orc.reset(1)                                  # explicit reset at step index 1
        out = [orc.envs["y"].copy()]
        k = first_step
        for ep in range(2):
            for _ in range(51):
                orc.step(orc.random_policy(k, lo, hi), step_idx=k)
                k += 1
            assert orc.done.all() and (orc.envs["counter"] == 0).all()
            out.append(orc.envs["y"].copy())          # the auto-reset's start position
        return out
    a = starts(first_step=2)                          # the rule
    for i in range(3):
        for j in range(i + 1, 3):
            assert (np.abs(a[i] - a[j]).max(axis=1) > 0).all()
    b = starts(first_step=1)                          # reset and first step share index 1
    assert np.array_equal(b[0], b[1])                 # ... and the first auto-reset repeats the explicit reset's draw
    assert (np.abs(b[1] - b[2]).max(axis=1) > 0).all()
