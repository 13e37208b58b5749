"""CPU: the C-ABI library loads and exports every symbol include/mrsim.h declares; host-side logic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="mrsim.h"):
    h = open(os.path.join(ROOT, "include", header)).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)          # declarations only, not the prose around them
    return sorted(set(re.findall(r"\b(mrsim_[a-z_0-9]+)\s*\(", h)))


def test_library_loads_and_exports_header_symbols():
    """include/mrsim.h = the product ABI, include/mrsim_bench.h = measurement / test aids: the library exports every symbol
    either declares, the binding lists exactly those, and the two sets are disjoint."""
    from mr_rl_amd import _lib
    L = _lib.lib()
    product, bench = _declared_symbols("mrsim.h"), _declared_symbols("mrsim_bench.h")
    assert product and bench, "no declarations parsed"
    for s in product + bench:
        assert hasattr(L, s), f"libmrsim.so does not export {s}"
    assert sorted(_lib.PRODUCT_SYMBOLS) == product
    assert sorted(_lib.BENCH_SYMBOLS) == bench
    assert not set(product) & set(bench)
    assert L.mrsim_abi_version() == _lib.ABI_VERSION == 5


def test_product_modules_do_not_call_the_measurement_aids():
    """mrsim_*_timed / *_events / debug_normals are bound for bench.py, tools/ and tests/; the product path (env, collector,
    rollout, recorder, learner, dist, partition) does not call them.  (MRVecEnv keeps thin wrappers that bench.py drives:
    step_timed, rollout(timed= / events=) -- those are the only call sites.)"""
    pkg = os.path.join(ROOT, "mr_rl_amd")
    for name in sorted(os.listdir(pkg)):
        if not name.endswith(".py") or name in ("_lib.py", "vec_env.py"):
            continue
        src = open(os.path.join(pkg, name)).read()
        for sym in ("mrsim_step_timed", "mrsim_rollout_timed", "mrsim_rollout_events", "mrsim_step_events", "mrsim_debug_normals"):
            assert sym not in src, (name, sym)


def test_default_params_match_reference_constants():
    from mr_rl_amd import _lib
    p = _lib.default_params()
    assert p.time_span == 0.030 and p.rtol == 0.030 / 100 and p.atol == 1e-4   # MR_simulator.py:12-13,91
    assert p.a0 == 1.0 and p.sigma == 1.0 and p.mismatched == 0                 # MR_env.py:167-169
    assert p.max_timesteps == 50 and p.min_dist2goal == 30.0                    # MR_env.py:62-63
    assert list(p.obs_low) == [-5000, -5000, -5000, -5000, 0] and list(p.obs_high) == [5000] * 4 + [80000]
    assert list(p.init_low) == [100, 100] and list(p.init_high) == [120, 120]
    assert p.integrator == _lib.INT_RK45 and p.reward_mode == _lib.REW_CONSTANT10 and p.auto_reset == 0


def test_struct_layouts_match_oracle_defaults():
    """MRConfig -> MrsimParams and the oracle's OrcParams agree on every shared field."""
    from mr_rl_amd import MRConfig
    from oracle import oracle as O
    from tests.util import orc_params_from_cfg
    cfg = MRConfig(noise_var=0.5, a0=1.5, is_mismatched=True, integrator="rk4", substeps=4, reward_mode="goal",
                   auto_reset=True)
    p, q = cfg.to_params(3, 7), orc_params_from_cfg(cfg, 3, 7)
    for f in ("time_span", "rtol", "atol", "a0", "sigma", "min_dist2goal", "mismatched", "integrator", "substeps",
              "reward_mode", "max_timesteps", "auto_reset", "goal_K", "goal_T"):
        assert getattr(p, f) == getattr(q, f), f
    for f in ("obs_low", "obs_high", "init_low", "init_high"):
        assert list(getattr(p, f)) == list(getattr(q, f)), f


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU every compute entry point must refuse (no CPU path)."""
    import torch
    from mr_rl_amd import _lib, MRVecEnv
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    assert L.mrsim_device_count() == 0
    p = _lib.default_params()
    buf = np.zeros(64, dtype=np.float64)
    st = _lib.MrsimState(buf.ctypes.data, buf.ctypes.data, buf.ctypes.data)
    io = _lib.MrsimStepIO(None, None, None, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, None, None, None,
                          None, None)
    assert L.mrsim_step(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.ENODEVICE
    assert L.mrsim_reset(C.byref(p), 4, 0, C.byref(st), None, None, None, None, 0, 0, 0, None) == _lib.ENODEVICE
    assert L.mrsim_random_policy(C.byref(p), 4, 0, buf.ctypes.data, 0, 0, None) == _lib.ENODEVICE
    h, d = C.c_void_p(), C.c_void_p()
    assert L.mrsim_host_alloc(192, C.byref(h), C.byref(d)) == _lib.ENODEVICE and not h.value
    assert L.mrsim_host_alloc(0, C.byref(h), C.byref(d)) == _lib.EINVAL
    assert L.mrsim_host_free(None) == _lib.EINVAL
    assert L.mrsim_stream_synchronize(None) == _lib.ENODEVICE
    with pytest.raises(RuntimeError):
        MRVecEnv(4)
    from mr_rl_amd import MR_Env
    with pytest.raises(RuntimeError):
        MR_Env()


def test_c_demo_program_builds_and_refuses_without_a_device():
    """examples/abi_demo (pure C++ consumer of include/mrsim.h) is built with the library; without a GPU it must stop
    with an error, not compute anything."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "examples", "abi_demo")
    assert os.path.exists(exe), "make -C mr_rl_amd/csrc builds it"
    if torch.cuda.is_available():
        pytest.skip("GPU present (the GPU suite runs it)")
    r = subprocess.run([exe, "64"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no HIP device" in r.stderr


def test_argument_validation():
    from mr_rl_amd import _lib
    L = _lib.lib()
    p = _lib.default_params()
    buf = np.zeros(64, dtype=np.float64)
    st = _lib.MrsimState(buf.ctypes.data, buf.ctypes.data, buf.ctypes.data)
    io = _lib.MrsimStepIO(None, None, None, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, None, None, None,
                          None, None)
    assert L.mrsim_step(None, 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL
    assert L.mrsim_step(C.byref(p), -1, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL
    assert L.mrsim_step(C.byref(p), 2**33, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.ERANGE
    assert L.mrsim_step(C.byref(p), 4, 0, None, C.byref(io), 0, 0, None) == _lib.EINVAL
    st_bad = _lib.MrsimState(buf.ctypes.data + 8, buf.ctypes.data, buf.ctypes.data)
    assert L.mrsim_step(C.byref(p), 4, 0, C.byref(st_bad), C.byref(io), 0, 0, None) == _lib.EALIGN
    p.integrator = 7
    assert L.mrsim_step(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL
    assert "align" in _lib.strerror(_lib.EALIGN)


def test_rollout_and_velocity_argument_validation():
    from mr_rl_amd import _lib
    L = _lib.lib()
    p = _lib.default_params()
    buf = np.zeros(256, dtype=np.float64)
    st = _lib.MrsimState(buf.ctypes.data, buf.ctypes.data, buf.ctypes.data)
    io = _lib.MrsimRolloutIO(-1, 0, None, None, None, None, None, None, None, None, None, None, None)
    assert L.mrsim_rollout(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL      # T < 0
    io.T = 0
    assert L.mrsim_rollout(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.OK          # T == 0: no-op
    assert L.mrsim_rollout(C.byref(p), 4, 0, C.byref(st), None, 0, 0, None) == _lib.EINVAL
    io.T = 3
    io.traj_xy = buf.ctypes.data + 8
    assert L.mrsim_rollout(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EALIGN
    io.traj_xy = None
    io.goal_table = buf.ctypes.data
    p.goal_K = 0
    assert L.mrsim_rollout(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL      # table without K/T
    p = _lib.default_params()
    p.noise_math = 9
    io.goal_table = None
    assert L.mrsim_rollout(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL
    p = _lib.default_params()
    p.sigma = -1.0
    assert L.mrsim_rollout(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL
    d = buf.ctypes.data
    assert L.mrsim_velocity(4, 0, 14, d, d, d, d, None, None) == _lib.EINVAL                            # T < 1
    assert L.mrsim_velocity(4, 8, 14, None, d, d, d, None, None) == _lib.EINVAL
    assert L.mrsim_velocity(4, 8, 14, d + 8, d, d, d, None, None) == _lib.EALIGN
    assert L.mrsim_advance_step_base(None, 1, None) == _lib.EINVAL


def test_abi4_entry_points_validate_their_arguments_and_have_no_cpu_path():
    """noise_law, mrsim_ddpg_update, mrsim_replay_push, mrsim_actor_pack_device: bad arguments -> MRSIM_EINVAL / MRSIM_EALIGN,
    good arguments without a GPU -> MRSIM_ENODEVICE (no CPU fallback anywhere)"""
    from mr_rl_amd import _lib
    L = _lib.lib()
    p = _lib.default_params()
    assert p.noise_law == _lib.LAW_COLLAPSED           # the library default since ABI 5 (include/mrsim.h: MRSIM_LAW_*)
    buf = np.zeros(4096, dtype=np.float64)
    st = _lib.MrsimState(buf.ctypes.data, buf.ctypes.data, buf.ctypes.data)
    io = _lib.MrsimStepIO(None, None, None, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data)
    p.noise_law = 7
    assert L.mrsim_step(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) == _lib.EINVAL
    p.noise_law = _lib.LAW_COLLAPSED
    assert L.mrsim_step(C.byref(p), 4, 0, C.byref(st), C.byref(io), 0, 0, None) in (_lib.ENODEVICE, _lib.OK)
    # learner
    big = np.zeros(_lib.DDPG_PARAMS + 16, dtype=np.float32)
    base = big.ctypes.data + (-big.ctypes.data) % 16
    lr = _lib.MrsimDdpgLearner(base, base, base, base, base, base, base, 1e-5, 0.99, 0.001, 1e-3, 1e-2, 0.9, 0.999, 1e-8,
                               (C.c_float * 2)(20.0, 6.28))
    d = buf.ctypes.data
    call = lambda batch, n_upd=1, learner=lr, s=d: L.mrsim_ddpg_update(C.byref(learner) if learner else None, batch, n_upd, s, d, d, d,  # noqa: E731
                                                                      d, None, 0, 0, 0, None, None, None)
    assert call(64, learner=None) == _lib.EINVAL
    assert call(63) == _lib.EINVAL and call(32) == _lib.EINVAL and call(_lib.DDPG_MAX_BATCH + 64) == _lib.EINVAL
    assert call(64, n_upd=0) == _lib.EINVAL and call(64, s=None) == _lib.EINVAL
    lr2 = _lib.MrsimDdpgLearner(base + 4, base, base, base, base, base, base, 1e-5, 0.99, 0.001, 1e-3, 1e-2, 0.9, 0.999, 1e-8,
                                (C.c_float * 2)(20.0, 6.28))
    assert call(64, learner=lr2) == _lib.EALIGN
    assert call(64) in (_lib.ENODEVICE, _lib.OK)
    # replay push
    sc = (C.c_float * 5)(1, 1, 1, 1, 1)
    push = lambda n, cap, head: L.mrsim_replay_push(16, 4, d, d, d, d, d, sc, n, d, d, d, d, d, cap, head, 0, 0, None)  # noqa: E731
    assert push(8, 0, 0) == _lib.EINVAL and push(8, 4, 0) == _lib.EINVAL and push(2, 4, 4) == _lib.EINVAL and push(-1, 4, 0) == _lib.EINVAL
    assert push(0, 4, 0) == _lib.OK                     # nothing to do: no device needed
    assert push(2, 4, 0) in (_lib.ENODEVICE, _lib.OK)
    # device-side pack
    bd = (C.c_float * 2)(20.0, 6.28)
    assert L.mrsim_actor_pack_device(None, d, 1e-5, sc, bd, base, None) == _lib.EINVAL
    assert L.mrsim_actor_pack_device(d, d, 0.0, sc, bd, base, None) == _lib.EINVAL
    assert L.mrsim_actor_pack_device(d, d, 1e-5, sc, bd, base + 4, None) == _lib.EALIGN
    assert L.mrsim_actor_pack_device(d, d, 1e-5, sc, bd, base, None) in (_lib.ENODEVICE, _lib.OK)


def test_cu_mask_stream_entry_points_validate_their_arguments():
    from mr_rl_amd import _lib
    L = _lib.lib()
    cus, xccs, h = C.c_int32(0), C.c_int32(0), C.c_void_p()
    m = (C.c_uint32 * 8)(*([0xffffffff] * 8))
    assert L.mrsim_device_cu_layout(0, None, C.byref(xccs)) == _lib.EINVAL
    assert L.mrsim_device_cu_layout(-1, C.byref(cus), C.byref(xccs)) == _lib.EINVAL
    assert L.mrsim_stream_create_cu_mask(0, None, 8, C.byref(h)) == _lib.EINVAL
    assert L.mrsim_stream_create_cu_mask(0, m, 0, C.byref(h)) == _lib.EINVAL
    assert L.mrsim_stream_create_cu_mask(0, m, 8, None) == _lib.EINVAL
    assert L.mrsim_stream_destroy(None) == _lib.EINVAL
    assert L.mrsim_device_cu_layout(0, C.byref(cus), C.byref(xccs)) in (_lib.ENODEVICE, _lib.OK)
    rc = L.mrsim_stream_create_cu_mask(0, m, 8, C.byref(h))
    assert rc in (_lib.ENODEVICE, _lib.OK)
    if rc == _lib.OK:
        assert L.mrsim_stream_destroy(h) == _lib.OK


def test_partition_masks_keep_a_unit_of_every_xcc_on_both_sides():
    """mr_rl_amd.partition.partition_masks (host logic of CuPartition): complementary masks; mask bit i is a unit of XCC i % xccs"""
    from mr_rl_amd.partition import partition_masks
    for per_xcc in (1, 2, 4):
        learner, collect = partition_masks(256, 8, per_xcc)
        assert len(learner) == len(collect) == 8
        bits = lambda m: {32 * w + b for w in range(len(m)) for b in range(32) if (m[w] >> b) & 1}  # noqa: E731
        L, K = bits(learner), bits(collect)
        assert L | K == set(range(256)) and not (L & K) and len(L) == 8 * per_xcc
        for x in range(8):
            assert sum(1 for b in L if b % 8 == x) == per_xcc and sum(1 for b in K if b % 8 == x) == 32 - per_xcc
    with pytest.raises(ValueError):
        partition_masks(256, 8, 0)
    with pytest.raises(ValueError):
        partition_masks(256, 8, 17)


def test_host_wait_word_polls_a_host_word_without_a_device():
    """mrsim_host_wait_word is plain host code (the host side of MrsimStepIO.done_word): returns at once when the word holds the
    value, MRSIM_ETIMEOUT after the stated time when it does not, and sees a store made by another thread."""
    import ctypes as C
    import threading
    import time
    from mr_rl_amd import _lib
    L = _lib.lib()
    w = np.zeros(4, dtype=np.int32)
    p = C.c_void_p(w.ctypes.data)
    w[0] = 7
    assert L.mrsim_host_wait_word(p, 7, 0) == _lib.OK
    t0 = time.perf_counter()
    assert L.mrsim_host_wait_word(p, 8, 20000) == _lib.ETIMEOUT
    assert 0.015 < time.perf_counter() - t0 < 1.0
    assert L.mrsim_host_wait_word(None, 1, 10) == _lib.EINVAL and L.mrsim_host_wait_word(p, 1, -1) == _lib.EINVAL
    threading.Timer(0.05, lambda: w.__setitem__(0, 9)).start()
    assert L.mrsim_host_wait_word(p, 9, 5000000) == _lib.OK      # (ctypes releases the GIL during the call)
