"""ctypes binding of the CPU oracle (oracle/libmrsim_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (mr_rl_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRSIM_ORACLE_LIB: load another build of the same source instead (tests/test_oracle_sanitizers.py: the ASan / UBSan build)
_LIB_PATH = os.environ.get("MRSIM_ORACLE_LIB") or os.path.join(_HERE, "libmrsim_oracle.so")

INT_RK45, INT_EULER, INT_RK4 = 0, 1, 2
REW_CONSTANT10, REW_GOAL = 0, 1
NOISE_NONE, NOISE_PHILOX, NOISE_TAPE = 0, 1, 2
LAW_PER_STAGE, LAW_COLLAPSED = 0, 1
STREAM_DYN, STREAM_CTOR, STREAM_RESET_POS, STREAM_RESET_CTOR, STREAM_POLICY = 0, 1, 2, 3, 4


class OrcParams(C.Structure):
    _fields_ = [
        ("time_span", C.c_double), ("rtol", C.c_double), ("atol", C.c_double),
        ("a0", C.c_double), ("sigma", C.c_double), ("min_dist2goal", C.c_double),
        ("obs_low", C.c_double * 5), ("obs_high", C.c_double * 5),
        ("init_low", C.c_double * 2), ("init_high", C.c_double * 2),
        ("mismatched", C.c_int32), ("integrator", C.c_int32), ("substeps", C.c_int32),
        ("reward_mode", C.c_int32), ("max_timesteps", C.c_int32), ("auto_reset", C.c_int32),
        ("goal_K", C.c_int32), ("goal_T", C.c_int32), ("auto_reset_fresh_env", C.c_int32), ("noise_law", C.c_int32),
    ]


class OrcEnv(C.Structure):
    _fields_ = [
        ("y", C.c_double * 2), ("t", C.c_double), ("f", C.c_double * 2), ("h_abs", C.c_double),
        ("state_prime", C.c_double * 2), ("ep_ret", C.c_double), ("err_norm0", C.c_double), ("err_margin", C.c_double),
        ("counter", C.c_int32), ("n_rhs", C.c_int32), ("n_attempts", C.c_int32), ("status", C.c_int32),
    ]


ENV_DTYPE = np.dtype([
    ("y", "<f8", 2), ("t", "<f8"), ("f", "<f8", 2), ("h_abs", "<f8"), ("state_prime", "<f8", 2),
    ("ep_ret", "<f8"), ("err_norm0", "<f8"), ("err_margin", "<f8"), ("counter", "<i4"), ("n_rhs", "<i4"), ("n_attempts", "<i4"), ("status", "<i4"),
])


class OrcActor(C.Structure):
    _fields_ = [
        ("w1", C.c_float * 320), ("b1", C.c_float * 64), ("w2", C.c_float * 4096), ("b2", C.c_float * 64),
        ("w3", C.c_float * 128), ("b3", C.c_float * 2), ("bound", C.c_float * 2),
        ("ou_theta_dt", C.c_float), ("ou_sigma_sqrt_dt", C.c_float), ("ou_enabled", C.c_int32),
        ("ou_reset_on_done", C.c_int32), ("math", C.c_int32), ("reserved0", C.c_int32), ("w2_split", C.c_float * 12288),
        ("w1_split", C.c_float * 960),
    ]


class OrcNoise(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("seed", C.c_uint64), ("step_idx", C.c_uint64),
        ("tape", C.POINTER(C.c_double)), ("tape_len", C.c_int64), ("tape_pos", C.c_int64),
    ]


def build(force=False):
    if os.environ.get("MRSIM_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ("mrsim_oracle.c", "mrsim_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmrsim_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        assert L.orc_sizeof_env() == C.sizeof(OrcEnv) == ENV_DTYPE.itemsize, "OrcEnv layout mismatch"
        assert L.orc_sizeof_params() == C.sizeof(OrcParams), "OrcParams layout mismatch"
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_box_muller.argtypes = [C.c_uint32, C.c_uint32, fp, fp]
        L.orc_normals4.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, fp]
        L.orc_uniform2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, dp]
        L.orc_sim_reset.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcEnv), C.c_double, C.c_double, C.c_int,
                                    C.POINTER(OrcNoise), C.c_uint32, C.c_int]
        L.orc_sim_step.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcEnv), C.c_double, C.c_double,
                                   C.POINTER(OrcNoise), C.c_uint32]
        L.orc_sim_step.restype = C.c_int
        L.orc_env_step.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcEnv), fp, C.c_double, C.c_double,
                                   C.POINTER(OrcNoise), C.c_uint32, dp, dp, C.POINTER(C.c_uint8), dp, dp,
                                   C.POINTER(C.c_int32)]
        L.orc_env_step.restype = C.c_int
        L.orc_env_reset.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcEnv), fp, C.c_double, C.c_double, C.c_int,
                                    C.POINTER(OrcNoise), C.c_uint32, dp]
        L.orc_calculate_reward.argtypes = [C.POINTER(OrcParams), dp, C.c_int32]
        L.orc_calculate_reward.restype = C.c_double
        L.orc_vec_reset.argtypes = [C.POINTER(OrcParams), C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_vec_step.argtypes = [C.POINTER(OrcParams), C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vec_step.restype = C.c_int
        L.orc_vec_random_policy.argtypes = [C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_uint64, dp, dp,
                                            C.c_void_p, C.c_int]
        assert L.orc_sizeof_actor() == C.sizeof(OrcActor), "OrcActor layout mismatch"
        L.orc_spec_tanhf.argtypes = [C.c_float]
        L.orc_spec_tanhf.restype = C.c_float
        L.orc_actor_forward.argtypes = [C.POINTER(OrcActor), fp, fp]
        L.orc_actor_prepare.argtypes = [C.POINTER(OrcActor)]
        L.orc_vec_actor_policy.argtypes = [C.POINTER(OrcActor), C.c_int, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def default_params(**kw):
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        cur = getattr(p, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(p, k, v)
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_actor(w, ou=True, theta=0.15, sigma=0.3, dt=1e-2, reset_on_done=False, math="f32"):
    """OrcActor from the inference-form weights dict {w1 [64,5], b1, w2 [64,64], b2, w3 [2,64], b3, obs_scale [5],
    action_bound [2]} (mr_rl_amd.actor.fold_actor's output): the obs scaling is folded into w1 exactly as
    mrsim_actor_pack_host does (float32 product)."""
    lib()
    a = OrcActor()
    w1 = (np.asarray(w["w1"], np.float32) * np.asarray(w.get("obs_scale", np.ones(5)), np.float32)[None, :]).astype(np.float32)
    for name, arr in (("w1", w1), ("b1", w["b1"]), ("w2", w["w2"]), ("b2", w["b2"]), ("w3", w["w3"]), ("b3", w["b3"]),
                      ("bound", w["action_bound"])):
        flat = np.ascontiguousarray(arr, dtype=np.float32).ravel()
        dst = getattr(a, name)
        assert len(flat) == len(dst), name
        C.memmove(dst, flat.ctypes.data, flat.nbytes)
    a.ou_theta_dt = np.float32(np.float64(np.float32(theta)) * np.float64(np.float32(dt)))
    a.ou_sigma_sqrt_dt = np.float32(np.float64(np.float32(sigma)) * np.sqrt(np.float64(np.float32(dt))))
    a.ou_enabled, a.ou_reset_on_done = int(bool(ou)), int(bool(reset_on_done))
    a.math = {"f32": 0, "bf16x3": 1, "bf16": 2}[math]
    lib().orc_actor_prepare(C.byref(a))
    return a


def actor_forward(actor, obs):
    """[n,5] float32 -> [n,2] float32: ActorNetwork.predict, no noise."""
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    out = np.zeros((obs.shape[0], 2), dtype=np.float32)
    a0 = OrcActor.from_buffer_copy(actor)
    a0.ou_enabled = 0
    lib().orc_vec_actor_policy(C.byref(a0), INT_RK45, obs.shape[0], 0, _ptr(obs), None, None, 0, 0, _ptr(out), 1)
    return out


def actor_policy(actor, obs, ou, seed, step_idx, env_id0=0, counter=None, integrator=INT_RK45, threads=1):
    """actions [n,2] = actor.predict(obs) + actor_noise(); `ou` [n,2] float32 is updated in place."""
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    n = obs.shape[0]
    out = np.zeros((n, 2), dtype=np.float32)
    if ou is not None:
        assert ou.dtype == np.float32 and ou.shape == (n, 2) and ou.flags.c_contiguous
    cnt = None if counter is None else np.ascontiguousarray(counter, dtype=np.int32)
    lib().orc_vec_actor_policy(C.byref(actor), integrator, n, env_id0, _ptr(obs), _ptr(cnt), _ptr(ou), seed, step_idx,
                               _ptr(out), threads)
    return out


# ---------------------------------------------------------------------------
# RNG helpers
# ---------------------------------------------------------------------------
def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def box_muller(ua, ub):
    z0, z1 = C.c_float(), C.c_float()
    lib().orc_box_muller(ua, ub, C.byref(z0), C.byref(z1))
    return z0.value, z1.value


def c0(stream, block=0, call=0):
    return (stream << 28) | (block << 4) | call


def fill_normals(seed, env_id, step_idx, c0_start, ncalls):
    out = np.zeros(4 * ncalls, dtype=np.float32)
    lib().orc_fill_normals.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int64, C.c_void_p]
    lib().orc_fill_normals(seed, env_id, step_idx, c0_start, ncalls, _ptr(out))
    return out


def normals4(seed, env_id, step_idx, c0_):
    z = (C.c_float * 4)()
    lib().orc_normals4(seed, env_id, step_idx, c0_, z)
    return np.array(z[:], dtype=np.float32)


# ---------------------------------------------------------------------------
# Simulator-level driver (one env) used by the golden-vector tests
# ---------------------------------------------------------------------------
class Sim:
    """Mirror of MR_simulator.Simulator + the MR_env.reset ordering quirk."""

    def __init__(self, params, noise_kind=NOISE_NONE, tape=None, seed=0, env_id=0):
        self.p = params
        self.e = OrcEnv()
        self.nz = OrcNoise()
        self.nz.kind = noise_kind
        self.nz.seed = seed
        self.env_id = env_id
        self._tape = None
        if tape is not None:
            self._tape = np.ascontiguousarray(tape, dtype=np.float64)
            self.nz.tape = self._tape.ctypes.data_as(C.POINTER(C.c_double))
            self.nz.tape_len = len(self._tape)
            self.nz.tape_pos = 0

    def reset(self, x0, y0, ctor_mismatched=False, step_idx=0):
        self.nz.step_idx = step_idx
        lib().orc_sim_reset(C.byref(self.p), C.byref(self.e), x0, y0, int(ctor_mismatched), C.byref(self.nz),
                            self.env_id, STREAM_RESET_CTOR)

    def step(self, f_t, alpha_t, step_idx=0):
        self.nz.step_idx = step_idx
        rc = lib().orc_sim_step(C.byref(self.p), C.byref(self.e), f_t, alpha_t, C.byref(self.nz), self.env_id)
        if rc:
            raise RuntimeError(f"oracle: RK45 failed (rc={rc}) -- the reference would raise here")
        return np.array(self.e.y[:])


# ---------------------------------------------------------------------------
# batched env driver (n envs) -- parity checker for the HIP path, cpu_baseline
# ---------------------------------------------------------------------------
class VecOracle:
    def __init__(self, n, params, seed=0, env_id0=0, goal_table=None, threads=1):
        self.n, self.p, self.seed, self.env_id0, self.threads = n, params, seed, env_id0, threads
        self.envs = np.zeros(n, dtype=ENV_DTYPE)
        self.goal_table = None
        if goal_table is not None:
            self.goal_table = np.ascontiguousarray(goal_table, dtype=np.float32)
            assert self.goal_table.shape == (params.goal_K, params.goal_T, 2)
        self.obs = np.zeros((n, 5)); self.rew = np.zeros(n); self.done = np.zeros(n, dtype=np.uint8)
        self.final_obs = np.full((n, 5), np.nan); self.final_ret = np.full(n, np.nan)
        self.final_len = np.zeros(n, dtype=np.int32)

    def reset(self, step_idx, init_xy=None):
        if init_xy is not None:
            init_xy = np.ascontiguousarray(init_xy, dtype=np.float64)
            assert init_xy.shape == (self.n, 2)
        lib().orc_vec_reset(C.byref(self.p), self.n, self.env_id0, _ptr(self.envs), _ptr(self.goal_table),
                            _ptr(init_xy), self.seed, step_idx, _ptr(self.obs), self.threads)
        return self.obs.copy()

    def step(self, actions, step_idx):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        assert a.shape == (self.n, 2)
        rc = lib().orc_vec_step(C.byref(self.p), self.n, self.env_id0, _ptr(self.envs), _ptr(self.goal_table),
                                _ptr(a), self.seed, step_idx, _ptr(self.obs), _ptr(self.rew), _ptr(self.done),
                                _ptr(self.final_obs), _ptr(self.final_ret), _ptr(self.final_len), self.threads)
        if rc:
            raise RuntimeError("oracle: RK45 failed in at least one env")
        return self.obs, self.rew, self.done

    def random_policy(self, step_idx, lo, hi):
        a = np.zeros((self.n, 2), dtype=np.float32)
        lo_ = (C.c_double * 2)(*lo); hi_ = (C.c_double * 2)(*hi)
        lib().orc_vec_random_policy(self.p.integrator, self.n, self.env_id0, self.seed, step_idx, lo_, hi_, _ptr(a),
                                    self.threads)
        return a
