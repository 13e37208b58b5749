"""ctypes binding of libmrsim.so (include/mrsim.h).  There is no fallback: if the
HIP library is missing or a call fails, this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmrsim.so")

OK, EINVAL, ENODEVICE, ELAUNCH, EALIGN, ERANGE, ETIMEOUT = 0, -1, -2, -3, -4, -5, -6
INT_RK45, INT_EULER, INT_RK4 = 0, 1, 2
REW_CONSTANT10, REW_GOAL = 0, 1
OBS_AOS, OBS_SOA = 0, 1
NOISE_FAST, NOISE_SPEC = 0, 1
LAW_PER_STAGE, LAW_COLLAPSED = 0, 1
ABI_VERSION = 5
ACTOR_HIDDEN, ACTOR_BLOB_FLOATS = 64, 10888
ACTOR_F32, ACTOR_BF16X3, ACTOR_BF16 = 0, 1, 2
DDPG_PARAMS, DDPG_MAX_BATCH = 7680, 4096


def ddpg_batch_scratch_floats(batch):
    """MRSIM_DDPG_BATCH_SCRATCH_FLOATS(batch) of include/mrsim.h"""
    return (int(batch) // 64) * (DDPG_PARAMS + 4) + int(batch) + 64

# every symbol include/mrsim.h declares -- the product ABI (tests check the .so exports exactly these + BENCH_SYMBOLS)
PRODUCT_SYMBOLS = (
    "mrsim_abi_version", "mrsim_strerror", "mrsim_default_params", "mrsim_reset", "mrsim_step",
    "mrsim_random_policy", "mrsim_random_policy_steps", "mrsim_rollout", "mrsim_advance_step_base", "mrsim_device_cu_layout",
    "mrsim_stream_create_cu_mask", "mrsim_stream_destroy", "mrsim_velocity",
    "mrsim_device_count", "mrsim_device_name",
    "mrsim_actor_fold_bn_host", "mrsim_actor_pack_host", "mrsim_actor_forward", "mrsim_ddpg_update",
    "mrsim_replay_push", "mrsim_replay_add_step", "mrsim_actor_pack_device",
    "mrsim_host_alloc", "mrsim_host_wait_word", "mrsim_host_free", "mrsim_stream_synchronize",
)
# include/mrsim_bench.h: measurement and test aids (bench.py, tools/, tests/); nothing in mr_rl_amd's product path calls them
BENCH_SYMBOLS = (
    "mrsim_step_timed", "mrsim_rollout_timed", "mrsim_event_create", "mrsim_event_destroy", "mrsim_event_elapsed_ms",
    "mrsim_rollout_events", "mrsim_step_events", "mrsim_debug_normals",
)
SYMBOLS = PRODUCT_SYMBOLS + BENCH_SYMBOLS


class MrsimParams(C.Structure):
    _fields_ = [
        ("time_span", C.c_double), ("rtol", C.c_double), ("atol", C.c_double), ("a0", C.c_double),
        ("sigma", C.c_double), ("min_dist2goal", C.c_double),
        ("obs_low", C.c_double * 5), ("obs_high", C.c_double * 5),
        ("init_low", C.c_double * 2), ("init_high", C.c_double * 2),
        ("act_low", C.c_double * 2), ("act_high", C.c_double * 2),
        ("mismatched", C.c_int32), ("integrator", C.c_int32), ("substeps", C.c_int32),
        ("reward_mode", C.c_int32), ("max_timesteps", C.c_int32), ("auto_reset", C.c_int32),
        ("goal_K", C.c_int32), ("goal_T", C.c_int32), ("obs_layout", C.c_int32), ("noise_math", C.c_int32),
        ("auto_reset_fresh_env", C.c_int32), ("noise_law", C.c_int32),
        ("step_base", C.c_void_p),
    ]


class MrsimState(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("aux", C.c_void_p), ("ep_ret", C.c_void_p)]


class MrsimActorWeights(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p),
                ("b3", C.c_void_p), ("obs_scale", C.c_float * 5), ("action_bound", C.c_float * 2)]


class MrsimActor(C.Structure):
    _fields_ = [("blob", C.c_void_p), ("ou_state", C.c_void_p), ("ou_theta", C.c_float), ("ou_sigma", C.c_float),
                ("ou_dt", C.c_float), ("ou_reset_on_done", C.c_int32), ("math", C.c_int32), ("reserved0", C.c_int32)]


class MrsimDdpgLearner(C.Structure):
    _fields_ = [("online", C.c_void_p), ("target", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
                ("grad_scratch", C.c_void_p), ("steps", C.c_void_p), ("bn_stats", C.c_void_p), ("bn_eps", C.c_float),
                ("gamma", C.c_float), ("tau", C.c_float), ("actor_lr", C.c_float), ("critic_lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("adam_eps", C.c_float), ("action_bound", C.c_float * 2),
                ("batch_scratch", C.c_void_p), ("batch_scratch_floats", C.c_int64),
                ("actor_blob", C.c_void_p), ("actor_obs_scale", C.c_float * 5), ("reserved0", C.c_int32)]


class MrsimReplaySink(C.Structure):
    _fields_ = [("s", C.c_void_p), ("a", C.c_void_p), ("r", C.c_void_p), ("done", C.c_void_p), ("s2", C.c_void_p),
                ("ended2", C.c_void_p), ("capacity", C.c_int32), ("head", C.c_int32), ("obs_scale", C.c_float * 5),
                ("reserved0", C.c_int32)]


class MrsimStepIO(C.Structure):
    _fields_ = [
        ("actions", C.c_void_p), ("actions_out", C.c_void_p), ("goal_table", C.c_void_p),
        ("obs", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p), ("state_prime", C.c_void_p),
        ("final_obs", C.c_void_p), ("final_ret", C.c_void_p), ("final_len", C.c_void_p),
        ("status", C.c_void_p), ("actor", C.POINTER(MrsimActor)), ("attempts", C.c_void_p),
        ("replay", C.POINTER(MrsimReplaySink)), ("done_word", C.c_void_p), ("done_value", C.c_int32), ("reserved0", C.c_int32),
    ]


class MrsimRolloutIO(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("shared_actions", C.c_int32), ("actions", C.c_void_p), ("goal_table", C.c_void_p),
        ("traj_xy", C.c_void_p), ("state_prime_T", C.c_void_p), ("obs_T", C.c_void_p), ("rew_T", C.c_void_p),
        ("done_T", C.c_void_p), ("actions_out_T", C.c_void_p), ("final_ret", C.c_void_p),
        ("final_len", C.c_void_p), ("status", C.c_void_p),
        ("row_stride", C.c_int64), ("carry_f64", C.c_int32), ("actions_f64", C.c_int32),
        ("actor", C.POINTER(MrsimActor)),
    ]


class MrsimError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: {strerror(code)} (MRSIM code {code})")


_lib = None


def load(path):
    """Bind one libmrsim build (the default in-tree one, or an A/B variant for tools/ab_rollout.py)."""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C mr_rl_amd/csrc`. mr_rl_amd has no CPU fallback.")
    # One HIP runtime per process.  torch ships its own libamdhip64 / libhsa-runtime64; dlopen'ing libmrsim.so BEFORE torch
    # maps /opt/rocm's copies instead, torch then maps its own next to them and finds no device (seen on the MI355X box:
    # build() followed by smoke() in one process).  With torch imported first, libmrsim's NEEDED libamdhip64 resolves to the
    # copy that is already mapped.  (A C / C++ consumer without torch links /opt/rocm's runtime: examples/abi_demo.cpp.)
    import torch  # noqa: F401
    L = C.CDLL(path)
    vp, i64, u32, u64, i32 = C.c_void_p, C.c_int64, C.c_uint32, C.c_uint64, C.c_int32
    PP, PS, PIO = C.POINTER(MrsimParams), C.POINTER(MrsimState), C.POINTER(MrsimStepIO)
    L.mrsim_abi_version.restype = C.c_int
    L.mrsim_strerror.argtypes = [C.c_int]
    L.mrsim_strerror.restype = C.c_char_p
    L.mrsim_default_params.argtypes = [PP]
    L.mrsim_reset.argtypes = [PP, i64, u32, PS, vp, vp, vp, vp, i32, u64, u64, vp]
    L.mrsim_step.argtypes = [PP, i64, u32, PS, PIO, u64, u64, vp]
    L.mrsim_step_timed.argtypes = [PP, i64, u32, PS, PIO, u64, u64, vp, C.POINTER(C.c_float)]
    L.mrsim_random_policy.argtypes = [PP, i64, u32, vp, u64, u64, vp]
    L.mrsim_random_policy_steps.argtypes = [PP, i64, u32, vp, C.c_int32, u64, u64, vp]
    L.mrsim_rollout.argtypes = [PP, i64, u32, PS, C.POINTER(MrsimRolloutIO), u64, u64, vp]
    L.mrsim_rollout_timed.argtypes = L.mrsim_rollout.argtypes + [C.POINTER(C.c_float)]
    L.mrsim_velocity.argtypes = [i64, i32, i32, vp, vp, vp, vp, vp, vp]
    L.mrsim_advance_step_base.argtypes = [vp, u64, vp]
    L.mrsim_device_cu_layout.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.mrsim_stream_create_cu_mask.argtypes = [C.c_int32, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_void_p)]
    L.mrsim_stream_destroy.argtypes = [vp]
    L.mrsim_event_create.argtypes = [C.POINTER(vp)]
    L.mrsim_event_destroy.argtypes = [vp]
    L.mrsim_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.mrsim_rollout_events.argtypes = L.mrsim_rollout.argtypes + [vp, vp]
    L.mrsim_step_events.argtypes = L.mrsim_step.argtypes + [vp, vp]
    L.mrsim_debug_normals.argtypes = [i64, u32, u64, u64, u32, i32, vp, vp]
    L.mrsim_actor_fold_bn_host.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp]
    L.mrsim_actor_pack_host.argtypes = [C.POINTER(MrsimActorWeights), vp]
    L.mrsim_actor_forward.argtypes = [PP, i64, u32, C.POINTER(MrsimActor), PS, vp, vp, u64, u64, vp]
    L.mrsim_ddpg_update.argtypes = [C.POINTER(MrsimDdpgLearner), i32, i32, vp, vp, vp, vp, vp, vp, i32, u64, u64, vp, vp, vp]
    L.mrsim_replay_push.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, u64, u64, vp]
    L.mrsim_actor_pack_device.argtypes = [vp, vp, C.c_float, vp, vp, vp, vp]
    L.mrsim_replay_add_step.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.mrsim_host_alloc.argtypes = [i64, C.POINTER(vp), C.POINTER(vp)]
    L.mrsim_host_free.argtypes = [vp]
    L.mrsim_stream_synchronize.argtypes = [vp]
    L.mrsim_host_wait_word.argtypes = [vp, i32, i64]
    L.mrsim_device_count.restype = C.c_int
    L.mrsim_device_name.argtypes = [C.c_int, C.c_char_p, i32]
    for name in SYMBOLS:
        if name != "mrsim_strerror":
            getattr(L, name).restype = C.c_int
    if L.mrsim_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI {L.mrsim_abi_version()} != binding ABI {ABI_VERSION}")
    assert C.sizeof(MrsimParams) == 8 * 6 + 8 * 18 + 4 * 12 + 8
    assert C.sizeof(MrsimRolloutIO) == 8 + 8 * 11 + 8 + 4 + 4 + 8
    assert C.sizeof(MrsimStepIO) == 8 * 16 and C.sizeof(MrsimReplaySink) == 8 * 6 + 4 * 8
    assert C.sizeof(MrsimDdpgLearner) == 144
    assert C.sizeof(MrsimActor) == 40 and C.sizeof(MrsimActorWeights) == 48 + 28 + 4
    return L


def lib():
    """Load libmrsim.so (built by `make -C mr_rl_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


def strerror(code):
    return lib().mrsim_strerror(int(code)).decode()


def check(code, what):
    if code != OK:
        raise MrsimError(code, what)


class EventPair:
    """Two HIP events attached to one dispatch by mrsim_rollout_events / mrsim_step_events (measurement aid):
    elapsed_ms() waits for the stop event and returns that kernel's duration."""

    def __init__(self):
        self.start, self.stop = C.c_void_p(), C.c_void_p()
        check(lib().mrsim_event_create(C.byref(self.start)), "mrsim_event_create")
        check(lib().mrsim_event_create(C.byref(self.stop)), "mrsim_event_create")

    def elapsed_ms(self):
        ms = C.c_float(0.0)
        check(lib().mrsim_event_elapsed_ms(self.start, self.stop, C.byref(ms)), "mrsim_event_elapsed_ms")
        return ms.value

    def close(self):
        for e in (self.start, self.stop):
            if e:
                lib().mrsim_event_destroy(e)
        self.start, self.stop = C.c_void_p(), C.c_void_p()


def default_params():
    p = MrsimParams()
    check(lib().mrsim_default_params(C.byref(p)), "mrsim_default_params")
    return p
