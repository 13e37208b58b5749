"""CPU: the checker itself is checked (sanitizers run on the CPU restatement; GPU ASan is not available on
this pool).  oracle/libmrsim_oracle_asan.so -- the same mrsim_oracle.c built with -fsanitize=address,undefined
(-fno-sanitize-recover: the first undefined operation aborts) -- replays, in a child process with libasan preloaded:
golden trajectories (sigma = 0 and a taped sigma > 0 run, near the origin where rk_step attempts are rejected and the
noise stream draws hundreds of values per step), the re-used-env auto-reset golden, a Philox-noise vec run that starts
on the origin, the env-level paths (goal reward, goal table, auto-reset, out-of-bounds), and the actor policy."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import numpy as np
from oracle import oracle as O
from tests.util import load_cases
import tests.test_oracle_golden as TG

# golden replays (assertions inside): all sigma = 0 simulator cases, the taped noisy ones, env episodes, re-used env
for name in sorted(TG.SIM):
    TG.test_simulator_sigma0(name)
for name in sorted(TG.NOISE):
    TG.test_simulator_noise_tape(name)
for name in sorted(TG.REUSED):
    TG.test_auto_reset_is_the_reused_env_object(name)
TG.test_known_answers_from_survey()

# Philox noise from the origin (step splitting, rejected attempts), mismatched law, goal table + goal reward + auto-reset
tab = np.zeros((3, 12, 2), dtype=np.float32); tab[1] += 0.5; tab[2, :, 0] = np.arange(12)
for mis in (0, 1):
    p = O.default_params(sigma=1.0, mismatched=mis, auto_reset=1, reward_mode=O.REW_GOAL, max_timesteps=9, min_dist2goal=0.4,
                         init_low=[-0.01, -0.01], init_high=[0.01, 0.01], goal_K=3, goal_T=12)
    v = O.VecOracle(1000, p, seed=5, env_id0=4000000000 - 7, goal_table=tab, threads=4)   # env ids up to the uint32 range
    v.reset(0)
    for t in range(25):
        a = v.random_policy(t + 1, [-20, -6.3], [20, 6.3])
        if t % 5 == 0:
            a[:50] = 0.0
        v.step(a, step_idx=t + 1)
    assert np.isfinite(v.envs["y"]).all() and v.envs["n_attempts"].max() > 1      # attempts were rejected on the way
# out of bounds + fixed-step modes
for integ in (O.INT_EULER, O.INT_RK4):
    p = O.default_params(sigma=0.5, integrator=integ, substeps=7, init_low=[4999.0, -4999.0], init_high=[5000.0, -4998.0])
    v = O.VecOracle(64, p, seed=1, threads=1)
    v.reset(0)
    for t in range(6):
        v.step(np.tile(np.float32([20.0, 0.0]), (64, 1)), step_idx=t + 1)
    assert v.done.any()
# actor policy
rng = np.random.default_rng(0)
w = {"w1": rng.normal(size=(64, 5)), "b1": rng.normal(size=64), "w2": rng.normal(size=(64, 64)) / 8, "b2": rng.normal(size=64),
     "w3": rng.normal(size=(2, 64)) / 8, "b3": rng.normal(size=2), "obs_scale": np.full(5, 0.01), "action_bound": [20.0, 6.2831853]}
A = O.make_actor(w, ou=True, reset_on_done=True)
obs = rng.uniform(-200, 200, (333, 5)).astype(np.float32)
ou = np.zeros((333, 2), dtype=np.float32)
for t in range(5):
    act = O.actor_policy(A, obs, ou, 9, t, env_id0=12345, counter=np.arange(333) % 3, threads=3)
assert np.isfinite(act).all() and np.abs(act[:, 0]).max() <= 20 + 1.0
print("SANITIZED-ORACLE-OK")
'''


def test_oracle_is_clean_under_asan_and_ubsan():
    lib = os.path.join(ROOT, "oracle", "libmrsim_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libmrsim_oracle_asan.so"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "gcc has no libasan here"
    env = dict(os.environ, MRSIM_ORACLE_LIB=lib, LD_PRELOAD=libasan, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0, report
    assert "SANITIZED-ORACLE-OK" in r.stdout, report
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, report
