#!/usr/bin/env python3
"""Times the REFERENCE's own Python path (MR_Env.step loop, unmodified files under /root/reference) on this
container's cores -- BASELINE.md section 4, item 1.  Build-container only (the reference never travels to the
GPU box).  Workload = BASELINE config 4 per env: reset(), 51 steps of uniform actions in the actor range,
sigma = 1.  One process per core; `gym`, `turtle`, `tkinter` are absent and are stood in exactly as in
tests/golden/make_golden.py."""
import contextlib
import io
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))


def worker(args):
    seed, episodes = args
    import numpy as np
    import make_golden as mg
    mg._install_standins()
    import MR_env
    np.random.seed(seed)
    rng = np.random.default_rng(seed)
    env = MR_env.MR_Env()
    steps = 0
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(episodes):
            env.reset()
            done = False
            while not done:
                a = np.array([rng.uniform(-20, 20), rng.uniform(-2 * np.pi, 2 * np.pi)])
                _, _, done, _ = env.step(a)
                steps += 1
    return steps, time.perf_counter() - t0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    import json
    import numpy
    import scipy
    cores = len(os.sched_getaffinity(0))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    episodes = int(args[0]) if args else 40
    s1, t1 = worker((0, episodes))
    print(f"1 process : {s1} env-steps in {t1:.2f} s = {s1 / t1:,.0f} env-steps/s")
    t0 = time.perf_counter()
    with mp.Pool(cores) as pool:
        res = pool.map(worker, [(k + 1, episodes) for k in range(cores)])
    wall = time.perf_counter() - t0
    tot = sum(r[0] for r in res)
    rate_all = sum(r[0] / r[1] for r in res)
    print(f"{cores} processes: {tot} env-steps, sum of per-process rates = {rate_all:,.0f} env-steps/s "
          f"(wall incl. imports {wall:.1f} s)")
    out = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--json=")]
    if out:   # the record bench.py quotes in cpu_baseline.reference_python (never run on the GPU box: the reference stays here)
        json.dump({"what": "the reference's own Python path, unmodified files: MR_Env.reset() then MR_Env.step() until done "
                           "(51 steps), uniform actions in the DDPG actor range, sigma = 1 (BASELINE config 4 per env), one "
                           "process per core; gym / turtle / tkinter stood in as in tests/golden/make_golden.py",
                   "source": "tools/ref_python_baseline.py %d" % episodes, "where": "build container (no GPU)",
                   "cpu_model": cpu_model(), "cores": cores, "python": sys.version.split()[0], "numpy": numpy.__version__,
                   "scipy": scipy.__version__, "episodes_per_process": episodes, "unit": "env-steps/s",
                   "value_one_process": round(s1 / t1, 1), "value_all_processes": round(rate_all, 1),
                   "processes": cores}, open(out[0], "w"), indent=1)
        print("wrote", out[0])
