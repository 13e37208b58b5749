#!/usr/bin/env python3
"""Package power (hwmon power1_input, PPT) and shader clock (freq1_input) of the GPU while the fused-rollout workload
runs for ~2 s (RolloutCollector, 262 144 envs, two sub-shard streams), sampled every 20 ms from sysfs.  The sensor of the
GPU this process runs on is the one whose power rises.  Answers "why does the kernel run at 1.84 GHz of 2.4": it sits at
the package power limit.  Usage (GPU box): python tools/power_probe.py [--workload ddpg|mixed] [--lib tag]"""
import argparse, glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig
from mr_rl_amd.collector import RolloutCollector

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ddpg")
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--obs-layout", default="aos")
ap.add_argument("--T", type=int, default=51, help="steps per launch group")
ap.add_argument("--depth", type=int, default=2, help="rotating [T, N, ...] buffer sets")
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--noise-law", default="per_stage")
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--batch", type=int, default=400, help="launch groups between two synchronisations")
ap.add_argument("--series", action="store_true", help="print every batch: us per group, package power, sclk / mclk readings")
ap.add_argument("--lib", default=None, help="an A/B build: mr_rl_amd/variants/libmrsim_<tag>.so (make -C mr_rl_amd/csrc variants)")
a = ap.parse_args()
mons = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(h + "/power1_input")]


def rd(path):
    try:
        return int(open(path).read())
    except Exception:
        return -1


cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, obs_layout=a.obs_layout, noise_law=a.noise_law)
tab = None
if a.workload == "mixed":
    import bench
    tab = bench.mixed_goal_table(cfg, 7)
col = RolloutCollector(a.envs, cfg=cfg, seed=7, streams=a.streams, goal_table=tab, T=a.T, depth=a.depth)
if a.lib:
    from mr_rl_amd import _lib
    col.env._L = _lib.load(os.path.join(ROOT, "mr_rl_amd", "variants", f"libmrsim_{a.lib}.so"))
col.reset()
samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(), [rd(m + "/power1_input") for m in mons], [rd(m + "/freq1_input") for m in mons],
                        [rd(m + "/freq2_input") for m in mons]))
        time.sleep(0.02)


th = threading.Thread(target=sampler)
th.start()
time.sleep(0.3)
t0 = time.perf_counter()
n = 0
series = []
while time.perf_counter() - t0 < a.seconds:
    tb = time.perf_counter()
    for _ in range(a.batch):
        col.collect()
    torch.cuda.synchronize()
    n += a.batch
    series.append((tb, time.perf_counter()))
t1 = time.perf_counter()
time.sleep(0.3)
stop = True
th.join()
col.check_status()
load = [s for s in samples if t0 + 0.5 * (t1 - t0) < s[0] < t1]
idle = [s for s in samples if s[0] < t0 - 0.05]
j = max(range(len(mons)), key=lambda k: sum(s[1][k] for s in load))
W = lambda ss: sum(s[1][j] for s in ss) / len(ss) * 1e-6
F = lambda ss: sum(s[2][j] for s in ss) / len(ss) * 1e-6
mb = a.depth * a.T * a.envs * 33 / 1e6
print(f"workload {a.workload} law {a.noise_law} streams {a.streams}{' [' + a.lib + ']' if a.lib else ''} obs {a.obs_layout}, {a.envs} envs, {a.T} steps per launch group, "
      f"{a.depth} buffer set(s) = {mb:.0f} MB of transition buffers: {n} groups in {t1 - t0:.3f} s = {(t1 - t0) / n * 1e6:.1f} us per group "
      f"({a.envs * a.T * n / (t1 - t0) / 1e9:.1f} G env-steps/s)")
if a.series:
    for tb, te in series:
        ss = [x for x in samples if tb <= x[0] <= te] or [min(samples, key=lambda x: abs(x[0] - te))]
        print(f"  t={tb - t0:6.3f}s  {(te - tb) / a.batch * 1e6:7.1f} us/group  {sum(x[1][j] for x in ss) / len(ss) * 1e-6:7.1f} W  "
              f"sclk {sum(x[2][j] for x in ss) / len(ss) * 1e-6:5.0f}  mclk {sum(x[3][j] for x in ss) / len(ss) * 1e-6:5.0f} MHz")
print(f"sensor {mons[j]}: power cap {rd(mons[j] + '/power1_cap') * 1e-6:.0f} W")
print(f"  before the load : {W(idle):7.1f} W   sclk {F(idle):6.0f} MHz")
print(f"  under load      : {W(load):7.1f} W   sclk {F(load):6.0f} MHz   (mean of the second half of the load window, "
      f"{len(load)} samples; sclk is the driver's averaged reading)")
