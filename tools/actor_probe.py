#!/usr/bin/env python3
"""Per-launch timing (HIP events attached to the dispatch) of the fused rollout with the DDPG actor as its policy source,
for one or more libmrsim builds: python tools/actor_probe.py [--launches 200] [--envs 262144] [tag ...]
(tag = mr_rl_amd/variants/libmrsim_<tag>.so; no tag = the in-tree library).  Also usable under rocprofv3 --pmc."""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib
from mr_rl_amd.actor import DeviceActor
from mr_rl_amd.ddpg import Actor

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="*")
ap.add_argument("--launches", type=int, default=200)
ap.add_argument("--discard", type=int, default=50)
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--no-ou", action="store_true")
ap.add_argument("--math", default="f32", choices=["f32", "bf16x3", "bf16"])
ap.add_argument("--power", type=float, default=0.0, help="also run back to back for this many seconds and report package power / clock (hwmon)")
ap.add_argument("--streams", type=int, nargs="*", default=[], help="also time the RolloutCollector (wall clock) with these stream counts")
ap.add_argument("--mismatched", action="store_true")
ap.add_argument("--noise-law", default="collapsed", choices=["collapsed", "per_stage"], help="bench.py's default law is collapsed")
a = ap.parse_args()
T, WANT = 51, ("obs", "rew", "done", "actions")
torch.manual_seed(0)
module = Actor().eval()
flop = 2 * (5 * 64 + 64 * 64 + 64 * 2)
for tag in (a.variants or [None]):
    cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, is_mismatched=a.mismatched, noise_law=a.noise_law)
    e = MRVecEnv(a.envs, cfg=cfg, seed=7)
    if tag is not None:
        e._L = _lib.load(os.path.join(ROOT, "mr_rl_amd", "variants", f"libmrsim_{tag}.so"))
    act = DeviceActor.from_module(module, obs_scale=[0.01] * 5, device=e.device, ou=not a.no_ou, math=a.math)
    e.reset()
    buf = {}
    pool = [_lib.EventPair() for _ in range(a.launches)]
    for k in range(a.launches):
        e.rollout(T, want=WANT, out=buf, actor=act, carry="f64", events=pool[k])
    torch.cuda.synchronize()
    ms = [p.elapsed_ms() for p in pool][min(a.discard, a.launches - 1):]
    e.check_status()
    med = statistics.median(ms)
    print(f"{tag or 'in-tree':14s} median {med * 1e3:9.1f} us  min {min(ms) * 1e3:9.1f} us  {a.envs * T / med / 1e6:7.2f} G env-steps/s "
          f"in-kernel  actor {a.envs * T * flop / med / 1e9:6.1f} TFLOP/s (algorithmic, math={a.math})", flush=True)
    if a.power > 0:
        import threading, time
        import bench
        h = bench.hwmon_of(e.device)
        rd = lambda nm: int(open(os.path.join(h, nm)).read())  # noqa: E731
        samples, stop = [], [False]

        def sampler():
            while not stop[0]:
                samples.append((time.perf_counter(), rd("power1_input"), rd("freq1_input")))
                time.sleep(0.02)
        th = threading.Thread(target=sampler, daemon=True); th.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < a.power:
            for _ in range(50):
                e.rollout(T, want=WANT, out=buf, actor=act, carry="f64")
            torch.cuda.synchronize(); n += 50
        t1 = time.perf_counter(); stop[0] = True; th.join()
        load = [x for x in samples if t0 + 0.5 * (t1 - t0) < x[0] < t1]
        print(f"{'':14s} power leg: {sum(x[1] for x in load) / len(load) * 1e-6:7.1f} W of cap {rd('power1_cap') * 1e-6:.0f} W, driver sclk "
              f"{sum(x[2] for x in load) / len(load) * 1e-6:5.0f} MHz, {(t1 - t0) / n * 1e6:8.1f} us per launch over {t1 - t0:.1f} s", flush=True)
    for S in a.streams:
        import time
        from mr_rl_amd.collector import RolloutCollector
        col = RolloutCollector(a.envs, cfg=MRConfig(noise_var=1.0, auto_reset=True, seed=7, is_mismatched=a.mismatched, noise_law=a.noise_law), seed=7,
                               streams=S, policy=act)
        col.env._L = e._L
        col.reset()
        for k in range(40):
            col.collect(); col.ready(); col.release()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.launches):
            col.collect(); col.ready(); col.release()
        col.join(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{'':14s} collector streams={S}: {el / a.launches * 1e6:9.1f} us per episode  {a.envs * T * a.launches / el / 1e9:7.2f} G env-steps/s", flush=True)
