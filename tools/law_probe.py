#!/usr/bin/env python3
"""Per-stage vs collapsed noise law (MrsimParams.noise_law) on the fused rollout kernel: interleaved A/B timing in ONE
process with per-launch HIP events, long blocks whose first launches are discarded (the chip settles its clock per
variant; cdna_hip_programming.md rule 24).  Same measurement idiom as tools/ab_rollout.py, but the variants are
configurations of the in-tree library, not builds.

Usage (GPU box):  python tools/law_probe.py [--workload ddpg|mixed] [--mismatched] [--carry f64] [--launches 400] [--rounds 4]
"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--launches", type=int, default=400)
ap.add_argument("--discard", type=int, default=100)
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--workload", default="ddpg")
ap.add_argument("--mismatched", action="store_true")
ap.add_argument("--carry", default="f64")
ap.add_argument("--laws", default="per_stage,collapsed")
ap.add_argument("--obs-layout", default="aos")
ap.add_argument("--verbose", action="store_true", help="per-block quantiles")
a = ap.parse_args()
T, WANT = 51, ("obs", "rew", "done", "actions")
envs = []
for law in a.laws.split(","):
    cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, is_mismatched=a.mismatched, rollout_carry=a.carry, noise_law=law, obs_layout=a.obs_layout)
    tab = None
    if a.workload == "mixed":
        import bench
        tab = bench.mixed_goal_table(cfg, 7)
    e = MRVecEnv(a.envs, cfg=cfg, seed=7, goal_table=tab)
    e.reset()
    envs.append((law, e, {}, []))
for _ in range(150):
    for v, e, b, ms in envs:
        e.rollout(T, want=WANT, out=b)
torch.cuda.synchronize()
pool = [_lib.EventPair() for _ in range(a.launches)]
for r in range(a.rounds):
    order = envs if r % 2 == 0 else envs[::-1]
    for v, e, b, ms in order:
        for k in range(a.launches):
            e.rollout(T, want=WANT, out=b, events=pool[k])
        blk = [p.elapsed_ms() for p in pool]
        ms.extend(blk[min(a.discard, a.launches - 1):])
        if a.verbose:
            q = sorted(blk[min(a.discard, a.launches - 1):])
            pc = lambda f: q[int(f * (len(q) - 1))] * 1e3
            print(f"  round {r} {v:10s} first20 {statistics.mean(blk[:20])*1e3:7.1f}  p10 {pc(.1):7.1f} p25 {pc(.25):7.1f} p50 {pc(.5):7.1f} "
                  f"p75 {pc(.75):7.1f} p90 {pc(.9):7.1f} us", flush=True)
print(f"# N={a.envs} T={T} workload={a.workload} mismatched={a.mismatched} carry={a.carry} rounds={a.rounds} x launches={a.launches} "
      f"(first {a.discard} of each block discarded; blocks interleaved in one process)")
base = statistics.median(envs[0][3])
for v, e, b, ms in envs:
    e.check_status()
    med, mn = statistics.median(ms), min(ms)
    print(f"{v:10s} median {med*1e3:8.2f} us  mean {statistics.mean(ms)*1e3:8.2f} us  min {mn*1e3:8.2f} us  vs first {base/med:6.3f}x  "
          f"{a.envs*T/med/1e6:7.2f} G env-steps/s in-kernel  mean|pos| {float(e.pos.abs().mean()):.3f}", flush=True)
