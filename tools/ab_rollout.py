#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants of the fused rollout (cdna_hip_programming.md rule 24: N variants x M rounds
in ONE process, per-launch HIP events, median and min reported), in blocks of several hundred back-to-back launches
whose first hundred are discarded: the chip lowers its clock under this load within tens of ms (DVFS), and by a
different amount for variants that draw different power.  Variants are libmrsim builds with different -D
switches (make -C mr_rl_amd/csrc variants -> mr_rl_amd/variants/libmrsim_<tag>.so), optionally ":f64" for the fp64 carry.
Usage (GPU box):  python tools/ab_rollout.py [--rounds 10] [--launches 30] [--workload ddpg|mixed] tag[:f64] ...
Also prints max |pos difference| of every variant against the first one after the same number of steps."""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--launches", type=int, default=400,
                help="back-to-back launches per variant and round; long blocks so that each variant runs at the clock the "
                     "chip holds for IT (short interleaved blocks let a power-hungrier variant ride the previous one's clock)")
ap.add_argument("--discard", type=int, default=100, help="leading launches of every block left out of the statistics")
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--workload", default="ddpg")
ap.add_argument("--mismatched", action="store_true")
ap.add_argument("--noise-law", default="per_stage")
a = ap.parse_args()
T, WANT = 51, ("obs", "rew", "done", "actions")
envs = []
for v in a.variants:
    tag, _, carry = v.partition(":")
    cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, is_mismatched=a.mismatched, rollout_carry=carry or "f32", noise_law=a.noise_law)
    tab = None
    if a.workload == "mixed":
        sys.path.insert(0, ROOT)
        import bench
        tab = bench.mixed_goal_table(cfg, 7)
    e = MRVecEnv(a.envs, cfg=cfg, seed=7, goal_table=tab)
    e._L = _lib.load(os.path.join(ROOT, "mr_rl_amd", "variants", f"libmrsim_{tag}.so"))
    e.reset()
    envs.append((v, e, {}, []))
for _ in range(150):  # settle the clocks (~35 ms of load) on every variant
    for v, e, b, ms in envs:
        e.rollout(T, want=WANT, out=b)
torch.cuda.synchronize()
pool = [_lib.EventPair() for _ in range(a.launches)]
for r in range(a.rounds):
    order = envs if r % 2 == 0 else envs[::-1]
    for v, e, b, ms in order:
        for k in range(a.launches):
            e.rollout(T, want=WANT, out=b, events=pool[k])
        ms.extend([p.elapsed_ms() for p in pool][min(a.discard, a.launches - 1):])
ref = envs[0][1].pos.clone()
print(f"# N={a.envs} T={T} workload={a.workload} mismatched={a.mismatched} rounds={a.rounds} x launches={a.launches} "
      f"(first {a.discard} of each block discarded; blocks interleaved in one process)")
base = statistics.median(envs[0][3])
for v, e, b, ms in envs:
    e.check_status()
    med, mn = statistics.median(ms), min(ms)
    d = float((e.pos - ref).abs().max())
    print(f"{v:14s} median {med*1e3:8.2f} us  min {mn*1e3:8.2f} us  vs first {base/med:6.3f}x  "
          f"{a.envs*T/med/1e6:7.2f} G env-steps/s in-kernel  max|pos - first| {d:.3e}", flush=True)
