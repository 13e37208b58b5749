#!/usr/bin/env python3
"""Per-wave duration and placement (XCC / CU / SIMD / wave slot) of rollout launches, block by block.  Needs the instrumented build
    mkdir -p mr_rl_amd/variants && (cd mr_rl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
        -fno-slp-vectorize -DMRSIM_WAVE_PROBE -shared -I../../include -o ../variants/libmrsim_waveprobe.so mrsim_kernels.hip)
Run on the GPU box:  python tools/wave_time_probe.py [--noise-law collapsed] [--rounds 6] [--launches 300]
Every round = `launches` back-to-back one-stream launches with HIP events (median reported), then the per-wave record of the
last launch: wave durations by hardware wave slot, slots per SIMD, spread inside a SIMD."""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mr_rl_amd import MRConfig, MRVecEnv, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--noise-law", default="per_stage")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--launches", type=int, default=300)
ap.add_argument("--lib", default="waveprobe")
a = ap.parse_args()
N, T = 262144, 51
e = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True, noise_law=a.noise_law, rollout_carry="f64"), seed=7)
e._L = _lib.load(os.path.join(ROOT, "mr_rl_amd", "variants", f"libmrsim_{a.lib}.so"))
e.reset()
b = {}
WANT = ("obs", "rew", "done", "actions")
pool = [_lib.EventPair() for _ in range(a.launches)]
for r in range(a.rounds):
    for k in range(a.launches):
        e.rollout(T, actions=None, want=WANT, out=b, events=pool[k])
    ms = [p.elapsed_ms() for p in pool]
    torch.cuda.synchronize()
    fl = e.final_len.cpu().numpy().reshape(-1, 64)
    dur, xcc, hw, t0 = fl[:, 0].astype(np.int64), fl[:, 1] & 0xf, fl[:, 2], fl[:, 3].astype(np.int64)
    end = (t0 - t0.min()) + dur
    cyc = fl[:, 4].astype(np.int64)
    ghz = cyc / (dur * 10.0)          # cycles / ns
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; slot = hw & 0xf
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simdkey = key * 4 + simd
    us, inv = np.unique(simdkey, return_inverse=True)
    cntw = np.bincount(inv)
    mx = np.zeros(len(us)); mn = np.full(len(us), 1e18)
    np.maximum.at(mx, inv, end); np.minimum.at(mn, inv, end)
    # multiset of (slot & 3) per SIMD: how many SIMDs hold four DISTINCT priority phases
    phases = np.zeros((len(us), 4), dtype=np.int64)
    np.add.at(phases, (inv, slot & 3), 1)
    distinct = (phases > 0).sum(axis=1)
    print(f"round {r}: median {statistics.median(ms[50:])*1e3:7.1f} us last {ms[-1]*1e3:7.1f} us | waves {len(dur)} dur min/mean/max "
          f"{dur.min()/100:.1f}/{dur.mean()/100:.1f}/{dur.max()/100:.1f} us | SIMDs {len(us)} waves/SIMD {cntw.min()}..{cntw.max()} "
          f"| distinct (slot&3) per SIMD: " + " ".join(f"{d}:{(distinct == d).sum()}" for d in range(1, 5)) +
          f" | shader clock seen by the waves (s_memtime / s_memrealtime) mean {ghz.mean():.3f} min {ghz.min():.3f} max {ghz.max():.3f} GHz"
          f" | in-SIMD end spread mean {(mx - mn).mean()/100:.1f} max {(mx - mn).max()/100:.1f} us | slots used {sorted(set(slot.tolist()))}", flush=True)
    for sl in sorted(set(slot.tolist())):
        m = slot == sl
        print(f"    slot {sl:2d}: waves {m.sum():5d} start mean {(t0[m]-t0.min()).mean()/100:6.1f} dur mean {dur[m].mean()/100:7.1f} end mean {end[m].mean()/100:7.1f} us")
