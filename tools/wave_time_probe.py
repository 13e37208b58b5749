#!/usr/bin/env python3
"""Per-wave duration and placement (XCC / CU / SIMD / wave slot) of one rollout launch.  Needs the instrumented build:
    make -C mr_rl_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DMRSIM_WAVE_PROBE"
(rebuild without the flag afterwards).  Run on the GPU box."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from mr_rl_amd import MRConfig, MRVecEnv
N, T = 262144, 51
e = MRVecEnv(N, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=7)
e.reset()
b = {}
for _ in range(400):
    e.rollout(T, actions=None, want=("obs", "rew", "done", "actions"), out=b)
torch.cuda.synchronize()
fl = e.final_len.cpu().numpy().reshape(-1, 64)
dur, xcc, hw, t0 = fl[:, 0].astype(np.int64), fl[:, 1] & 0xf, fl[:, 2], fl[:, 3].astype(np.int64)
print("waves", len(dur), "clock ticks (100 MHz): dur min/mean/max", dur.min(), dur.mean(), dur.max())
end = (t0 - t0.min()) + dur
print("start spread", (t0 - t0.min()).max(), "end min/mean/max", end.min(), end.mean(), end.max())
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: waves {m.sum():5d} dur mean {dur[m].mean():9.1f} max {dur[m].max():7d} end mean {end[m].mean():9.1f} max {end[m].max()}")
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
key = xcc * 1000 + se * 100 + sh * 50 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs", len(u), "waves per CU min/max", cnt.min(), cnt.max())
per = np.array([dur[key == k].mean() for k in u])
print("per-CU mean dur: min", per.min(), "median", np.median(per), "max", per.max())
simdkey = key * 4 + simd
us, inv = np.unique(simdkey, return_inverse=True)
mx = np.zeros(len(us)); mn = np.full(len(us), 1e18); cntw = np.zeros(len(us))
np.maximum.at(mx, inv, end); np.minimum.at(mn, inv, end); np.add.at(cntw, inv, 1)
print("SIMDs", len(us), "waves per SIMD min/max", cntw.min(), cntw.max())
print("per-SIMD last-wave end: min/mean/max", mx.min(), mx.mean(), mx.max(), " first-wave end: min/mean/max", mn.min(), mn.mean(), mn.max())
print("within-SIMD spread (max-min): mean", (mx - mn).mean(), "max", (mx - mn).max())
slot = hw & 0xf
for sl in range(int(slot.max()) + 1):
    m = slot == sl
    if m.any():
        print(f"slot {sl}: waves {m.sum()} end mean {end[m].mean():9.1f}")
