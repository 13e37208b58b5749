#!/usr/bin/env python3
"""DDPG learner rates on one GPU (RL/MR_ddpg.py:288-305: sample 64 -> critic target -> critic step -> actor step -> two soft
updates): eager PyTorch update, the same captured as one hipGraph, and the fused libmrsim kernel when the library has it.
    python tools/learner_probe.py [--updates 2000] [--batch 64]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mr_rl_amd import MRConfig, MRVecEnv
from mr_rl_amd.ddpg import DDPG

ap = argparse.ArgumentParser()
ap.add_argument("--updates", type=int, default=2000)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--fused", action="store_true")
a = ap.parse_args()
env = MRVecEnv(4096, cfg=MRConfig(noise_var=1.0, auto_reset=True), seed=0, track_actions=True)


def filled(**kw):
    ag = DDPG(env, seed=0, obs_scale=[0.01] * 5, min_batch=a.batch, **kw)
    g = torch.Generator(device="cuda").manual_seed(1)
    n = 10000
    s = torch.randn(n, 5, device="cuda", generator=g)
    ag.buffer.add(s, torch.randn(n, 2, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g),
                  (torch.rand(n, device="cuda", generator=g) < 0.02).float(), s + 0.01 * torch.randn(n, 5, device="cuda", generator=g))
    return ag


def rate(f, n):
    f(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f(n)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


ag = filled()
r_eager = rate(lambda n: [ag.update() for _ in range(n)], min(a.updates, 300))
print(f"eager PyTorch update (no host sync)      : {r_eager:10.1f} updates/s  ({1e6 / r_eager:8.1f} us per update, batch {a.batch})", flush=True)
ag2 = filled()
ag2.capture_update()
r_graph = rate(lambda n: ag2.update_graphed(n), a.updates)
print(f"the same update as ONE hipGraph replay   : {r_graph:10.1f} updates/s  ({1e6 / r_graph:8.1f} us per update)", flush=True)
lc, la = ag2.last_losses
print(f"   losses after {ag2._updates} graphed updates: critic {float(lc):.4g} actor {float(la):.4g}")
if a.fused:
    import ctypes as C
    from mr_rl_amd import _lib
    ag3 = filled(fused=True)
    F = ag3.fused
    idx = torch.randint(0, 10000, (a.batch,), device="cuda", dtype=torch.int32)
    b = ag3.buffer
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    strm = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def raw(n):
        for _ in range(n):
            _lib.check(F._L.mrsim_ddpg_update(C.byref(F.struct), a.batch, 1, p(b.s), p(b.a), p(b.r), p(b.t), p(b.s2), p(idx), 0, 0, 0, None, p(F.losses), strm), "upd")
    r_raw = rate(raw, a.updates)
    print(f"fused kernel alone (fixed indices)       : {r_raw:10.1f} updates/s  ({1e6 / r_raw:8.1f} us per update)", flush=True)
    r_f = rate(lambda n: [ag3.update() for _ in range(n)], a.updates)
    print(f"fused libmrsim kernel (one launch)       : {r_f:10.1f} updates/s  ({1e6 / r_f:8.1f} us per update)", flush=True)
