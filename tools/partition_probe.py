#!/usr/bin/env python3
"""Collection-only rate of the fused rollout with the bf16 actor in the kernel, on ordinary streams and on CU-masked streams
(mr_rl_amd.partition): python tools/partition_probe.py [math]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_rl_amd import MRConfig, _lib
from mr_rl_amd.actor import DeviceActor
from mr_rl_amd.collector import RolloutCollector
from mr_rl_amd.ddpg import Actor
from mr_rl_amd.partition import CuPartition
math = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
torch.manual_seed(7)
module = Actor().eval()
cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7, noise_law="collapsed")
N, EP = 262144, 150


def run(S, stream_list=None, cur=None, policy=True):
    pol = DeviceActor.from_module(module, obs_scale=[0.01] * 5, device=dev, math=math) if policy else None
    ctx = torch.cuda.stream(cur) if cur is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        col = RolloutCollector(N, cfg=cfg, device=dev, seed=7, streams=S, policy=pol, stream_list=stream_list)
        col.reset()
        for _ in range(30):
            col.collect(); col.ready(); col.release()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(EP):
            col.collect(); col.ready(); col.release()
        col.join()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        col.check_status()
    return N * 51 * EP / el / 1e9, el / EP * 1e6


def full_mask_streams(n):
    L = _lib.lib()
    out = []
    for _ in range(n):
        m = (C.c_uint32 * 8)(*([0xffffffff] * 8))
        h = C.c_void_p()
        _lib.check(L.mrsim_stream_create_cu_mask(0, m, 8, C.byref(h)), "create")
        out.append(torch.cuda.ExternalStream(h.value, device=dev))
    return out


for pol in (True, False):
    print("policy in kernel:", pol)
    for S in (2, 4, 8, 16):
        print("  ordinary streams       S=%2d: %6.2f G env-steps/s  %7.1f us/episode" % ((S,) + run(S, policy=pol)))
    for S in (2, 8):
        st = full_mask_streams(S + 1)
        print("  full-mask ext streams  S=%2d: %6.2f G env-steps/s  %7.1f us/episode" % ((S,) + run(S, st[:S], st[S], policy=pol)))
    for S in (2, 4, 8, 16):
        part = CuPartition(dev, per_xcc=1, collection_streams=S)
        print("  partition (248 units)  S=%2d: %6.2f G env-steps/s  %7.1f us/episode" % ((S,) + run(S, part.collection_streams, part.learner_stream, policy=pol)))
        part.close()
