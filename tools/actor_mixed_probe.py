import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mr_rl_amd import MRConfig, _lib
from mr_rl_amd.actor import DeviceActor
from mr_rl_amd.collector import RolloutCollector
from mr_rl_amd.ddpg import Actor
N = 262144
LIB = _lib.load(sys.argv[1]) if len(sys.argv) > 1 else None   # optional: another build of the library
torch.manual_seed(0)
m = Actor().eval()
for math in ("f32", "bf16x3", "bf16"):
    for mixed in (False, True):
        cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=0)
        tab = bench.mixed_goal_table(cfg, 0) if mixed else None
        pol = DeviceActor.from_module(m, obs_scale=[0.01] * 5, device="cuda", math=math)
        col = RolloutCollector(N, cfg=cfg, seed=0, streams=2, carry="f64", policy=pol, goal_table=tab)
        if LIB is not None:
            col.env._L = LIB
        col.reset()
        for _ in range(30):
            col.collect(); col.ready(); col.release()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        E = 100
        for _ in range(E):
            col.collect(); col.ready(); col.release()
        col.join(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        col.check_status()
        print(f"math={math:7s} goal_table={mixed!s:5s} {N * 51 * E / el / 1e9:6.2f} G env-steps/s  {el / E * 1e3:7.3f} ms per episode", flush=True)
