#!/usr/bin/env python3
"""bench.py's 20-step region (driver flags) starts right behind a blocking synchronize that ended a long queue; the first
launches after it have cost 18-83 us of host time from run to run.  Does housekeeping between the barrier and the clock (an event
record + synchronize on every stream, a short sleep, a stream query) make the enqueue cost of the region steady?  Variants
interleaved, bench.py's own sequence (drain, 2 ms, 60 episodes, 5 warm-up steps, barrier) in front of each."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mr_rl_amd import MRConfig
args = bench.parse([])
cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7)
dev = torch.device("cuda", 0)
reg = bench.make_region(args, cfg, 262144, 0, 1, dev, 7, 2)
reg.run(51 * 100); torch.cuda.synchronize()
evs = [torch.cuda.Event() for _ in reg.col.streams] + [torch.cuda.Event()]


def flush():
    for ev, st in zip(evs, list(reg.col.streams) + [torch.cuda.current_stream(dev)]):
        ev.record(st)
    for ev in evs:
        ev.synchronize()


VARIANTS = {"none": lambda: None, "events": flush, "sleep200us": lambda: time.sleep(2e-4),
            "events+sleep": lambda: (flush(), time.sleep(2e-4)), "query": lambda: [s.query() for s in reg.col.streams],
            "sync2": lambda: torch.cuda.synchronize(dev)}
res = {k: [] for k in VARIANTS}
for trial in range(8):
    for name, fn in VARIANTS.items():
        reg.run(51 * 400)
        torch.cuda.synchronize(); time.sleep(2e-3); reg.run(51 * 60)
        reg.run(5)
        reg.col.prime(reg.schedule(20))
        reg.barrier()
        fn()
        t0 = time.perf_counter(); reg.run(20); t1 = time.perf_counter()
        reg.g.finish(); torch.cuda.synchronize(dev); t2 = time.perf_counter()
        res[name].append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, 262144 * 20 / (t2 - t0) / 1e9))
        reg.run(51 * 3 - 25); torch.cuda.synchronize()
for name, r in res.items():
    enq, wait, g = zip(*r)
    print(f"{name:14s} enqueue median {statistics.median(enq):6.1f} us (min {min(enq):5.1f}, max {max(enq):6.1f}); wait median "
          f"{statistics.median(wait):5.1f} us; {statistics.median(g):5.1f} G env-steps/s median, {min(g):5.1f} worst")
