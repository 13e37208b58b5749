#!/usr/bin/env python3
"""Where the host time of bench.py's short timed region goes (the driver runs --steps 20 --warmup 5: ONE 20-step launch group per
sub-shard stream): host enqueue cost of that group behind (a) a short queue -- the host never blocks for long -- and (b) 400
queued episodes, i.e. a ~45 ms blocking synchronize in front of the region, as in bench.py; with and without a 0.2 ms busy loop
between the synchronize and the clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mr_rl_amd import MRConfig
args = bench.parse([])
cfg = MRConfig(noise_var=1.0, auto_reset=True, seed=7)
dev = torch.device("cuda", 0)
reg = bench.make_region(args, cfg, 262144, 0, 1, dev, 7, 2)
reg.run(51 * 100); torch.cuda.synchronize()
for trial in range(12):
    long_queue, spin = trial % 2 == 1, trial % 4 >= 2
    if long_queue:
        reg.run(51 * 400)
        if spin:   # variant: drain the long queue first, give the runtime 2 ms for its housekeeping, re-warm with a short burst
            torch.cuda.synchronize(); time.sleep(2e-3); reg.run(51 * 60)
    reg.run(5)
    reg.col.prime(reg.schedule(20))
    tb = time.perf_counter(); reg.barrier(); tb = time.perf_counter() - tb
    t0 = time.perf_counter(); reg.run(20); t1 = time.perf_counter()
    reg.g.finish(); torch.cuda.synchronize(dev); t2 = time.perf_counter()
    print(f"trial {trial:2d} queue {'400 episodes' if long_queue else 'short       '} (blocked {tb * 1e3:6.2f} ms) busy-loop {int(spin)}: enqueue "
          f"{(t1 - t0) * 1e6:6.1f} us, wait for the GPU {(t2 - t1) * 1e6:6.1f} us, {262144 * 20 / (t2 - t0) / 1e9:5.1f} G env-steps/s", flush=True)
    reg.run(51 * 3 - 25); torch.cuda.synchronize()
