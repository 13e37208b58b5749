#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (make -C mr_rl_amd/csrc usage)."""
import re, subprocess, sys
t = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/mrsim_usage.txt").read()
errs = [l for l in t.splitlines() if "error" in l]
blocks = re.split(r"Function Name: ", t)[1:]
for b in blocks:
    name = b.split()[0]
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(.*", "", d).replace("mrsim::", "").replace("void ", "")
    print(f"{d:45s} vgpr={g(' VGPRs'):>4} sgpr={g('TotalSGPRs'):>4} occ={g('Occupancy .waves/SIMD.')} scratch={g('ScratchSize .bytes/lane.')} lds={g('LDS Size .bytes/block.')}")
for e in errs[:10]:
    print(e)
