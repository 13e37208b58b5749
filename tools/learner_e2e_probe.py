#!/usr/bin/env python3
"""bench.py's `learner` leg on its own: python tools/learner_e2e_probe.py [num_envs]   (default 262144; 261120 = 1020 workgroups
leaves one of the 256 compute units free for the learner's workgroup)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = bench.parse([])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
t0 = time.time()
out = bench.measure_learner(a, n, torch.device("cuda", 0), 7, 2)
out["num_envs"] = n
print(json.dumps(out, indent=1))
print("leg seconds", time.time() - t0)
