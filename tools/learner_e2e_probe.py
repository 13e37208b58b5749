import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
class A: pass
a = bench.parse([])
t0=time.time()
out = bench.measure_learner(a, 262144, torch.device("cuda",0), 7, 2)
import json; print(json.dumps(out, indent=1)); print("leg seconds", time.time()-t0)
