"""GPU probe (diagnostics): where the host spends its time when the headline step is issued launch by launch."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as BC

dev = torch.device("cuda:0")
cfg = BC.build(sys.argv[1] if len(sys.argv) > 1 else "anymal", dev, None, seed=3000)
for _ in range(20):
    cfg.step()
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K):
    cfg.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e6 * (t1 - t0) / K:.1f} us per step, finished {1e6 * (t2 - t0) / K:.1f} us per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    cfg.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
