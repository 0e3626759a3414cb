"""GPU probe: host time of a HIP-graph replay of the rough-terrain step against the device time (diagnostics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, bench_configs as BC
from isaaclab_amd.distributed import OverlappedLogReducer

dev = torch.device("cuda:0")
cfg = BC.build("anymal", dev, 16384, 3000)
red = OverlappedLogReducer(cfg.log_sums.numel(), dev)
for overlap in (False, True):
    cfg.env.concurrent_chains = overlap
    graphs = bench.capture_step(cfg, red)
    for _ in range(20):
        graphs[0].replay()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        graphs[0].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"concurrent_chains={overlap}: host loop {1e6 * (t1 - t0) / K:.1f} us per replay, with the final sync {1e6 * (t2 - t0) / K:.1f} us")
