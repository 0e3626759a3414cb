"""GPU probe: a HIP-graph replay of the rough-terrain step, host submission time against device time (diagnostics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, bench_configs as BC
from isaaclab_amd.distributed import OverlappedLogReducer

dev = torch.device("cuda:0")
cfg = BC.build("anymal", dev, 16384, 3000)
red = OverlappedLogReducer(cfg.log_sums.numel(), dev)
for overlap in (False, True):
    cfg.overlap_in_graph = overlap
    graphs = bench.capture_step(cfg, red)
    for _ in range(20):
        graphs[0].replay()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K):
        graphs[0].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # one replay at a time: the device time of a replay whose submission has nothing to hide behind
    t3 = time.perf_counter()
    for _ in range(50):
        graphs[0].replay()
        torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"chains as branches={overlap}: host submits a replay in {1e6 * (t1 - t0) / K:.1f} us; back to back {1e6 * (t2 - t0) / K:.1f} us per replay; "
          f"one at a time {1e6 * (t4 - t3) / 50:.1f} us")
