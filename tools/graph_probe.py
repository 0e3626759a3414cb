"""GPU probe: time of a HIP-graph replay of the rough-terrain step, by how the step's two chains are recorded (diagnostics)."""
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
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"chains as branches={overlap}: {1e6 * (t2 - t0) / K:.1f} us per replay")
