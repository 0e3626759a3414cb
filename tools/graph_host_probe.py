"""GPU probe (diagnostics): is a replayed step bound by the host's graph launches or by the device?  Prints, for the
headline configuration recorded as two branches and as one chain, the host time to issue K replays (loop return, no
synchronisation) and the time until the device has finished them.  The HIP runtime's graph settings are read from the
environment at start-up: run the script once per setting."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bench_configs as BC

from isaaclab_amd.distributed import OverlappedLogReducer

dev = torch.device("cuda:0")
K = 300
cfg = BC.build("anymal", dev, 16384, seed=3000)
red = OverlappedLogReducer(cfg.log_sums.numel(), dev)
graphs = bench.capture_step(cfg, red)
for label, seq in (("graph 0 only", (0,)), ("graph 1 only", (1,)), ("alternating", (0, 1)), ("graph 0 only", (0,)), ("alternating", (0, 1))):
    cfg.restart_sensor_clocks()
    for k in range(30):
        graphs[seq[k % len(seq)]].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        graphs[seq[k % len(seq)]].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:14s}: host issue {1e6 * (t1 - t0) / K:6.1f} us per replay, finished {1e6 * (t2 - t0) / K:6.1f} us per replay", flush=True)
