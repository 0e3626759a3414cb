"""GPU probe: which chain of the rough-terrain step bounds a HIP-graph replay (diagnostics): replays with one chain left out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, bench_configs as BC
from isaaclab_amd.distributed import OverlappedLogReducer

dev = torch.device("cuda:0")
cfg = BC.build("anymal", dev, 16384, 3000)
red = OverlappedLogReducer(cfg.log_sums.numel(), dev)
env = cfg.env
full_a, full_b, full_m = env.step_actuators, env.step_sensors, env.step_managers
for label, a, b, m in (("full step", full_a, full_b, full_m), ("without the actuator chain", lambda raw: None, full_b, full_m),
                       ("without the sensor chain", full_a, lambda: None, full_m), ("without the managers", full_a, full_b, lambda *x: None),
                       ("actuator chain only", full_a, lambda: None, lambda *x: None), ("sensor chain only", lambda raw: None, full_b, lambda *x: None)):
    env.step_actuators, env.step_sensors, env.step_managers = a, b, m
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
    print(f"{label:32s} {1e6 * (t2 - t0) / K:7.1f} us per replay")
