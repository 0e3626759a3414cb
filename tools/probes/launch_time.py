"""Diagnostic: duration of named launches of one configuration (events on the launch stream, back to back behind a
blocker), and optionally of the whole step replayed from a graph:
    python tools/probes/launch_time.py anymal "actuator_net_mlp+scene_update,raycaster_cast" [--step]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import bench_configs as BC  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "anymal"
names = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else None
dev = torch.device("cuda:0")
cfg = BC.build(config, dev, BC.ENVS[config], seed=3000)
for _ in range(3):
    cfg.step()
torch.cuda.synchronize()
for rep in range(3):
    k = bench.time_kernels(cfg, dev, only=names)
    print(config, {n: round(v, 2) for n, v in k.items()})
if "--step" in sys.argv:
    graphs = bench.capture_step(cfg, None)
    for rep in range(3):
        cfg.restart_sensor_clocks()
        for _ in range(20):
            graphs[0].replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            graphs[0].replay()
        b.record()
        torch.cuda.synchronize()
        print("step replayed:", round(a.elapsed_time(b) * 1e3 / 200, 2), "us")
