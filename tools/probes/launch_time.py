"""Diagnostic: duration of named launches of one configuration (events on the launch stream, back to back behind a
blocker), and optionally of the whole step replayed from a graph:
    python tools/probes/launch_time.py anymal "actuator_net_mlp+scene_update,raycaster_cast" [--step [--stats]]
(--stats: with the episode statistics, two graphs in turn, as bench.py replays the step; --two-branches: the recording
with the sensor chain as a parallel branch instead of the one-chain headline; --blocks: per-100-step times of 3000 steps)"""
import sys
import time

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
for rep in range(0 if "--no-kernels" in sys.argv else 3):
    k = bench.time_kernels(cfg, dev, only=names)
    print(config, {n: round(v, 2) for n, v in k.items()})
if "--step" in sys.argv:
    from isaaclab_amd.distributed import OverlappedLogReducer
    reducer = OverlappedLogReducer(cfg.log_sums.numel(), dev) if "--stats" in sys.argv and cfg.log_sums is not None else None
    cfg.overlap_in_graph = "--two-branches" in sys.argv   # (the sensor chain as a branch beside the actuator chain)
    graphs = bench.capture_step(cfg, reducer)
    run_on = torch.cuda.Stream() if "--own-stream" in sys.argv else torch.cuda.current_stream()
    torch.cuda.set_stream(run_on)
    for rep in range(3):
        cfg.restart_sensor_clocks()
        for k in range(40):
            graphs[k % len(graphs)].replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        h0 = time.perf_counter()
        for k in range(200):
            graphs[k % len(graphs)].replay()
        h1 = time.perf_counter()
        b.record()
        torch.cuda.synchronize()
        h2 = time.perf_counter()
        print(f"step replayed ({len(graphs)} graph(s) in turn):", round(a.elapsed_time(b) * 1e3 / 200, 2), "us;  the host's share of a replay:",
              round((h1 - h0) * 1e6 / 200, 2), "us;  by the host's clock, with the wait for the last one:", round((h2 - h0) * 1e6 / 200, 2), "us")
    if "--blocks" in sys.argv:
        cfg.restart_sensor_clocks()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
        ev[0].record()
        for blk in range(30):
            for k in range(100):
                graphs[k % len(graphs)].replay()
            ev[blk + 1].record()
        torch.cuda.synchronize()
        print("per step, blocks of 100 steps:", " ".join(f"{ev[b].elapsed_time(ev[b + 1]) * 10:.1f}" for b in range(30)))
