"""GPU probe: time the fused MDP kernel by phase subset (diagnostics, not a test).

    python tools/mdp_probe.py [config] [envs] [device-count]
Launches are queued behind a long-running blocker so that the device, not the host, paces them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as BC
from isaaclab_amd.envs import fused_mdp as FM

name = sys.argv[1] if len(sys.argv) > 1 else "anymal"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else BC.ENVS[name]
dev = torch.device("cuda:0")
cfg = BC.build(name, dev, envs, 3000)
for _ in range(3):
    cfg.step()
torch.cuda.synchronize()
blk = torch.randn(4096, 4096, device=dev)
out = torch.empty_like(blk)

def span(fn, reps=100, rounds=7):
    """Median over `rounds` of the mean time of `reps` queued launches."""
    for _ in range(3): fn()
    times = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2): torch.mm(blk, blk, out=out)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(times)[len(times) // 2]

m = cfg.env.mdp
if len(sys.argv) > 3 and sys.argv[3] == "device-count":
    m.count_steps_on_device()
print(f"{name} {envs} envs" + (" (step count kept on the device)" if len(sys.argv) > 3 else ""))
for label, ph in (("all", FM.PHASE_ALL), ("termination", FM.PHASE_TERMINATION), ("term+reward", FM.PHASE_TERMINATION | FM.PHASE_REWARD),
                  ("term+reward+reset", 7), ("command", FM.PHASE_COMMAND), ("observation", FM.PHASE_OBSERVATION),
                  ("all but observation", 15)):
    print(f"{label:20s} {span(lambda: m.step(ph, hold_step_count=True)):8.1f} us")
