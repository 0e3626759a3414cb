"""GPU probe: time the fused MDP kernel by phase subset on the benchmark workload (diagnostics, not a test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from isaaclab_amd.envs import fused_mdp as FM

dev = torch.device("cuda:0")
env, state, _ = bench.build_env(dev, 16384, 3000)
raw = state["raw_action"].to(dev)
for _ in range(3):
    env.step(raw)
torch.cuda.synchronize()

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

m = env.mdp
for name, ph in (("all", FM.PHASE_ALL), ("termination", FM.PHASE_TERMINATION), ("term+reward", FM.PHASE_TERMINATION | FM.PHASE_REWARD),
                 ("term+reward+reset", 7), ("command", FM.PHASE_COMMAND), ("observation", FM.PHASE_OBSERVATION)):
    print(f"{name:20s} {timeit(lambda: m.step(ph)):8.1f} us")
ls = m.t.pop("log_sums"); m.desc.log_sums = None
print(f"{'all, no log_sums':20s} {timeit(lambda: m.step(FM.PHASE_ALL)):8.1f} us")
m.bind(log_sums=ls)
m.bind(noise_u=torch.rand(16384, 235, device=dev))
print(f"{'all, injected noise':20s} {timeit(lambda: m.step(FM.PHASE_ALL)):8.1f} us")
