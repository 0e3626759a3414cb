"""GPU probe (diagnostics build: make CXXFLAGS+=-DILX_MDP_STAMPS): where a block of the fused MDP kernel spends its time.
    python tools/mdp_stamps.py [config] [envs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_configs as BC
from isaaclab_amd import _abi
from isaaclab_amd.envs import fused_mdp as FM

name = sys.argv[1] if len(sys.argv) > 1 else "anymal"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else BC.ENVS[name]
cfg = BC.build(name, torch.device("cuda:0"), envs, 3000)
for _ in range(3):
    cfg.step()
m = cfg.env.mdp
for _ in range(5):
    m.step(FM.PHASE_ALL)
torch.cuda.synchronize()
blocks = min(2048, (envs + 15) // 16)
buf = (ctypes.c_ulonglong * (16 * 2048))()
assert _abi.lib.ilx_debug_mdp_stamps(buf, 16 * 2048) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16)[:blocks, :16].astype(np.int64)
t0 = t[:, 0].min()
labels = ["entry", "stage issued", "stage landed", "force peaks", "terminations", "rewards", "barrier", "reset+command", "barrier",
          "sums/flags out", "observations out"]
print(f"{name} {envs} envs, {blocks} blocks; ticks of 10 ns; per stamp: mean since kernel start, mean since previous stamp")
labels += ["  obs: lane set up", "  obs: loads issued (env 0)", "  obs: noise drawn", "  obs: columns stored", "-"]
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 14, 10]
prev = t[:, 0]
for k in order:
    lab = labels[k]
    print(f"{k:2d} {lab:18s} {(t[:, k] - t0).mean() / 100:8.2f} us   +{(t[:, k] - prev).mean() / 100:6.2f}   (max +{(t[:, k] - prev).max() / 100:6.2f})")
    prev = t[:, k]
print(f"first entry -> last exit {(t[:, 10].max() - t0) / 100:.2f} us; entry spread {(t[:, 0].max() - t0) / 100:.2f} us")
