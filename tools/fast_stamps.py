"""GPU probe (diagnostics build: make CXXFLAGS+=-DILX_FAST_STAMPS): where the wavefronts of a DECIDE block of the compiled
fused MDP kernel spend their time.   python tools/fast_stamps.py [config] [envs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_configs as BC
from isaaclab_amd import _abi

name = sys.argv[1] if len(sys.argv) > 1 else "anymal"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else BC.ENVS[name]
cfg = BC.build(name, torch.device("cuda:0"), envs, 3000)
for _ in range(3):
    cfg.step()
m = cfg.env.mdp
for _ in range(5):
    m.step()
torch.cuda.synchronize()
blocks = min(512, (envs + 63) // 64)
buf = (ctypes.c_ulonglong * (512 * 5 * 8))()
assert _abi.lib.ilx_debug_fast_stamps(buf, 512 * 5 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 5, 8)[:blocks].astype(np.int64)
t0 = t[:, :, 0].min()
labels = ["entry", "terminations done", "rewards done", "obs values / command done", "before barrier", "after barrier", "in-place stores issued", "kernel end"]
print(f"{name} {envs} envs ({m.kernel}), {blocks} DECIDE blocks; us since the first wavefront's entry, mean over blocks, per wavefront 0..4 (4: command)")
for k, lab in enumerate(labels):
    print(f"{k} {lab:28s} " + "  ".join(f"{(t[:, w, k] - t0).mean() / 100:6.2f}" for w in range(5)) +
          f"    max {(t[:, :, k] - t0).max() / 100:6.2f}")
print(f"entry spread over blocks: {(t[:, 0, 0].max() - t0) / 100:.2f} us; last exit {(t[:, :, 7].max() - t0) / 100:.2f} us")

sb = (ctypes.c_ulonglong * (2048 * 8))()
if hasattr(_abi.lib, "ilx_debug_scan_stamps") and _abi.lib.ilx_debug_scan_stamps(sb, 2048 * 8) == 0:
    st = np.frombuffer(sb, dtype=np.uint64).reshape(2048, 8).astype(np.int64)
    st = st[st[:, 0] > 0]
    if len(st):
        ent, ext = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0
        print(f"{len(st)} SCAN blocks: entry " + "  ".join(f"p{q}={np.percentile(ent, q):.2f}" for q in (0, 10, 50, 90, 100)) +
              "; exit " + "  ".join(f"p{q}={np.percentile(ext, q):.2f}" for q in (0, 10, 50, 90, 100)) + f"; mean life {(ext - ent).mean():.2f} us")
        trips = [(st[:, 2 + k] - st[:, 0]).mean() / 100.0 for k in range(6) if (st[:, 2 + k] > 0).all()]
        print("stores of trip k issued, us after the block's entry (first wavefront): " + "  ".join(f"{x:.2f}" for x in trips))
