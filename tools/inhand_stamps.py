"""GPU probe (diagnostics build: make EXTRA=-DILX_INHAND_STAMPS): where a workgroup of the in-hand step kernel spends its time.
    python tools/inhand_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_configs as BC
from isaaclab_amd import _abi

cfg = BC.build("shadow", torch.device("cuda:0"), BC.ENVS["shadow"], 5000)
for _ in range(5):
    cfg.step()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 1024))()
assert _abi.lib.ilx_debug_inhand_stamps(buf, 8 * 1024) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[:512].astype(np.int64)
t0 = t[:, 0].min()
labels = ["entry", "loads issued, actions staged, barrier", "dones decided", "reward, sums", "decisions stored", "observation stored"]
prev = t[:, 0]
for k, lab in enumerate(labels):
    print(f"{k} {lab:26s} {(t[:, k] - t0).mean() / 100:7.2f} us   +{(t[:, k] - prev).mean() / 100:6.2f}   (max +{(t[:, k] - prev).max() / 100:6.2f})")
    prev = t[:, k]
print("entry spread:", (t[:, 0].max() - t0) / 100, "us; last exit", (t[:, 5].max() - t0) / 100, "us")
