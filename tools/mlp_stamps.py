"""GPU probe (diagnostics build: CXXFLAGS+=-DILX_MLP_STAMPS): where a block of the actuator-network kernel spends its time.
    python tools/mlp_stamps.py [envs] [launch name, e.g. actuator_net_mlp+scene_update]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_configs as BC
from isaaclab_amd import _abi

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = BC.build("anymal", torch.device("cuda:0"), envs, 3000)
for _ in range(3):
    cfg.step()
torch.cuda.synchronize()
fn = cfg.launches[sys.argv[2] if len(sys.argv) > 2 else "actuator_net_mlp"]
fn = fn[0] if isinstance(fn, tuple) else fn
for _ in range(5):
    fn()
torch.cuda.synchronize()
blocks = min(2048, (envs * 12 + 127) // 128, 256 * int(os.environ.get("ILX_STAY", 3)))
buf = (ctypes.c_ulonglong * (8 * 2048))()
assert _abi.lib.ilx_debug_mlp_stamps(buf, 8 * 2048) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8)[:blocks].astype(np.int64)
t0 = t[:, 0].min()
labels = ["entry", "row loads issued", "image in LDS", "tile 0: activation 0 done", "tile 0 stored", "tile 1 stored", "last tile stored"]
prev = t[:, 0]
for k, lab in enumerate(labels):
    print(f"{k} {lab:20s} {(t[:, k] - t0).mean() / 100:8.2f} us   +{(t[:, k] - prev).mean() / 100:6.2f}   (max +{(t[:, k] - prev).max() / 100:6.2f})")
    prev = t[:, k]
end = (t[:, 6] - t0) / 100.0
print("entry: " + "  ".join(f"p{q}={np.percentile((t[:, 0] - t0) / 100.0, q):.2f}" for q in (0, 50, 90, 100)))
print("exit:  " + "  ".join(f"p{q}={np.percentile(end, q):.2f}" for q in (0, 10, 50, 90, 100)))
