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
labels = ["entry", "stage issued", "stage landed", "-", "terminations", "rewards", "barrier", "reset+command", "barrier",
          "sums/flags out", "observations out"]
print(f"{name} {envs} envs, {blocks} blocks; ticks of 10 ns; per stamp: mean since kernel start, mean since previous stamp")
labels += ["  obs: lane set up", "  obs: loads issued (env 0)", "  obs: noise drawn", "stage: work list reached", "arguments in LDS"]
order = [0, 14, 1, 2, 4, 5, 6, 7, 8, 9, 11, 12, 13, 10]
prev = t[:, 0]
for k in order:
    lab = labels[k]
    print(f"{k:2d} {lab:18s} {(t[:, k] - t0).mean() / 100:8.2f} us   +{(t[:, k] - prev).mean() / 100:6.2f}   (max +{(t[:, k] - prev).max() / 100:6.2f})")
    prev = t[:, k]
print(f"first entry -> last exit {(t[:, 10].max() - t0) / 100:.2f} us; entry spread {(t[:, 0].max() - t0) / 100:.2f} us")
end = (t[:, 10] - t0) / 100.0
print("block exit (us since first entry): " + "  ".join(f"p{q}={np.percentile(end, q):.2f}" for q in (10, 50, 90, 99, 100)))
slow = end >= np.percentile(end, 95)
print("per-phase time of the slowest 5% of blocks minus the mean block:")
prev_k = 0
for k in order[1:]:
    dtk = (t[:, k] - t[:, prev_k]) / 100.0
    print(f"   {labels[k]:28s} mean {dtk.mean():6.2f}  slow {dtk[slow].mean():6.2f}")
    prev_k = k
print("entry time by blockIdx % 8 (XCD): " + "  ".join(f"{((t[i::8, 0] - t0) / 100.0).mean():.2f}" for i in range(8)))
print("exit  time by blockIdx % 8 (XCD): " + "  ".join(f"{end[i::8].mean():.2f}" for i in range(8)))
hw = (ctypes.c_uint * (8 * 2048))()
if hasattr(_abi.lib, "ilx_debug_mdp_hwid") and _abi.lib.ilx_debug_mdp_hwid(hw, 8 * 2048) == 0:
    h = np.frombuffer(hw, dtype=np.uint32).reshape(2048, 8)[:blocks]
    simd = (h[:, :4] >> 4) & 3
    cu = (h[:, :4] >> 8) & 15
    sh = (h[:, :4] >> 12) & 1
    se = (h[:, :4] >> 13) & 7
    xcc = h[:, 4:] & 15
    print("SIMD of wavefronts 0..3, first 12 blocks:", simd[:12].tolist())
    print("wavefront 0's SIMD, histogram:", np.bincount(simd[:, 0], minlength=4).tolist())
    where = (xcc[:, 0].astype(np.int64) << 12) | (se[:, 0] << 8) | (sh[:, 0] << 4) | cu[:, 0]
    uniq, inv = np.unique(where, return_inverse=True)
    print(f"{len(uniq)} distinct CUs; blocks per CU: {np.bincount(np.bincount(inv)).tolist()} (index = count)")
    for c in range(3):
        members = np.nonzero(inv == c)[0]
        print(f"  CU {uniq[c]:#x}: blocks {members.tolist()}, wave-0 SIMDs {simd[members, 0].tolist()}, exit {end[members].round(1).tolist()}")
    per_cu_end = np.array([end[inv == c].max() for c in range(len(uniq))])
    print("last exit per CU: " + "  ".join(f"p{q}={np.percentile(per_cu_end, q):.2f}" for q in (0, 10, 50, 90, 100)))
