"""GPU probe: steady timing of every launch closure of a bench configuration (median over rounds of queued launches).
    python tools/kernel_probe.py [config] [kernel-name-substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as BC

name = sys.argv[1] if len(sys.argv) > 1 else "anymal"
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
cfg = BC.build(name, dev, BC.ENVS[name], 3000)
for _ in range(3):
    cfg.step()
torch.cuda.synchronize()
blk = torch.randn(4096, 4096, device=dev)
out = torch.empty_like(blk)


def span(fn, reps=100, rounds=9):
    for _ in range(3):
        fn()
    times = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            torch.mm(blk, blk, out=out)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / reps * 1e3)
    times.sort()
    return times[len(times) // 2], times[0]


for kernel, (fn, count) in cfg.launches.items():
    if only not in kernel:
        continue
    if isinstance(fn, tuple):
        (med, low), (omed, olow) = span(fn[0]), span(fn[1])
        print(f"{kernel:24s} {med - omed:7.2f} us median, {low - olow:7.2f} us best (x{count} per step; overhead launch subtracted)")
    else:
        med, low = span(fn)
        print(f"{kernel:24s} {med:7.2f} us median, {low:7.2f} us best (x{count} per step)")
