"""Diagnostic: duration of the fused MDP launch of one configuration (events on the launch stream, back to back)."""
import sys

import torch

sys.path.insert(0, ".")
import bench_configs as BC  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "anymal"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else BC.ENVS[config]
dev = torch.device("cuda:0")
cfg = BC.build(config, dev, envs, seed=3000)
for _ in range(3):
    cfg.step()
torch.cuda.synchronize()
fn = cfg.launches["mdp_step"][0]
fn = fn[0] if isinstance(fn, tuple) else fn
mdp = cfg.env.mdp
for compiled in (True, False):
    mdp.select_kernel(compiled)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    blocker = torch.randn(4096, 4096, device=dev)
    out = torch.empty_like(blocker)
    best = []
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(blocker, blocker, out=out)
        torch.mm(blocker, blocker, out=out)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) * 1e3 / 20)
    print(config, envs, mdp.kernel, "us per launch:", [round(x, 2) for x in best])
