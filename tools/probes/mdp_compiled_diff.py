"""Diagnostic: where the compiled and the plan-interpreting fused MDP kernels differ for one configuration."""
import sys

import torch

sys.path.insert(0, "tests")
import test_mdp_compiled_gpu as T  # noqa: E402

name, n = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
cfg = T._config(name)
tensors, state, g = T._inputs(cfg, n, seed=sum(map(ord, name)) + n)
fast, slow = (T._build(cfg, n, dev, tensors, state, c) for c in (True, False))
print(fast.kernel, slow.kernel)
for m in (fast, slow):
    m.step()
torch.cuda.synchronize()
for key in T.OUTPUTS + T.COMMAND_OUTPUTS:
    if not hasattr(fast, key):
        continue
    a, b = getattr(fast, key).cpu(), getattr(slow, key).cpu()
    bad = (a != b) & ~(torch.isnan(a.float()) & torch.isnan(b.float()))
    if bad.any():
        idx = bad.nonzero()[:8]
        print(key, int(bad.sum()), "differ; first:", idx.tolist())
        for i in idx[:4]:
            i = tuple(i.tolist())
            print("   ", i, float(a[i]), float(b[i]), (a[i].view(torch.int32) - b[i].view(torch.int32)).item() if a.dtype == torch.float32 else "")
