"""Golden vectors for the observation modifiers, produced by the REFERENCE functions and classes
(``isaaclab/utils/modifiers/modifier.py``).  Writes tests/golden/modifiers.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

R.bootstrap()
R.load("isaaclab.utils.modifiers.modifier_base", f"{R.L}/utils/modifiers/modifier_base.py")
mod = R.load("isaaclab.utils.modifiers.modifier", f"{R.L}/utils/modifiers/modifier.py")

out = {}
g = torch.Generator().manual_seed(5150)
n, d = 96, 7
x = torch.randn(n, d, generator=g) * 3.0
out["x"] = x
out["scale"] = mod.scale(x, 1.7)
out["bias"] = mod.bias(x, -0.35)
out["clip_both"] = mod.clip(x, (-1.5, 2.0))
out["clip_lo"] = mod.clip(x, (-0.5, None))
out["clip_hi"] = mod.clip(x, (None, 0.25))

A, B = [0.6, -0.2], [0.5, 0.3, 0.2]
out["filter_A"], out["filter_B"] = torch.tensor(A), torch.tensor(B)
f = mod.DigitalFilter(NS(A=A, B=B), data_dim=(n, d), device="cpu")
steps = torch.randn(8, n, d, generator=g)
reset_ids = torch.tensor([1, 5, 40, 95])
ys = []
for k in range(8):
    if k == 5:
        f.reset(reset_ids)
    if k == 7:
        f.reset()
    ys.append(f(steps[k]).clone())
out["filter_in"], out["filter_out"], out["filter_reset_ids"] = steps, torch.stack(ys), reset_ids
out["filter_x_n"], out["filter_y_n"] = f.x_n.clone(), f.y_n.clone()

# unit delay and moving average, the docstring's examples
for tag, a, b in (("delay", [0.0], [0.0, 1.0]), ("average", [0.0], [0.25, 0.25, 0.25, 0.25]), ("lowpass", [-0.8], [0.2])):
    f = mod.DigitalFilter(NS(A=a, B=b), data_dim=(n, d), device="cpu")
    out[f"{tag}_out"] = torch.stack([f(steps[k]).clone() for k in range(6)])

it = mod.Integrator(NS(dt=0.02), data_dim=(n, d), device="cpu")
ys = []
for k in range(8):
    if k == 4:
        it.reset(reset_ids)
    ys.append(it(steps[k]).clone())
out["integrator_out"] = torch.stack(ys)
out["integrator_y_prev"] = it.y_prev.clone()
np.savez_compressed(os.path.join(HERE, "modifiers.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote modifiers.npz with", len(out), "arrays")
