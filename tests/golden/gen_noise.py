"""Golden vectors for the noise functions and models, produced by the REFERENCE
(``isaaclab/utils/noise/noise_model.py``) with the RNG stream replayed so that the drawn values are stored as the draws the
kernel consumes.  Writes tests/golden/noise.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
nm = m["noise_model"]

out = {}
g = torch.Generator().manual_seed(2718)
n, d = 80, 6
x = torch.randn(n, d, generator=g) * 2.0
out["x"] = x
cols_a = torch.tensor([-0.3, -0.2, -0.1, 0.0, 0.1, 0.2])
cols_b = torch.tensor([0.4, 0.5, 0.6, 0.7, 0.8, 0.9])
out["cols_a"], out["cols_b"] = cols_a, cols_b
seed = 100
for op in ("add", "scale", "abs"):
    out[f"constant_{op}"] = nm.constant_noise(x, NS(bias=0.37, operation=op))
    out[f"constant_cols_{op}"] = nm.constant_noise(x, NS(bias=cols_b.clone(), operation=op))
    for tag, lo, hi in (("", -0.15, 0.45), ("_cols", cols_a, cols_b)):
        seed += 1
        torch.manual_seed(seed)
        out[f"uniform{tag}_{op}_draws"] = torch.rand_like(x)
        torch.manual_seed(seed)
        out[f"uniform{tag}_{op}"] = nm.uniform_noise(x, NS(n_min=lo, n_max=hi, operation=op))
        seed += 1
        torch.manual_seed(seed)
        out[f"gaussian{tag}_{op}_draws"] = torch.randn_like(x)
        torch.manual_seed(seed)
        out[f"gaussian{tag}_{op}"] = nm.gaussian_noise(x, NS(mean=lo, std=hi, operation=op))

# NoiseModelWithAdditiveBias: gaussian noise, gaussian bias drawn at reset (applied to the bias the env had)
noise_cfg = NS(func=nm.gaussian_noise, mean=0.0, std=0.05, operation="add")
bias_cfg = NS(func=nm.gaussian_noise, mean=0.0, std=0.3, operation="add")
model = nm.NoiseModelWithAdditiveBias(NS(noise_cfg=noise_cfg, bias_noise_cfg=bias_cfg), num_envs=n, device="cpu")
resets = [None, torch.tensor([2, 5, 40, 79]), torch.tensor([5, 6])]
for k, ids in enumerate(resets):
    count = n if ids is None else len(ids)
    torch.manual_seed(500 + k)
    out[f"model_reset{k}_draws"] = torch.randn(count, 1)
    torch.manual_seed(500 + k)
    model.reset(ids)
    out[f"model_reset{k}_ids"] = torch.arange(n) if ids is None else ids
    out[f"model_bias{k}"] = model._bias.clone()
    torch.manual_seed(600 + k)
    out[f"model_apply{k}_draws"] = torch.randn_like(x)
    torch.manual_seed(600 + k)
    out[f"model_apply{k}"] = model.apply(x)
np.savez_compressed(os.path.join(HERE, "noise.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote noise.npz with", len(out), "arrays")
