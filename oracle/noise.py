"""CPU restatement of the noise functions (test infrastructure only; never imported by the product).

Follows ``isaaclab/utils/noise/noise_model.py``: constant_noise :20-42, uniform_noise :45-70, gaussian_noise :73-98,
NoiseModelWithAdditiveBias :145-182; the draws are arguments.  Pinned by tests/golden/noise.npz, recorded from the reference.
"""

from __future__ import annotations

import torch


def constant(x, bias, op):
    return {"add": x + bias, "scale": x * bias, "abs": torch.zeros_like(x) + bias}[op]


def uniform(x, n_min, n_max, op, u):
    r = u * (n_max - n_min) + n_min
    return {"add": x + u * (n_max - n_min) + n_min, "scale": x * r, "abs": r}[op]


def gaussian(x, mean, std, op, z):
    return {"add": x + mean + std * z, "scale": x * (mean + std * z), "abs": mean + std * z}[op]


class AdditiveBiasModel:
    def __init__(self, n):
        self.bias = torch.zeros(n, 1)

    def reset(self, ids, mean, std, z):
        self.bias[ids] = gaussian(self.bias[ids], mean, std, "add", z)

    def apply(self, x, mean, std, z):
        return gaussian(x, mean, std, "add", z) + self.bias
