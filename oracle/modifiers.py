"""CPU restatement of the observation modifiers (test infrastructure only; never imported by the product).

Follows ``isaaclab/utils/modifiers/modifier.py``: scale :22-32, clip :35-46, bias :49-59, DigitalFilter :67-191,
Integrator :194-259.  Pinned by tests/golden/modifiers.npz, recorded from the reference itself.
"""

from __future__ import annotations

import torch


def scale(x, multiplier):
    return x * multiplier


def clip(x, bounds):
    return x.clip(min=bounds[0], max=bounds[1])


def bias(x, value):
    return x + value


class DigitalFilter:
    """Histories with slot 0 newest; y = sum_j B[j] x[i-j] - sum_j A[j] y[i-1-j]."""

    def __init__(self, A, B, shape):
        self.A, self.B = torch.tensor(A, dtype=torch.float32), torch.tensor(B, dtype=torch.float32)
        self.x_n, self.y_n = torch.zeros(*shape, len(B)), torch.zeros(*shape, len(A))

    def reset(self, env_ids=None):
        ids = slice(None) if env_ids is None else env_ids
        self.x_n[ids] = 0.0
        self.y_n[ids] = 0.0

    def __call__(self, x):
        self.x_n = torch.cat([x.unsqueeze(-1), self.x_n[..., :-1]], dim=-1)
        y = (self.x_n * self.B).sum(-1) - (self.y_n * self.A).sum(-1)
        self.y_n = torch.cat([y.unsqueeze(-1), self.y_n[..., :-1]], dim=-1)
        return y


class Integrator:
    def __init__(self, dt, shape):
        self.dt, self.integral, self.y_prev = dt, torch.zeros(*shape), torch.zeros(*shape)

    def reset(self, env_ids=None):
        ids = slice(None) if env_ids is None else env_ids
        self.integral[ids] = 0.0
        self.y_prev[ids] = 0.0

    def __call__(self, x):
        self.integral = self.integral + (x + self.y_prev) / 2 * self.dt
        self.y_prev = x.clone()
        return self.integral
