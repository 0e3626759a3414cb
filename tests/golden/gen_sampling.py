"""Golden vectors for the samplers of ``isaaclab/utils/math.py`` (:1225-1390), produced by the REFERENCE functions; the
``torch.rand`` / ``torch.randn`` / ``torch.normal`` draws they consume are re-drawn after re-seeding and stored.
Writes tests/golden/sampling.npz."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

M = R.bootstrap()["math"]
out = {}
n = 500


def redraw(seed, fn):
    torch.manual_seed(seed)
    return fn()


torch.manual_seed(1)
out["uniform_scalar"] = M.sample_uniform(-0.5, 2.0, (n, 3), "cpu")
out["uniform_scalar_u"] = redraw(1, lambda: torch.rand(n, 3))
lo, hi = torch.tensor([-1.0, 0.0, 2.0, -3.0, 0.5, 0.0]), torch.tensor([1.0, 0.0, 2.5, 3.0, 0.75, 6.0])
torch.manual_seed(2)
out["uniform_cols"] = M.sample_uniform(lo, hi, (n, 6), "cpu")
out["uniform_cols_u"], out["uniform_cols_lo"], out["uniform_cols_hi"] = redraw(2, lambda: torch.rand(n, 6)), lo, hi
torch.manual_seed(3)
out["uniform_1d"] = M.sample_uniform(0.1, 0.2, n, "cpu")
out["uniform_1d_u"] = redraw(3, lambda: torch.rand(n))
torch.manual_seed(4)
out["log_uniform"] = M.sample_log_uniform(1e-3, 10.0, (n, 2), "cpu")
out["log_uniform_u"] = redraw(4, lambda: torch.rand(n, 2))
torch.manual_seed(5)
out["triangle"] = M.sample_triangle(-2.0, 3.0, (n, 2), "cpu")
out["triangle_u"] = redraw(5, lambda: torch.rand(n, 2))
torch.manual_seed(6)
out["gaussian"] = M.sample_gaussian(0.5, 2.0, (n, 3), "cpu")
out["gaussian_z"] = (redraw(6, lambda: torch.normal(mean=0.5, std=2.0, size=(n, 3))) - 0.5) / 2.0   # the draws torch scaled
mean, std = torch.randn(n, 3), torch.rand(n, 3) + 0.1
torch.manual_seed(7)
out["gaussian_t"] = M.sample_gaussian(mean, std, (n, 3), "cpu")
torch.manual_seed(7)
out["gaussian_t_z"] = (torch.normal(mean=mean, std=std) - mean) / std
out["gaussian_t_mean"], out["gaussian_t_std"] = mean, std
torch.manual_seed(8)
out["random_orientation"] = M.random_orientation(n, "cpu")
out["random_orientation_z"] = redraw(8, lambda: torch.randn(n, 4))
torch.manual_seed(9)
out["random_yaw"] = M.random_yaw_orientation(n, "cpu")
out["random_yaw_u"] = redraw(9, lambda: torch.rand(n))
out["default_orientation"] = M.default_orientation(7, "cpu")
np.savez_compressed(os.path.join(HERE, "sampling.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote sampling.npz with", len(out), "arrays")
