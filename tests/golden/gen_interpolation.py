"""Golden vectors for LinearInterpolation, produced by the REFERENCE class
(``isaaclab/utils/interpolation/linear_interpolation.py``).  Writes tests/golden/interpolation.npz."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

R.bootstrap()
li = R.load("isaaclab.utils.interpolation.linear_interpolation", f"{R.L}/utils/interpolation/linear_interpolation.py")
g = torch.Generator().manual_seed(12)
x = torch.sort(torch.randn(9, generator=g))[0]
x[4] = x[3]                                   # a repeated sample: a step in the function
y = torch.randn(9, generator=g)
q = torch.cat([torch.randn(40, 5, generator=g) * 1.5, x[None, :5].clone(), torch.tensor([[-10.0, 10.0, float(x[0]), float(x[-1]), 0.0]])])
out = dict(x=x, y=y, q=q, f=li.LinearInterpolation(x, y, "cpu").compute(q))
one = li.LinearInterpolation(torch.tensor([0.3]), torch.tensor([2.0]), "cpu")
out["one_sample"] = one.compute(q)
np.savez_compressed(os.path.join(HERE, "interpolation.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote interpolation.npz")
