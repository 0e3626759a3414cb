"""``LinearInterpolation`` (interface of ``isaaclab/utils/interpolation/linear_interpolation.py``) on the HIP kernel
``ilx_linear_interpolation``."""

from __future__ import annotations

import ctypes

import torch

from ... import _abi

_interp = _abi._sig("ilx_linear_interpolation", _abi.P, _abi.P, ctypes.c_int32, _abi.P, _abi.P, ctypes.c_int64, _abi.P)


class LinearInterpolation:
    """linear_interpolation.py:11-88: y(x) through the samples, the boundary value outside them."""

    def __init__(self, x: torch.Tensor, y: torch.Tensor, device: str):
        self._x = x.view(-1).clone().to(device=device, dtype=torch.float32)
        self._y = y.view(-1).clone().to(device=device, dtype=torch.float32)
        if self._x.numel() == 0:
            raise ValueError("Input tensor x is empty!")
        if self._x.numel() != self._y.numel():
            raise ValueError(f"Input tensors x and y have different sizes: {self._x.numel()} != {self._y.numel()}")
        if torch.any(self._x[1:] < self._x[:-1]):
            raise ValueError("Input tensor x is not sorted in ascending order!")

    def compute(self, q: torch.Tensor) -> torch.Tensor:
        src = _abi.f32c(q.to(self._x.device))
        out = torch.empty_like(src)
        _abi.check(_interp(_abi.ptr(self._x), _abi.ptr(self._y), self._x.numel(), _abi.ptr(src), _abi.ptr(out), src.numel(), _abi.stream()),
                   "ilx_linear_interpolation")
        return out.view(q.shape)
