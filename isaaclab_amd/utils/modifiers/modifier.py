"""Observation modifiers (interface of ``isaaclab/utils/modifiers/modifier.py``) on the HIP kernel ``ilx_modifier_apply``.

Every call is one launch over the (num_envs, d) block.  The input may be a column range of a wider row (a strided view with
unit inner stride): no staging copy is made for it.
"""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from ctypes import c_float, c_int32, c_int64, c_void_p

import torch

from ... import _abi
from .modifier_base import ModifierBase

COPY, SCALE, CLIP, BIAS, DIGITAL_FILTER, INTEGRATOR = range(6)
MAX_TAPS = 16


class ModifierDesc(ctypes.Structure):
    """Mirror of ``ilx_modifier`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("inp", c_void_p), ("out", c_void_p), ("in_stride", c_int64), ("out_stride", c_int64), ("n", c_int64), ("d", c_int32),
        ("p0", c_float), ("p1", c_float), ("has_lo", c_int32), ("has_hi", c_int32), ("na", c_int32), ("nb", c_int32),
        ("A", c_float * MAX_TAPS), ("B", c_float * MAX_TAPS), ("x_n", c_void_p), ("y_n", c_void_p),
    ]


_apply = _abi._sig("ilx_modifier_apply", c_int32, ctypes.POINTER(ModifierDesc), _abi.P)
_reset = _abi._sig("ilx_modifier_reset", _abi.P, c_int64, _abi.P, c_int64, _abi.P, c_int64, _abi.P)


def _rows(t: torch.Tensor) -> tuple[torch.Tensor, int, int, int]:
    """(tensor, rows, width, row stride) of ``t`` seen as a block of rows with unit inner stride."""
    if not t.is_cuda:
        raise _abi.IlxError("isaaclab_amd kernels need GPU tensors (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t, t.shape[0], t.shape[1], t.stride(0)
    t = t.contiguous()
    n = t.shape[0] if t.dim() > 0 else 1
    d = t.numel() // max(n, 1)
    return t, n, d, d


def apply(kind: int, data: torch.Tensor, out: torch.Tensor | None = None, **fields) -> torch.Tensor:
    """One launch of ``ilx_modifier_apply``; ``out`` defaults to a new dense tensor of ``data``'s shape."""
    src, n, d, stride = _rows(data)
    if out is None:
        out = torch.empty(data.shape, dtype=torch.float32, device=data.device)
    dst, n2, d2, ostride = _rows(out)
    if dst is not out and dst.data_ptr() != out.data_ptr():
        raise _abi.IlxError("modifier output must be float32 rows with unit inner stride")
    if (n, d) != (n2, d2):
        raise ValueError(f"modifier input {tuple(data.shape)} and output {tuple(out.shape)} differ in shape")
    desc = ModifierDesc(inp=src.data_ptr(), out=dst.data_ptr(), in_stride=stride, out_stride=ostride, n=n, d=d, **fields)
    _abi.check(_apply(kind, ctypes.byref(desc), _abi.stream()), "ilx_modifier_apply")
    return out


def scale(data: torch.Tensor, multiplier: float) -> torch.Tensor:
    """modifier.py:22-32."""
    return apply(SCALE, data, p0=float(multiplier))


def clip(data: torch.Tensor, bounds: tuple[float | None, float | None]) -> torch.Tensor:
    """modifier.py:35-46: a bound that is None is not applied."""
    lo, hi = bounds
    if lo is None and hi is None:
        raise RuntimeError("torch.clamp: At least one of 'min' or 'max' must not be None")
    return apply(CLIP, data, p0=float(lo or 0.0), p1=float(hi or 0.0), has_lo=int(lo is not None), has_hi=int(hi is not None))


def bias(data: torch.Tensor, value: float) -> torch.Tensor:
    """modifier.py:49-59."""
    return apply(BIAS, data, p0=float(value))


def _zero_rows(a: torch.Tensor, b: torch.Tensor, env_ids):
    n = a.shape[0]
    la, lb = a.numel() // max(n, 1), b.numel() // max(n, 1)
    if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)):
        ids, count = None, n
    else:
        ids = torch.as_tensor(env_ids, device=a.device, dtype=torch.long).contiguous()
        count = ids.numel()
    _abi.check(_reset(_abi.ptr(a), la, _abi.ptr(b), lb, None if ids is None else ids.data_ptr(), count, _abi.stream()),
               "ilx_modifier_reset")


class DigitalFilter(ModifierBase):
    """modifier.py:67-191: ``y_i = sum_j B[j] x_{i-j} - sum_j A[j] y_{i-1-j}`` with the input and output histories kept per
    element (slot 0 newest), shifted and evaluated in one launch."""

    def __init__(self, cfg, data_dim: tuple[int, ...], device: str) -> None:
        if cfg.A is None or cfg.B is None:
            raise ValueError("Digital filter coefficients A and B must not be None. Please provide valid coefficients.")
        super().__init__(cfg, data_dim, device)
        if not (1 <= len(cfg.A) <= MAX_TAPS and 1 <= len(cfg.B) <= MAX_TAPS):
            raise ValueError(f"Digital filter supports 1..{MAX_TAPS} coefficients per side; got A={len(cfg.A)}, B={len(cfg.B)}")
        self.A = torch.tensor(cfg.A, dtype=torch.float32, device=device).unsqueeze(1)
        self.B = torch.tensor(cfg.B, dtype=torch.float32, device=device).unsqueeze(1)
        self.x_n = torch.zeros(self._data_dim + (len(cfg.B),), device=device)
        self.y_n = torch.zeros(self._data_dim + (len(cfg.A),), device=device)

    def reset(self, env_ids: Sequence[int] | None = None):
        _zero_rows(self.x_n, self.y_n, env_ids)

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        a, b = self._cfg.A, self._cfg.B
        return apply(DIGITAL_FILTER, data, na=len(a), nb=len(b), A=(c_float * MAX_TAPS)(*a), B=(c_float * MAX_TAPS)(*b),
                     x_n=_abi.ptr(self.x_n), y_n=_abi.ptr(self.y_n))


class Integrator(ModifierBase):
    """modifier.py:194-259: running integral by the trapezoid of the current and the previous sample.  As in the
    reference, the returned tensor IS the integral buffer: in-place work on it by the caller changes the state."""

    def __init__(self, cfg, data_dim: tuple[int, ...], device: str):
        super().__init__(cfg, data_dim, device)
        self.integral = torch.zeros(self._data_dim, device=device)
        self.y_prev = torch.zeros(self._data_dim, device=device)

    def reset(self, env_ids: Sequence[int] | None = None):
        _zero_rows(self.integral, self.y_prev, env_ids)

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        return apply(INTEGRATOR, data, out=self.integral, p0=float(self._cfg.dt), x_n=_abi.ptr(self.integral), y_n=_abi.ptr(self.y_prev))
