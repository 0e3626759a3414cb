"""Noise functions and models (interface of ``isaaclab/utils/noise/noise_model.py``) on the HIP kernel ``ilx_noise_apply``.

One launch per call.  The reference draws with ``torch.rand_like`` / ``torch.randn_like``; here the draws are Philox, keyed by
the sampling seed, a call counter, the row and the column, or are handed in through ``_draws`` (parity tests replay torch's).
"""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from ctypes import c_float, c_int32, c_int64, c_uint64, c_void_p

import torch

from ... import _abi
from .. import rng as _rng

CONSTANT, UNIFORM, GAUSSIAN = range(3)
_OPS = {"add": 0, "scale": 1, "abs": 2}


def set_noise_seed(seed: int):
    """Seed of the device generators (the noise functions share it with every other drawing module)."""
    _rng.seed_all(seed)


class NoiseDesc(ctypes.Structure):
    """Mirror of ``ilx_noise`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("data", c_void_p), ("out", c_void_p), ("data_stride", c_int64), ("out_stride", c_int64), ("n", c_int64), ("d", c_int32),
        ("row_ids", c_void_p), ("a", c_float), ("b", c_float), ("range", c_float), ("a_cols", c_void_p), ("b_cols", c_void_p),
        ("draws", c_void_p), ("row_bias", c_void_p), ("seed", c_uint64), ("draw", c_uint64),
    ]


_apply = _abi._sig("ilx_noise_apply", c_int32, c_int32, ctypes.POINTER(NoiseDesc), _abi.P)


def _param(value, d: int, device, keep: list):
    """(scalar, per-column tensor or None) of a parameter given as a float or a tensor broadcast along the last dimension."""
    if isinstance(value, torch.Tensor) and value.numel() > 1:
        if value.numel() != d:
            raise ValueError(f"noise parameter of {value.numel()} elements does not match the data's last dimension {d}")
        t = _abi.f32c(value.to(device)).reshape(d)
        keep.append(t)
        return 0.0, t
    return float(value), None


def _run(kind: int, data: torch.Tensor, operation: str, a, b=0.0, *, out=None, row_ids=None, row_bias=None, draws=None):
    if operation not in _OPS:
        raise ValueError(f"Unknown operation in noise: {operation}")
    if not data.is_cuda:
        raise _abi.IlxError("isaaclab_amd kernels need GPU tensors (got a CPU tensor); there is no CPU fallback")
    src = _abi.f32c(data)
    dst = torch.empty_like(src) if out is None else out
    width = src.shape[-1] if src.dim() > 1 else 1
    keep: list = []
    a_s, a_t = _param(a, width, src.device, keep)
    b_s, b_t = _param(b, width, src.device, keep)
    per_col = a_t is not None or b_t is not None
    if per_col:   # parameters broadcast along the last dimension: a row is one run of it
        if a_t is None:
            a_t = torch.full((width,), a_s, device=src.device)
        if b_t is None:
            b_t = torch.full((width,), b_s, device=src.device)
        if row_bias is not None and src.dim() > 2:
            raise NotImplementedError("per-column noise parameters with a per-env bias need 2-D data")
        n_rows, d = src.numel() // max(width, 1), width
    else:         # a row is everything behind the first dimension (what a per-env bias is broadcast over)
        n_rows = src.shape[0] if src.dim() > 0 else 1
        d = src.numel() // max(n_rows, 1)
    ids = None
    if row_ids is not None:
        ids = torch.as_tensor(row_ids, device=src.device, dtype=torch.long).contiguous()
        n_rows = ids.numel()
    if draws is not None:
        draws = _abi.f32c(draws.to(src.device))
        if draws.numel() != n_rows * d:
            raise ValueError(f"{draws.numel()} injected draws for {n_rows * d} elements")
    desc = NoiseDesc(data=src.data_ptr(), out=dst.data_ptr(), data_stride=d, out_stride=d, n=n_rows, d=d,
                     row_ids=None if ids is None else ids.data_ptr(), a=a_s, b=b_s, range=float(b_s - a_s) if not per_col else 0.0,
                     a_cols=_abi.ptr(a_t) if per_col else None, b_cols=_abi.ptr(b_t) if per_col else None, draws=_abi.ptr(draws),
                     row_bias=_abi.ptr(row_bias), seed=_rng.seed(), draw=_rng.next_draw("noise"))
    _abi.check(_apply(kind, _OPS[operation], ctypes.byref(desc), _abi.stream()), "ilx_noise_apply")
    return dst.view(data.shape) if out is None else out


def constant_noise(data: torch.Tensor, cfg) -> torch.Tensor:
    """noise_model.py:20-42."""
    return _run(CONSTANT, data, cfg.operation, cfg.bias)


def uniform_noise(data: torch.Tensor, cfg, _draws: torch.Tensor | None = None) -> torch.Tensor:
    """noise_model.py:45-70.  ``_draws`` (no counterpart in the reference): the U[0,1) values to use."""
    return _run(UNIFORM, data, cfg.operation, cfg.n_min, cfg.n_max, draws=_draws)


def gaussian_noise(data: torch.Tensor, cfg, _draws: torch.Tensor | None = None) -> torch.Tensor:
    """noise_model.py:73-98.  ``_draws`` (no counterpart in the reference): the N(0,1) values to use."""
    return _run(GAUSSIAN, data, cfg.operation, cfg.mean, cfg.std, draws=_draws)


_KINDS = {constant_noise: CONSTANT, uniform_noise: UNIFORM, gaussian_noise: GAUSSIAN}


def _kind_and_params(cfg):
    kind = _KINDS.get(cfg.func)
    if kind is None:
        return None
    if kind == CONSTANT:
        return kind, cfg.bias, 0.0
    return (kind, cfg.n_min, cfg.n_max) if kind == UNIFORM else (kind, cfg.mean, cfg.std)


class NoiseModel:
    """noise_model.py:106-142."""

    def __init__(self, noise_model_cfg, num_envs: int, device: str):
        self._noise_model_cfg, self._num_envs, self._device = noise_model_cfg, num_envs, device

    def reset(self, env_ids: Sequence[int] | None = None):
        pass

    def apply(self, data: torch.Tensor, _draws: torch.Tensor | None = None) -> torch.Tensor:
        cfg = self._noise_model_cfg.noise_cfg
        return cfg.func(data, cfg) if _draws is None else cfg.func(data, cfg, _draws)


class NoiseModelWithAdditiveBias(NoiseModel):
    """noise_model.py:145-182: noise plus a per-env bias that is re-drawn at reset (the draw is applied to the bias the env
    had, as in the reference: an "add" bias noise accumulates over resets).  Noise and bias go out in one launch."""

    def __init__(self, noise_model_cfg, num_envs: int, device: str):
        super().__init__(noise_model_cfg, num_envs, device)
        self._bias_noise_cfg = noise_model_cfg.bias_noise_cfg
        self._bias = torch.zeros((num_envs, 1), device=device)

    def reset(self, env_ids: Sequence[int] | None = None, _draws: torch.Tensor | None = None):
        cfg = self._bias_noise_cfg
        spec = _kind_and_params(cfg)
        if spec is None:   # a user-supplied function: the reference's data flow
            ids = slice(None) if env_ids is None else env_ids
            self._bias[ids] = cfg.func(self._bias[ids], cfg)
            return
        kind, a, b = spec
        ids = None if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)) else env_ids
        _run(kind, self._bias, cfg.operation, a, b, out=self._bias, row_ids=ids, draws=_draws)

    def apply(self, data: torch.Tensor, _draws: torch.Tensor | None = None) -> torch.Tensor:
        cfg = self._noise_model_cfg.noise_cfg
        spec = _kind_and_params(cfg)
        if spec is None or data.shape[0] != self._num_envs:
            return super().apply(data, _draws) + self._bias
        kind, a, b = spec
        return _run(kind, data, cfg.operation, a, b, row_bias=self._bias, draws=_draws)
