"""``CircularBuffer`` (interface of ``isaaclab/utils/buffers/circular_buffer.py``) on the HIP kernels
``ilx_circular_buffer_append`` / ``ilx_circular_buffer_reset``."""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from ctypes import c_int32, c_int64, c_void_p

import torch

from ... import _abi


class RingDesc(ctypes.Structure):
    """Mirror of ``ilx_circular_buffer`` (include/isaaclab_amd.h)."""

    _fields_ = [("buffer", c_void_p), ("num_pushes", c_void_p), ("max_len", c_int32), ("pointer", c_int32), ("n", c_int64), ("d", c_int32)]


_append = _abi._sig("ilx_circular_buffer_append", ctypes.POINTER(RingDesc), _abi.P, c_int64, _abi.P, c_int64, _abi.P, _abi.P, c_int64, _abi.P)
_reset = _abi._sig("ilx_circular_buffer_reset", ctypes.POINTER(RingDesc), _abi.P, c_int64, _abi.P)


def _rows(t: torch.Tensor, n: int) -> tuple[torch.Tensor, int]:
    """(float32 tensor, row stride) of ``t`` seen as n rows with unit inner stride (column ranges of wider rows stay in place)."""
    if not t.is_cuda:
        raise _abi.IlxError("isaaclab_amd kernels need GPU tensors (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t, t.stride(0)
    t = t.contiguous()
    return t, t.numel() // max(n, 1)


class CircularBuffer:
    """circular_buffer.py:10-166: a ring of the last ``max_len`` appended (batch_size, ...) blocks."""

    def __init__(self, max_len: int, batch_size: int, device: str):
        if max_len < 1:
            raise ValueError(f"The buffer size should be greater than zero. However, it is set to {max_len}!")
        self._batch_size, self._device, self._max_length = batch_size, torch.device(device), int(max_len)
        self._ALL_INDICES = torch.arange(batch_size, device=self._device)
        self._max_len = torch.full((batch_size,), max_len, dtype=torch.int, device=self._device)
        self._num_pushes = torch.zeros(batch_size, dtype=torch.long, device=self._device)
        self._pointer: int = -1
        self._buffer: torch.Tensor | None = None

    batch_size = property(lambda self: self._batch_size)
    device = property(lambda self: self._device)
    max_length = property(lambda self: self._max_length)

    @property
    def current_length(self) -> torch.Tensor:
        """circular_buffer.py:69-76."""
        return torch.minimum(self._num_pushes, self._max_len)

    def _desc(self) -> RingDesc:
        d = 0 if self._buffer is None else self._buffer[0, 0].numel() if self._batch_size else 0
        return RingDesc(buffer=_abi.ptr(self._buffer), num_pushes=self._num_pushes.data_ptr(), max_len=self._max_length,
                        pointer=max(self._pointer, 0), n=self._batch_size, d=d)

    def _launch(self, data=None, ordered=None, lags=None, delayed=None):
        n = self._batch_size
        src, s_stride = (None, 0) if data is None else _rows(data, n)
        o_stride = 0 if ordered is None else self._out_stride(ordered, n, "ordered_out")
        d_stride = 0 if delayed is None else self._out_stride(delayed, n, "delayed_out")
        desc = self._desc()
        _abi.check(_append(ctypes.byref(desc), None if src is None else src.data_ptr(), s_stride,
                           None if ordered is None else ordered.data_ptr(), o_stride, None if lags is None else lags.data_ptr(),
                           None if delayed is None else delayed.data_ptr(), d_stride, _abi.stream()), "ilx_circular_buffer_append")

    @staticmethod
    def _out_stride(out: torch.Tensor, n: int, what: str) -> int:
        """Row stride of an output tensor the kernel writes in place; a tensor that would have to be copied or cast
        (not float32, or without unit inner stride) is refused -- the kernel would write the copy, not the tensor."""
        rows, stride = _rows(out, n)
        if rows.data_ptr() != out.data_ptr() or rows.dtype != out.dtype:
            raise _abi.IlxError(f"CircularBuffer: {what} must be a float32 GPU tensor whose rows have unit inner stride")
        return stride

    @property
    def buffer(self) -> torch.Tensor:
        """circular_buffer.py:78-86: (batch_size, max_length, ...), oldest entry first."""
        out = torch.empty(self._batch_size, self._max_length, *self._buffer.shape[2:], device=self._device)
        self._launch(ordered=out.view(self._batch_size, -1))
        return out

    def reset(self, batch_ids: Sequence[int] | None = None):
        """circular_buffer.py:92-105."""
        if batch_ids is None or (isinstance(batch_ids, slice) and batch_ids == slice(None)):
            ids, count = None, self._batch_size
        else:
            ids = torch.as_tensor(batch_ids, device=self._device, dtype=torch.long).contiguous()
            count = ids.numel()
        desc = self._desc()
        _abi.check(_reset(ctypes.byref(desc), None if ids is None else ids.data_ptr(), count, _abi.stream()), "ilx_circular_buffer_reset")

    def append(self, data: torch.Tensor, ordered_out: torch.Tensor | None = None, time_lags: torch.Tensor | None = None,
               delayed_out: torch.Tensor | None = None):
        """circular_buffer.py:107-135.  ``ordered_out`` (batch, max_length * width) / ``time_lags`` + ``delayed_out`` (no
        counterpart in the reference) receive ``buffer`` / ``self[time_lags]`` from the same launch."""
        if data.shape[0] != self.batch_size:
            raise ValueError(f"The input data has {data.shape[0]} environments while expecting {self.batch_size}")
        if self._buffer is None:
            self._pointer = -1
            self._buffer = torch.empty((self._max_length, *data.shape), dtype=torch.float32, device=self._device)
        self._pointer = (self._pointer + 1) % self._max_length
        self._launch(data=data, ordered=ordered_out, lags=time_lags, delayed=delayed_out)

    def __getitem__(self, key: torch.Tensor) -> torch.Tensor:
        """circular_buffer.py:137-166: the entry ``key`` pushes back per batch element (the oldest when ``key`` is larger)."""
        if len(key) != self.batch_size:
            raise ValueError(f"The argument 'key' has length {key.shape[0]}, while expecting {self.batch_size}")
        if self._buffer is None or torch.any(self._num_pushes == 0):
            raise RuntimeError("Attempting to retrieve data on an empty circular buffer. Please append data first.")
        out = torch.empty(self._buffer.shape[1:], device=self._device)
        lags = key.to(device=self._device, dtype=torch.int).contiguous()
        self._launch(lags=lags, delayed=out.view(self._batch_size, -1))
        return out
