"""``DelayBuffer`` (interface of ``isaaclab/utils/buffers/delay_buffer.py``): append and delayed read in one launch."""

from __future__ import annotations

from collections.abc import Sequence

import torch

from .circular_buffer import CircularBuffer


class DelayBuffer:
    """delay_buffer.py:16-177: hands back what was appended ``time_lags[env]`` pushes ago."""

    def __init__(self, history_length: int, batch_size: int, device: str):
        self._history_length = max(0, history_length)
        self._circular_buffer = CircularBuffer(self._history_length + 1, batch_size, device)
        self._min_time_lag = self._max_time_lag = 0
        self._time_lags = torch.zeros(batch_size, dtype=torch.int, device=device)

    batch_size = property(lambda self: self._circular_buffer.batch_size)
    device = property(lambda self: self._circular_buffer.device)
    history_length = property(lambda self: self._history_length)
    min_time_lag = property(lambda self: self._min_time_lag)
    max_time_lag = property(lambda self: self._max_time_lag)
    time_lags = property(lambda self: self._time_lags)

    def set_time_lag(self, time_lag: int | torch.Tensor, batch_ids: Sequence[int] | None = None):
        """Install the lag (in appended samples) of the listed batch rows; same argument checks, exception types and
        messages as delay_buffer.py:105-147."""
        rows = slice(None) if batch_ids is None else batch_ids
        if isinstance(time_lag, torch.Tensor):
            if time_lag.dtype not in (torch.int, torch.long):
                raise TypeError(f"Invalid dtype for time_lag: {time_lag.dtype}. Expected torch.int or torch.long.")
            lags = time_lag.to(device=self.device, dtype=torch.int)
        elif isinstance(time_lag, int):
            lags = time_lag
        else:
            raise TypeError(f"Invalid type for time_lag: {type(time_lag)}. Expected int or integer tensor.")
        self._time_lags[rows] = lags
        # both bounds with one device-to-host copy
        lo, hi = (int(v) for v in torch.aminmax(self._time_lags))
        self._min_time_lag, self._max_time_lag = lo, hi
        if lo < 0:
            raise ValueError(f"The minimum time lag cannot be negative. Received: {lo}")
        if hi > self._history_length:
            raise ValueError(f"The maximum time lag cannot be larger than the history length. Received: {hi}")

    def reset(self, batch_ids: Sequence[int] | None = None):
        """delay_buffer.py:149-156."""
        self._circular_buffer.reset(batch_ids)

    def compute(self, data: torch.Tensor) -> torch.Tensor:
        """delay_buffer.py:158-177."""
        out = torch.empty(data.shape, dtype=torch.float32, device=self.device)
        self._circular_buffer.append(data, time_lags=self._time_lags, delayed_out=out.view(data.shape[0], -1))
        return out
