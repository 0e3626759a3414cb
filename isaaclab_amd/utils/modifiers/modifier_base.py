"""``ModifierBase`` (``isaaclab/utils/modifiers/modifier_base.py``)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence

import torch


class ModifierBase(ABC):
    """modifier_base.py:18-78: a modifier with state, built once with the shape of the data it will see."""

    def __init__(self, cfg, data_dim: tuple[int, ...], device: str) -> None:
        self._cfg, self._data_dim, self._device = cfg, tuple(data_dim), device

    @abstractmethod
    def reset(self, env_ids: Sequence[int] | None = None):
        raise NotImplementedError

    @abstractmethod
    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError
