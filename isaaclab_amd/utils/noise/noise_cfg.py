"""Noise configuration records (``isaaclab/utils/noise/noise_cfg.py``)."""

from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass

import torch


@dataclass
class NoiseCfg:
    """noise_cfg.py:18-29.  ``operation``: "add", "scale" or "abs"."""

    func: Callable | None = None
    operation: str = "add"


@dataclass
class ConstantNoiseCfg(NoiseCfg):
    """noise_cfg.py:32-39."""

    bias: torch.Tensor | float = 0.0

    def __post_init__(self):
        if self.func is None:
            from .noise_model import constant_noise

            self.func = constant_noise


@dataclass(init=False)
class UniformNoiseCfg(NoiseCfg):
    """noise_cfg.py:42-51.  The bounds also go first positionally: ``UniformNoiseCfg(-0.1, 0.1)``."""

    n_min: torch.Tensor | float = -1.0
    n_max: torch.Tensor | float = 1.0

    def __init__(self, n_min=-1.0, n_max=1.0, operation: str = "add", func: Callable | None = None):
        self.n_min, self.n_max, self.operation = n_min, n_max, operation
        if func is None:
            from .noise_model import uniform_noise as func
        self.func = func


@dataclass
class GaussianNoiseCfg(NoiseCfg):
    """noise_cfg.py:54-63."""

    mean: torch.Tensor | float = 0.0
    std: torch.Tensor | float = 1.0

    def __post_init__(self):
        if self.func is None:
            from .noise_model import gaussian_noise

            self.func = gaussian_noise


@dataclass
class NoiseModelCfg:
    """noise_cfg.py:71-79."""

    class_type: type | None = None
    noise_cfg: NoiseCfg | None = None

    def __post_init__(self):
        if self.class_type is None:
            from .noise_model import NoiseModel

            self.class_type = NoiseModel


@dataclass
class NoiseModelWithAdditiveBiasCfg(NoiseModelCfg):
    """noise_cfg.py:82-92: the bias is drawn with ``bias_noise_cfg`` at every reset of the model."""

    bias_noise_cfg: NoiseCfg | None = None

    def __post_init__(self):
        if self.class_type is None:
            from .noise_model import NoiseModelWithAdditiveBias

            self.class_type = NoiseModelWithAdditiveBias
