"""Noise functions and models with the reference's module layout (``isaaclab/utils/noise``)."""

from .noise_cfg import (ConstantNoiseCfg, GaussianNoiseCfg, NoiseCfg, NoiseModelCfg, NoiseModelWithAdditiveBiasCfg,  # noqa: F401
                        UniformNoiseCfg)
from .noise_model import (NoiseModel, NoiseModelWithAdditiveBias, constant_noise, gaussian_noise, set_noise_seed,  # noqa: F401
                          uniform_noise)
