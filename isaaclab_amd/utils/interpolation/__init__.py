"""``isaaclab/utils/interpolation``."""

from .linear_interpolation import LinearInterpolation  # noqa: F401
