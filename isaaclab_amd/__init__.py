"""isaaclab_amd — MI355X-native post-physics rollout path for Isaac Lab style environments.

The package mirrors the reference's Python interfaces for this path (``isaaclab.utils.math``, manager term
functions, actuator / ray-caster / differential-IK classes) on top of hand-written gfx950 HIP kernels that
are reached through a C ABI (``include/isaaclab_amd.h``, bound in :mod:`isaaclab_amd._abi`).
"""

__version__ = "0.1.0"
