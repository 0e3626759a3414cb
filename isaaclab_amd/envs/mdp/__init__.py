"""Term library with the reference's module layout: ``isaaclab.envs.mdp`` (+ the task packages' extra terms)."""

from .observations import *  # noqa: F401,F403
from .rewards import *  # noqa: F401,F403
from .terminations import *  # noqa: F401,F403
