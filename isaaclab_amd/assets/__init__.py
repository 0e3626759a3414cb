"""Assets with the reference's module layout (``isaaclab/assets``)."""

from .articulation import Articulation  # noqa: F401
from .articulation_data import ArticulationData  # noqa: F401
from .rigid_object import RigidObject, RigidObjectData  # noqa: F401
