"""Import-path compatibility with ``isaaclab.utils.warp`` (names only; nothing here uses Warp).

The reference exposes ``raycast_mesh`` and ``convert_to_warp_mesh`` from ``isaaclab.utils.warp``; code written
against that path can import the same names from here and gets the HIP BVH implementation of
:mod:`isaaclab_amd.utils.mesh`.
"""

from ..mesh import Mesh, raycast_mesh
from ..mesh import convert_to_mesh as convert_to_warp_mesh

__all__ = ["Mesh", "raycast_mesh", "convert_to_warp_mesh"]
